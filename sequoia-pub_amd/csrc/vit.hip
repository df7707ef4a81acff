// Softmax-attention ViT baseline (src/vit.py:49-115: Attention, Transformer, ViT), forward + backward.
//
// Same flat-parameter design as ViS.  Projections / FeedForward / head are launches of the MFMA GEMM engine
// (NT for activations, TN for weight gradients); the N = 100-token attention core is small (16 heads x 64 dims,
// 100 x 100 scores) and runs in two dedicated kernels, one workgroup per (slide, head), everything in LDS:
//   forward : S = Q K^T / 8, softmax over the 100 keys (wave64 shuffles), O = P V; P kept for backward
//   backward: dP = dO V^T, dS = P * (dP - rowsum(P dP)), dQ = dS K / 8, dK = dS^T Q / 8, dV = P^T dO
#include "../../include/sequoia_hip.h"
#include "elementwise.h"
#include "gemm.h"
#include "vis.h"

namespace {

constexpr int DH = 64;          // dim_head (src/main.py:143,161-163)
constexpr int MAXN = 128;       // tokens per slide handled by the attention kernels

template <typename T> __device__ __forceinline__ float ldT(const T* p);
template <> __device__ __forceinline__ float ldT<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldT<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }
template <typename T> __device__ __forceinline__ void stT(T* p, float v);
template <> __device__ __forceinline__ void stT<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stT<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }

// qkv: [B*N, 3*I] (q | k | v, head h at columns h*64); o: [B*N, I]; P: [B, H, N, N] f32 or NULL
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ o, float* __restrict__ P, int N,
                                                       int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sq = sm;                       // [N][64]
    float* sk = sq + N * DH;              // [N][65]
    float* sv = sk + N * (DH + 1);        // [N][64]
    float* sp = sv + N * DH;              // [4 waves][128]
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int I = H * DH, ld = 3 * I;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const T* base = qkv + (size_t)b * N * ld + h * DH;
    for (int i = tid; i < N * DH; i += 256) {
        const int r = i >> 6, d = i & 63;
        sq[r * DH + d] = ldT<T>(base + (size_t)r * ld + d);
        sk[r * (DH + 1) + d] = ldT<T>(base + (size_t)r * ld + I + d);
        sv[r * DH + d] = ldT<T>(base + (size_t)r * ld + 2 * I + d);
    }
    __syncthreads();
    float* prow = sp + wv * MAXN;
    for (int i = wv; i < N; i += 4) {
        // lane j scores columns j and j + 64
        float s0 = 0.f, s1 = 0.f;
        const int j0 = lane, j1 = lane + 64;
        const bool v0 = j0 < N, v1 = j1 < N;
        for (int d = 0; d < DH; ++d) {
            const float q = sq[i * DH + d];
            if (v0) s0 += q * sk[j0 * (DH + 1) + d];
            if (v1) s1 += q * sk[j1 * (DH + 1) + d];
        }
        s0 = v0 ? s0 * scale : -INFINITY;
        s1 = v1 ? s1 * scale : -INFINITY;
        float mx = fmaxf(s0, s1);
#pragma unroll
        for (int of = 32; of > 0; of >>= 1) mx = fmaxf(mx, __shfl_xor(mx, of, 64));
        const float e0 = v0 ? expf(s0 - mx) : 0.f, e1 = v1 ? expf(s1 - mx) : 0.f;
        const float sum = wave_sum(e0 + e1);
        const float p0 = e0 / sum, p1 = e1 / sum;
        if (v0) prow[j0] = p0;
        if (v1) prow[j1] = p1;
        if (P) {
            float* pr = P + (((size_t)b * H + h) * N + i) * N;
            if (v0) pr[j0] = p0;
            if (v1) pr[j1] = p1;
        }
        // O[i][d = lane] = sum_j P[i][j] V[j][d]   (prow was written by this wave only: wave-synchronous LDS)
        __builtin_amdgcn_wave_barrier();
        float acc = 0.f;
        for (int j = 0; j < N; ++j) acc += prow[j] * sv[j * DH + lane];
        stT<T>(o + ((size_t)b * N + i) * I + h * DH + lane, acc);
        __builtin_amdgcn_wave_barrier();
    }
}

// dO: [B*N, I] f32; dqkv: [B*N, 3*I] T
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const T* __restrict__ qkv, const float* __restrict__ P, const float* __restrict__ dO,
                                                       T* __restrict__ dqkv, int N, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sq = sm;                        // [N][64]
    float* sk = sq + N * DH;               // [N][64]
    float* sv = sk + N * DH;               // [N][65]
    float* sdo = sv + N * (DH + 1);        // [N][64]
    float* sds = sdo + N * DH;             // [N][N]  dS
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int I = H * DH, ld = 3 * I;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const T* base = qkv + (size_t)b * N * ld + h * DH;
    const float* Pb = P + ((size_t)b * H + h) * N * N;
    for (int i = tid; i < N * DH; i += 256) {
        const int r = i >> 6, d = i & 63;
        sq[r * DH + d] = ldT<T>(base + (size_t)r * ld + d);
        sk[r * DH + d] = ldT<T>(base + (size_t)r * ld + I + d);
        sv[r * (DH + 1) + d] = ldT<T>(base + (size_t)r * ld + 2 * I + d);
        sdo[r * DH + d] = dO[((size_t)b * N + r) * I + h * DH + d];
    }
    __syncthreads();
    T* dq = dqkv + (size_t)b * N * ld + h * DH;
    // phase 1: rows of dS, dQ
    for (int i = wv; i < N; i += 4) {
        const int j0 = lane, j1 = lane + 64;
        const bool v0 = j0 < N, v1 = j1 < N;
        float d0 = 0.f, d1 = 0.f;
        for (int d = 0; d < DH; ++d) {
            const float g = sdo[i * DH + d];
            if (v0) d0 += g * sv[j0 * (DH + 1) + d];
            if (v1) d1 += g * sv[j1 * (DH + 1) + d];
        }
        const float p0 = v0 ? Pb[(size_t)i * N + j0] : 0.f, p1 = v1 ? Pb[(size_t)i * N + j1] : 0.f;
        const float dot = wave_sum(p0 * d0 + p1 * d1);
        const float ds0 = p0 * (d0 - dot), ds1 = p1 * (d1 - dot);
        if (v0) sds[i * N + j0] = ds0;
        if (v1) sds[i * N + j1] = ds1;
        __builtin_amdgcn_wave_barrier();
        float acc = 0.f;
        for (int j = 0; j < N; ++j) acc += sds[i * N + j] * sk[j * DH + lane];
        stT<T>(dq + (size_t)i * ld + lane, acc * scale);
    }
    __syncthreads();
    // phase 2: dK[j][d] = scale * sum_i dS[i][j] Q[i][d];  dV[j][d] = sum_i P[i][j] dO[i][d]
    for (int j = wv; j < N; j += 4) {
        float ak = 0.f, av = 0.f;
        for (int i = 0; i < N; ++i) {
            ak += sds[i * N + j] * sq[i * DH + lane];
            av += Pb[(size_t)i * N + j] * sdo[i * DH + lane];
        }
        stT<T>(dq + (size_t)j * ld + I + lane, ak * scale);
        stT<T>(dq + (size_t)j * ld + 2 * I + lane, av);
    }
}

struct VitBufs {
    float* Xin[SQ_MAX_DEPTH + 1]; void* Xin_lp[SQ_MAX_DEPTH + 1];
    void* Xn[SQ_MAX_DEPTH]; void* QKV[SQ_MAX_DEPTH]; float* P[SQ_MAX_DEPTH]; void* O[SQ_MAX_DEPTH];
    float* X1[SQ_MAX_DEPTH]; void* Y[SQ_MAX_DEPTH]; float* U[SQ_MAX_DEPTH]; void* H1[SQ_MAX_DEPTH];
    float* xm; void* xn;
    size_t bytes;
};

void vit_bufs(const sq_vit_config& c, int dtype, int B, int save, char* base, VitBufs* o) {
    Arena a{base, 0};
    const size_t es = sq_dtype_size(dtype);
    const size_t N = c.num_clusters, M = (size_t)B * N, D = c.dim, I = (size_t)c.heads * DH, F = c.mlp_dim;
    const int L = save ? c.depth : 1, nX = save ? c.depth + 1 : 1;
    for (int l = 0; l <= SQ_MAX_DEPTH; ++l) {
        if (l < nX) {
            o->Xin[l] = (float*)a.take(M * D * 4);
            o->Xin_lp[l] = dtype == SQ_BF16 ? a.take(M * D * 2) : (void*)o->Xin[l];
        } else { o->Xin[l] = o->Xin[0]; o->Xin_lp[l] = o->Xin_lp[0]; }
    }
    for (int l = 0; l < SQ_MAX_DEPTH; ++l) {
        if (l < L) {
            o->Xn[l] = a.take(M * D * es);
            o->QKV[l] = a.take(M * 3 * I * es);
            o->P[l] = save ? (float*)a.take((size_t)B * c.heads * N * N * 4) : nullptr;
            o->O[l] = a.take(M * I * es);
            o->X1[l] = (float*)a.take(M * D * 4);
            o->Y[l] = a.take(M * D * es);
            o->U[l] = save ? (float*)a.take(M * F * 4) : nullptr;
            o->H1[l] = a.take(M * F * es);
        } else {
            o->Xn[l] = o->Xn[0]; o->QKV[l] = o->QKV[0]; o->P[l] = o->P[0]; o->O[l] = o->O[0]; o->X1[l] = o->X1[0];
            o->Y[l] = o->Y[0]; o->U[l] = o->U[0]; o->H1[l] = o->H1[0];
        }
    }
    o->xm = (float*)a.take((size_t)B * D * 4);
    o->xn = a.take((size_t)B * D * es);
    o->bytes = sq_align_up(a.off, 256);
}

int check_vit(const sq_vit_config* c) {
    SQ_REQUIRE(c != nullptr, "vit: null config");
    SQ_REQUIRE(c->dim > 0 && c->dim % 64 == 0 && c->dim <= 4096, "vit: dim=%d must be a multiple of 64, <= 4096", c->dim);
    SQ_REQUIRE(c->depth >= 1 && c->depth <= SQ_MAX_DEPTH, "vit: depth=%d", c->depth);
    SQ_REQUIRE(c->heads >= 1 && c->heads <= 64, "vit: heads=%d", c->heads);
    SQ_REQUIRE(c->mlp_dim > 0 && c->mlp_dim % 64 == 0, "vit: mlp_dim=%d must be a multiple of 64", c->mlp_dim);
    SQ_REQUIRE(c->num_outputs >= 1 && c->num_clusters >= 1 && c->num_clusters <= MAXN, "vit: num_outputs=%d num_clusters=%d (<= %d)", c->num_outputs, c->num_clusters, MAXN);
    return SQ_OK;
}

struct VitBwdBufs {
    float* dXa; void* dXa_lp; float* dXb; void* dXb_lp;
    void* dU; float* dY; float* dO; void* dQKV; float* dXn; void* wT; void* dout_lp; void* whT; float* dxn; float* dxm; float* red_ws;
    float* skws; size_t skws_bytes; size_t bytes;
};

void vit_bwd_bufs(const sq_vit_config& c, int dtype, int B, char* base, VitBwdBufs* o) {
    Arena a{base, 0};
    const size_t es = sq_dtype_size(dtype);
    const bool lp = dtype == SQ_BF16;
    const size_t N = c.num_clusters, M = (size_t)B * N, D = c.dim, I = (size_t)c.heads * DH, F = c.mlp_dim, G = c.num_outputs;
    const size_t Gp = sq_align_up(G, 8);
    size_t W = D; if (3 * I > W) W = 3 * I; if (F > W) W = F;
    o->dXa = (float*)a.take(M * D * 4); o->dXa_lp = lp ? a.take(M * D * 2) : (void*)o->dXa;
    o->dXb = (float*)a.take(M * D * 4); o->dXb_lp = lp ? a.take(M * D * 2) : (void*)o->dXb;
    o->dU = a.take(M * F * es);
    o->dY = (float*)a.take(M * D * 4);
    o->dO = (float*)a.take(M * I * 4);
    o->dQKV = a.take(M * 3 * I * es);
    o->dXn = (float*)a.take(M * D * 4);
    o->wT = a.take(W * (D > F ? D : F) * es);
    o->dout_lp = a.take((size_t)B * Gp * es);
    o->whT = a.take(D * Gp * es);
    o->dxn = (float*)a.take((size_t)B * D * 4); o->dxm = (float*)a.take((size_t)B * D * 4);
    size_t red = sq_ln_bwd_ws_floats((int)D);
    const size_t cs = sq_colsum_ws_floats((int)(G > W ? G : W));
    if (cs > red) red = cs;
    o->red_ws = (float*)a.take(red * 4);
    o->skws_bytes = (size_t)4 * W * (D > F ? D : F) * 4;
    o->skws = (float*)a.take(o->skws_bytes);
    o->bytes = sq_align_up(a.off, 256);
}

size_t attn_fwd_lds(int N) { return (size_t)(N * DH * 2 + N * (DH + 1) + 4 * MAXN) * 4; }
size_t attn_bwd_lds(int N) { return (size_t)(N * DH * 3 + N * (DH + 1) + N * N) * 4; }

}  // namespace

extern "C" int sq_vit_layout_init(const sq_vit_config* c, sq_vit_layout* out) {
    if (int e = check_vit(c)) return e;
    SQ_REQUIRE(out != nullptr, "vit: null layout");
    const int64_t D = c->dim, I = (int64_t)c->heads * DH, F = c->mlp_dim, G = c->num_outputs;
    int64_t off = 0;
    auto take = [&](int64_t n) { off = (off + 7) / 8 * 8; const int64_t o = off; off += n; return o; };
    out->pos = take((int64_t)c->num_clusters * D);
    for (int l = 0; l < SQ_MAX_DEPTH; ++l) {
        sq_vit_layer_offsets& L = out->layer[l];
        if (l >= c->depth) { L = sq_vit_layer_offsets{-1, -1, -1, -1, -1, -1, -1, -1, -1, -1}; continue; }
        L.ln1_g = take(D); L.ln1_b = take(D);
        L.qkv_w = take(3 * I * D);
        L.out_w = take(D * I);
        L.ln2_g = take(D); L.ln2_b = take(D);
        L.ff1_w = take(F * D); L.ff1_b = take(F);
        L.ff2_w = take(D * F); L.ff2_b = take(D);
    }
    out->head_ln_g = take(D); out->head_ln_b = take(D);
    out->head_w = take(G * D); out->head_b = take(G);
    out->total = (off + 7) / 8 * 8;
    return SQ_OK;
}

extern "C" size_t sq_vit_workspace_bytes(const sq_vit_config* c, int dtype, int batch, int save) {
    if (check_vit(c) != SQ_OK || batch < 1) return 0;
    VitBufs b;
    vit_bufs(*c, dtype, batch, save, nullptr, &b);
    return b.bytes;
}

extern "C" size_t sq_vit_backward_workspace_bytes(const sq_vit_config* c, int dtype, int batch) {
    if (check_vit(c) != SQ_OK || batch < 1) return 0;
    VitBwdBufs b;
    vit_bwd_bufs(*c, dtype, batch, nullptr, &b);
    return b.bytes;
}

#define RUN(expr) do { if (int _e = (expr)) return _e; } while (0)

extern "C" int sq_vit_forward(const sq_vit_config* c, int dtype, const float* params, const void* params_lp, const float* x,
                              float* out, int B, int save, void* workspace, size_t workspace_bytes, sq_stream_t stream_) {
    if (int e = check_vit(c)) return e;
    hipStream_t st = (hipStream_t)stream_;
    SQ_REQUIRE(dtype == SQ_F32 || dtype == SQ_BF16, "vit_forward: dtype %d", dtype);
    SQ_REQUIRE(params && x && out && workspace && B >= 1, "vit_forward: bad arguments");
    SQ_REQUIRE(dtype == SQ_F32 || params_lp, "vit_forward: bf16 mode needs the bf16 parameter shadow");
    sq_vit_layout lay;
    RUN(sq_vit_layout_init(c, &lay));
    VitBufs w;
    vit_bufs(*c, dtype, B, save, (char*)workspace, &w);
    if (w.bytes > workspace_bytes) { sq_set_error("vit_forward: workspace %zu < required %zu", workspace_bytes, w.bytes); return SQ_ERR_WORKSPACE; }
    const int N = c->num_clusters, D = c->dim, H = c->heads, I = H * DH, F = c->mlp_dim, G = c->num_outputs, M = B * N;
    const size_t es = sq_dtype_size(dtype);
    const bool lp = dtype == SQ_BF16;
    const char* wbase = lp ? (const char*)params_lp : (const char*)params;
    auto W = [&](int64_t off) { return (const void*)(wbase + (size_t)off * es); };
    auto Wrem = [&](int64_t off) { return (size_t)(lay.total - off) * es; };
    auto Pf = [&](int64_t off) { return params + off; };
    SQ_HIP_CHECK(hipFuncSetAttribute((const void*)attn_fwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    SQ_HIP_CHECK(hipFuncSetAttribute((const void*)attn_fwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    RUN(sq_k_add_pos(x, Pf(lay.pos), w.Xin[0], lp ? (bf16_t*)w.Xin_lp[0] : nullptr, B, N, D, st));
    for (int l = 0; l < c->depth; ++l) {
        const sq_vit_layer_offsets& L = lay.layer[l];
        const int s = save ? l : 0;
        float* Xin = w.Xin[s];
        float* Xout = w.Xin[save ? l + 1 : 0];
        void* Xout_lp = w.Xin_lp[save ? l + 1 : 0];
        RUN(sq_k_ln_rows(Xin, Pf(L.ln1_g), Pf(L.ln1_b), w.Xn[s], dtype, M, D, nullptr, nullptr, st));      // vit.py:63
        {   // qkv = to_qkv(x)   (no bias, vit.py:59,65)
            GemmArgs g; g.A = w.Xn[s]; g.lda = D; g.a_bytes = (size_t)M * D * es;
            g.B = W(L.qkv_w); g.ldb = D; g.b_bytes = Wrem(L.qkv_w);
            g.C = w.QKV[s]; g.out_dtype = dtype; g.ldc = 3 * I; g.M = M; g.N = 3 * I; g.K = D;
            RUN(sq_launch_gemm(g, dtype, st));
        }
        if (lp) hipLaunchKernelGGL(attn_fwd_kernel<bf16_t>, dim3(B * H), dim3(256), attn_fwd_lds(N), st, (const bf16_t*)w.QKV[s], (bf16_t*)w.O[s], w.P[s], N, H, 0.125f);
        else hipLaunchKernelGGL(attn_fwd_kernel<float>, dim3(B * H), dim3(256), attn_fwd_lds(N), st, (const float*)w.QKV[s], (float*)w.O[s], w.P[s], N, H, 0.125f);
        SQ_LAUNCH_CHECK();
        {   // x = to_out(out) + x   (no bias, vit.py:60,74,87)
            GemmArgs g; g.A = w.O[s]; g.lda = I; g.a_bytes = (size_t)M * I * es;
            g.B = W(L.out_w); g.ldb = I; g.b_bytes = Wrem(L.out_w);
            g.res = Xin; g.ldres = D; g.C = w.X1[s]; g.ldc = D; g.M = M; g.N = D; g.K = I;
            RUN(sq_launch_gemm(g, dtype, st));
        }
        RUN(sq_k_ln_rows(w.X1[s], Pf(L.ln2_g), Pf(L.ln2_b), w.Y[s], dtype, M, D, nullptr, nullptr, st));
        {
            GemmArgs g; g.A = w.Y[s]; g.lda = D; g.a_bytes = (size_t)M * D * es;
            g.B = W(L.ff1_w); g.ldb = D; g.b_bytes = Wrem(L.ff1_w); g.bias = Pf(L.ff1_b);
            g.act = SQ_ACT_GELU; g.Cpre = w.U[s]; g.ldpre = F;
            g.C = w.H1[s]; g.out_dtype = dtype; g.ldc = F; g.M = M; g.N = F; g.K = D;
            RUN(sq_launch_gemm(g, dtype, st));
        }
        {
            GemmArgs g; g.A = w.H1[s]; g.lda = F; g.a_bytes = (size_t)M * F * es;
            g.B = W(L.ff2_w); g.ldb = F; g.b_bytes = Wrem(L.ff2_w); g.bias = Pf(L.ff2_b);
            g.res = w.X1[s]; g.ldres = D; g.C = Xout; g.ldc = D; g.C2 = lp ? (bf16_t*)Xout_lp : nullptr; g.ldc2 = D;
            g.M = M; g.N = D; g.K = F;
            RUN(sq_launch_gemm(g, dtype, st));
        }
    }
    const float* Xfin = w.Xin[save ? c->depth : 0];
    RUN(sq_k_token_mean(Xfin, w.xm, nullptr, B, N, D, st));
    RUN(sq_k_ln_rows(w.xm, Pf(lay.head_ln_g), Pf(lay.head_ln_b), w.xn, dtype, B, D, nullptr, nullptr, st));
    {
        GemmArgs g; g.A = w.xn; g.lda = D; g.a_bytes = (size_t)B * D * es;
        g.B = W(lay.head_w); g.ldb = D; g.b_bytes = Wrem(lay.head_w); g.bias = Pf(lay.head_b);
        g.C = out; g.ldc = G; g.M = B; g.N = G; g.K = D;
        RUN(sq_launch_gemm(g, dtype, st));
    }
    return SQ_OK;
}

extern "C" int sq_vit_backward(const sq_vit_config* c, int dtype, const float* params, const void* params_lp,
                               const float* grad_out, float* grad_params, float* grad_x, int B, void* fwd_workspace,
                               size_t fwd_workspace_bytes, void* bwd_workspace, size_t bwd_workspace_bytes, sq_stream_t stream_) {
    if (int e = check_vit(c)) return e;
    hipStream_t st = (hipStream_t)stream_;
    SQ_REQUIRE(params && grad_out && grad_params && fwd_workspace && bwd_workspace, "vit_backward: null pointer");
    SQ_REQUIRE(dtype == SQ_F32 || params_lp, "vit_backward: bf16 mode needs the bf16 parameter shadow");
    sq_vit_layout lay;
    RUN(sq_vit_layout_init(c, &lay));
    VitBufs w; vit_bufs(*c, dtype, B, 1, (char*)fwd_workspace, &w);
    VitBwdBufs b; vit_bwd_bufs(*c, dtype, B, (char*)bwd_workspace, &b);
    if (w.bytes > fwd_workspace_bytes || b.bytes > bwd_workspace_bytes) {
        sq_set_error("vit_backward: workspaces %zu/%zu < required %zu/%zu", fwd_workspace_bytes, bwd_workspace_bytes, w.bytes, b.bytes);
        return SQ_ERR_WORKSPACE;
    }
    const int N = c->num_clusters, D = c->dim, H = c->heads, I = H * DH, F = c->mlp_dim, G = c->num_outputs, M = B * N;
    SQ_REQUIRE(attn_bwd_lds(N) <= 160 * 1024, "vit_backward: num_clusters=%d needs %zu B of LDS (> 160 KiB)", N, attn_bwd_lds(N));
    SQ_HIP_CHECK(hipFuncSetAttribute((const void*)attn_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    SQ_HIP_CHECK(hipFuncSetAttribute((const void*)attn_bwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int Gp = (int)sq_align_up(G, 8);
    const int es = sq_dtype_size(dtype);
    const bool lp = dtype == SQ_BF16;
    const char* wbase = lp ? (const char*)params_lp : (const char*)params;
    auto W = [&](int64_t off) { return (const void*)(wbase + (size_t)off * es); };
    auto Pf = [&](int64_t off) { return params + off; };
    auto Gr = [&](int64_t off) { return grad_params + off; };
    auto nt = [&](const void* A, int lda, const void* Bm, int ldb, float* C, int ldc, int M_, int N_, int K_) {
        GemmArgs g;
        g.A = A; g.lda = lda; g.a_bytes = ((size_t)(M_ - 1) * lda + K_) * es;
        g.B = Bm; g.ldb = ldb; g.b_bytes = ((size_t)(N_ - 1) * ldb + K_) * es;
        g.C = C; g.ldc = ldc; g.M = M_; g.N = N_; g.K = K_; g.splitk_ws = b.skws; g.splitk_ws_bytes = b.skws_bytes;
        return g;
    };
    auto tn = [&](const void* A, int lda, const void* Bm, int ldb, float* C, int ldc, int M_, int N_, int K_) {
        GemmArgs g;
        g.A = A; g.lda = lda; g.a_bytes = (size_t)K_ * lda * es;
        g.B = Bm; g.ldb = ldb; g.b_bytes = (size_t)K_ * ldb * es;
        g.C = C; g.ldc = ldc; g.M = M_; g.N = N_; g.K = K_; g.splitk_ws = b.skws; g.splitk_ws_bytes = b.skws_bytes;
        return g;
    };
    auto tr = [&](const void* src, int lds_, void* dst, int ldd, int R, int C) { return sq_k_transpose(src, lds_, dst, ldd, R, C, es, 1, 0, 0, st); };

    // head
    RUN(sq_k_cast_pad(grad_out, G, b.dout_lp, dtype, Gp, B, G, st));
    { GemmArgs g = tn(b.dout_lp, Gp, w.xn, D, Gr(lay.head_w), D, G, D, B); RUN(sq_launch_gemm_tn(g, dtype, st)); }
    RUN(sq_k_colsum(grad_out, SQ_F32, B, G, G, b.red_ws, Gr(lay.head_b), st));
    RUN(tr(W(lay.head_w), D, b.whT, Gp, G, D));
    { GemmArgs g = nt(b.dout_lp, Gp, b.whT, Gp, b.dxn, D, B, D, Gp); RUN(sq_launch_gemm(g, dtype, st)); }
    RUN(sq_k_ln_rows_bwd(b.dxn, w.xm, Pf(lay.head_ln_g), nullptr, b.dxm, nullptr, Gr(lay.head_ln_g), Gr(lay.head_ln_b), b.red_ws, B, D, st));
    RUN(sq_k_bcast_rows(b.dxm, 1.0f / (float)N, b.dXa, lp ? (bf16_t*)b.dXa_lp : nullptr, B, N, D, st));
    float* dXcur = b.dXa; void* dXcur_lp = b.dXa_lp;
    float* dXoth = b.dXb; void* dXoth_lp = b.dXb_lp;
    for (int l = c->depth - 1; l >= 0; --l) {
        const sq_vit_layer_offsets& L = lay.layer[l];
        // FeedForward
        { GemmArgs g = tn(dXcur_lp, D, w.H1[l], F, Gr(L.ff2_w), F, D, F, M); RUN(sq_launch_gemm_tn(g, dtype, st)); }
        RUN(sq_k_colsum(dXcur, SQ_F32, M, D, D, b.red_ws, Gr(L.ff2_b), st));
        RUN(tr(W(L.ff2_w), F, b.wT, D, D, F));                       // W2 [D, F] -> [F, D]
        {
            GemmArgs g = nt(dXcur_lp, D, b.wT, D, nullptr, F, M, F, D);
            g.C = b.dU; g.out_dtype = dtype; g.gelu_grad_of = w.U[l]; g.ldgg = F;
            RUN(sq_launch_gemm(g, dtype, st));
        }
        { GemmArgs g = tn(b.dU, F, w.Y[l], D, Gr(L.ff1_w), D, F, D, M); RUN(sq_launch_gemm_tn(g, dtype, st)); }
        RUN(sq_k_colsum(b.dU, dtype, M, F, F, b.red_ws, Gr(L.ff1_b), st));
        RUN(tr(W(L.ff1_w), D, b.wT, F, F, D));                       // W1 [F, D] -> [D, F]
        { GemmArgs g = nt(b.dU, F, b.wT, F, b.dY, D, M, D, F); RUN(sq_launch_gemm(g, dtype, st)); }
        RUN(sq_k_ln_rows_bwd(b.dY, w.X1[l], Pf(L.ln2_g), dXcur, dXoth, lp ? (bf16_t*)dXoth_lp : nullptr, Gr(L.ln2_g), Gr(L.ln2_b), b.red_ws, M, D, st));
        float* dX1 = dXoth; void* dX1_lp = dXoth_lp;
        // Attention
        { GemmArgs g = tn(dX1_lp, D, w.O[l], I, Gr(L.out_w), I, D, I, M); RUN(sq_launch_gemm_tn(g, dtype, st)); }
        RUN(tr(W(L.out_w), I, b.wT, D, D, I));                       // Wout [D, I] -> [I, D]
        { GemmArgs g = nt(dX1_lp, D, b.wT, D, b.dO, I, M, I, D); RUN(sq_launch_gemm(g, dtype, st)); }
        if (lp) hipLaunchKernelGGL(attn_bwd_kernel<bf16_t>, dim3(B * H), dim3(256), attn_bwd_lds(N), st, (const bf16_t*)w.QKV[l], w.P[l], b.dO, (bf16_t*)b.dQKV, N, H, 0.125f);
        else hipLaunchKernelGGL(attn_bwd_kernel<float>, dim3(B * H), dim3(256), attn_bwd_lds(N), st, (const float*)w.QKV[l], w.P[l], b.dO, (float*)b.dQKV, N, H, 0.125f);
        SQ_LAUNCH_CHECK();
        { GemmArgs g = tn(b.dQKV, 3 * I, w.Xn[l], D, Gr(L.qkv_w), D, 3 * I, D, M); RUN(sq_launch_gemm_tn(g, dtype, st)); }
        RUN(tr(W(L.qkv_w), D, b.wT, 3 * I, 3 * I, D));               // Wqkv [3I, D] -> [D, 3I]
        { GemmArgs g = nt(b.dQKV, 3 * I, b.wT, 3 * I, b.dXn, D, M, D, 3 * I); RUN(sq_launch_gemm(g, dtype, st)); }
        RUN(sq_k_ln_rows_bwd(b.dXn, w.Xin[l], Pf(L.ln1_g), dX1, dXcur, lp ? (bf16_t*)dXcur_lp : nullptr, Gr(L.ln1_g), Gr(L.ln1_b), b.red_ws, M, D, st));
    }
    RUN(sq_k_batch_sum(dXcur, grad_params + lay.pos, B, N * D, st));
    if (grad_x) SQ_HIP_CHECK(hipMemcpyAsync(grad_x, dXcur, (size_t)M * D * 4, hipMemcpyDeviceToDevice, st));
    return SQ_OK;
}
#undef RUN
