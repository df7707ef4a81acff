// UNI patch embedder: timm's VisionTransformer ``vit_large_patch16_224`` as the reference builds it
// (pre_processing/compute_features_hdf5.py:62-68: ``timm.create_model("vit_large_patch16_224", img_size=224,
// patch_size=16, init_values=1e-5, num_classes=0, dynamic_img_size=True)``; spatial_vis/visualize.py:220-232), forward
// only, on the MFMA GEMM engine.  timm and the gated UNI weights are absent from the build image, so this follows the
// PUBLISHED timm algorithm (timm/models/vision_transformer.py: PatchEmbed, Attention, LayerScale, Block,
// VisionTransformer.forward_features / forward_head with global_pool='token') -- parity unpinned (oracle/uni_oracle.py).
//
//   tokens  = [cls ; conv16x16/16(x) + b] + pos_embed                                     [T = 1 + (S/16)^2, D]
//   block   : x += ls1 * proj(softmax(q k^T / sqrt(64)) v),  (q, k, v) = qkv(LN(x))       LN eps 1e-6, qkv / proj biased
//             x += ls2 * fc2(GELU(fc1(LN(x))))
//   output  = LN(x)[cls]                                                                   [D]
// The LayerScale gains are folded into proj / fc2 (weight rows and bias) by the host when it builds the execution copy
// of the parameters, so a block is LN, 4 GEMM launches (bias / residual / GELU epilogues) and the attention core.
// Patches arrive as uint8 HWC (ToTensor + Normalize of compute_features_hdf5.py:53-56 fused into the im2col) or as
// normalised fp32 NCHW tensors (the reference call form).  Resize(224) is the identity for 224 x 224 patches; other
// sizes must be resized by the caller (S % 16 == 0 is all the kernels need: dynamic_img_size without pos-embed
// interpolation is only defined for the trained grid, so S == img_size is enforced).
#include "../../include/sequoia_hip.h"
#include "gemm.h"
#include "vis.h"

namespace {

constexpr int DH = 64;
constexpr int PS = 16;                    // patch edge
constexpr int KP = 3 * PS * PS;           // 768 = im2col row length (c, kh, kw)
constexpr float MEAN[3] = {0.485f, 0.456f, 0.406f};
constexpr float STD[3] = {0.229f, 0.224f, 0.225f};
constexpr int ATT_MAXT = 256;             // tokens the attention kernel handles (197 for the 224 px grid)

__device__ __forceinline__ float ld(const float* p) { return *p; }
__device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
__device__ __forceinline__ void st(float* p, float v) { *p = v; }
__device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }

// A0[(img, py, px), (c, kh, kw)] = normalised pixel; thread = 8 consecutive kw of one (row, c, kh)
template <typename T>
__global__ __launch_bounds__(256) void uni_im2col_kernel(const uint8_t* __restrict__ u8, const float* __restrict__ f32, T* __restrict__ out,
                                                         int n, int S) {
    const int G = S / PS;
    const uint32_t total = (uint32_t)n * G * G * (KP / 8);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t ch = i % (KP / 8), row = i / (KP / 8);
        const int k0 = ch * 8, c = k0 / (PS * PS), kh = (k0 / PS) % PS, kw0 = k0 % PS;
        const int px = row % G, py = (row / G) % G, img = row / (G * G);
        const int y = py * PS + kh, x0 = px * PS + kw0;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (u8) v[e] = ((float)u8[(((size_t)img * S + y) * S + x0 + e) * 3 + c] / 255.0f - MEAN[c]) / STD[c];
            else v[e] = f32[(((size_t)img * 3 + c) * S + y) * S + x0 + e];
        }
        T* o = out + (size_t)row * KP + k0;
#pragma unroll
        for (int e = 0; e < 8; ++e) st(o + e, v[e]);
    }
}

// X[img, 0] = cls + pos[0];  X[img, 1 + p] = E[img, p] + pos[1 + p]
__global__ void uni_tokens_kernel(const float4* __restrict__ E, const float4* __restrict__ cls, const float4* __restrict__ pos,
                                  float4* __restrict__ X, int n, int T, int D4) {
    const uint32_t total = (uint32_t)n * T * D4;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t d = i % D4, t = (i / D4) % T, img = i / (D4 * T);
        const float4 a = t == 0 ? cls[d] : E[((size_t)img * (T - 1) + (t - 1)) * D4 + d];
        const float4 p = pos[(size_t)t * D4 + d];
        X[i] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
    }
}

// y[r] = LayerNorm(x[r * row_stride]) * g + b, one wave per row, D <= 4096, two-pass statistics in fp32
template <typename T>
__global__ __launch_bounds__(256) void uni_ln_kernel(const float* __restrict__ x, size_t row_stride, const float* __restrict__ g,
                                                     const float* __restrict__ b, T* __restrict__ y, int R, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const float* xr = x + (size_t)row * row_stride;
    float v[64];                               // D / 64 values per lane
    const int per = (D + 63) / 64;
    float s = 0.f;
    for (int i = 0; i < per; ++i) {
        const int c = i * 64 + lane;
        v[i] = c < D ? xr[c] : 0.f;
        s += v[i];
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
    for (int i = 0; i < per; ++i) {
        const int c = i * 64 + lane;
        if (c < D) { const float t = v[i] - mean; q += t * t; }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
    for (int i = 0; i < per; ++i) {
        const int c = i * 64 + lane;
        if (c < D) st(y + (size_t)row * D + c, (v[i] - mean) * rstd * g[c] + b[c]);
    }
}

// The same for D = 512 NI (ViT-L: D = 1024, NI = 2): a lane owns 8 consecutive columns per 512-column block -- two 16-byte loads,
// one 16-byte bf16 store (or two fp32 ones), everything in registers with compile-time indices.  The generic kernel above indexes
// v[] with a run-time trip count (scratch memory) and stores 2 bytes per lane: 190 us per [50432, 1024] call = 1.6 TB/s,
// 19 % of the UNI forward (rocprofv3, round 4); this one streams the same rows at the fabric's rate.
template <typename T, int NI>
__global__ __launch_bounds__(256) void uni_ln8_kernel(const float* __restrict__ x, size_t row_stride, const float* __restrict__ g,
                                                      const float* __restrict__ b, T* __restrict__ y, int R, float eps) {
    constexpr int D = 512 * NI;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const float* xr = x + (size_t)row * row_stride + lane * 8;
    float v[NI][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(xr + i * 512), t1 = *reinterpret_cast<const f32x4*>(xr + i * 512 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[i][e] = t0[e]; v[i][4 + e] = t1[e]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[i][e];
    }
    const float mean = wave_sum(s) * (1.0f / (float)D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[i][e] -= mean; q += v[i][e] * v[i][e]; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / (float)D) + eps);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = i * 512 + lane * 8;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(g + c), g1 = *reinterpret_cast<const f32x4*>(g + c + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(b + c), b1 = *reinterpret_cast<const f32x4*>(b + c + 4);
        float o[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = v[i][e] * rstd * g0[e] + b0[e]; o[4 + e] = v[i][4 + e] * rstd * g1[e] + b1[e]; }
        if constexpr (sizeof(T) == 2) {
            *reinterpret_cast<u32x4*>(y + (size_t)row * D + c) = u32x4{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
        } else {
            float* d = reinterpret_cast<float*>(y) + (size_t)row * D + c;
            *reinterpret_cast<f32x4*>(d) = f32x4{o[0], o[1], o[2], o[3]};
            *reinterpret_cast<f32x4*>(d + 4) = f32x4{o[4], o[5], o[6], o[7]};
        }
    }
}

// softmax(q k^T * scale) v for one (image, head): q / k / v rows of this head staged in LDS as fp32,
// wave w owns query rows w, w+4, ...; lane = key (4 per lane) for the scores, lane = channel for P V
template <typename T>
__global__ __launch_bounds__(256) void uni_attn_kernel(const T* __restrict__ qkv, T* __restrict__ o, int Ttok, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sq = sm;                          // [T][64]
    float* sk = sq + Ttok * DH;              // [T][65]
    float* sv = sk + Ttok * (DH + 1);        // [T][64]
    float* sp = sv + Ttok * DH;              // [4 waves][ATT_MAXT]
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int I = H * DH, ldq = 3 * I;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const T* base = qkv + (size_t)b * Ttok * ldq + h * DH;
    for (int i = tid; i < Ttok * DH; i += 256) {
        const int r = i >> 6, d = i & 63;
        sq[r * DH + d] = ld(base + (size_t)r * ldq + d);
        sk[r * (DH + 1) + d] = ld(base + (size_t)r * ldq + I + d);
        sv[r * DH + d] = ld(base + (size_t)r * ldq + 2 * I + d);
    }
    __syncthreads();
    float* prow = sp + wv * ATT_MAXT;
    for (int i = wv; i < Ttok; i += 4) {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int d = 0; d < DH; ++d) {
            const float q = sq[i * DH + d];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = lane + 64 * u;
                if (j < Ttok) s[u] += q * sk[j * (DH + 1) + d];
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s[u] = lane + 64 * u < Ttok ? s[u] * scale : -INFINITY;
            mx = fmaxf(mx, s[u]);
        }
#pragma unroll
        for (int of = 32; of > 0; of >>= 1) mx = fmaxf(mx, __shfl_xor(mx, of, 64));
        float e[4], part = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) { e[u] = lane + 64 * u < Ttok ? expf(s[u] - mx) : 0.f; part += e[u]; }
        const float sum = wave_sum(part);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (lane + 64 * u < Ttok) prow[lane + 64 * u] = e[u] / sum;
        __builtin_amdgcn_wave_barrier();
        float acc = 0.f;
        for (int j = 0; j < Ttok; ++j) acc += prow[j] * sv[j * DH + lane];
        st(o + ((size_t)b * Ttok + i) * I + h * DH + lane, acc);
        __builtin_amdgcn_wave_barrier();
    }
}


// ---- bf16 mode: the same attention on the matrix cores --------------------------------------------------------
// One workgroup per (image, head); K [Tp][64] and V [Tp][64] of the head sit in LDS as bf16, both row-major with 16-byte
// writes (Tp = T rounded up to 32, padded keys are masked).  V is consumed TRANSPOSED -- the A operand of O^T = V^T P^T wants 8
// consecutive keys of one channel per lane -- through ds_read_b64_tr_b16: a 16-lane group hands the hardware [4 keys][16
// channels] and every lane gets 4 consecutive keys of its channel (two reads = one 32x32x16 fragment).  Round 3 transposed
// on the way INTO LDS with 2-byte stores (8 per 16 bytes loaded, neighbouring keys in the same dword): bank-serialised, it was
// the bulk of the kernel's 21 us per workgroup (rocprofv3, round 4).  A wave takes 32 queries at a time and computes everything TRANSPOSED so that a lane owns
// one query:   S^T = K Q^T  (A operand = K rows, B operand = the lane's query row, straight from global memory)
//   -> lane (query q, half h) holds the scores of keys {32t + (r&3) + 8(r>>2) + 4h}: row max / sum are in-lane
//      reductions plus ONE exchange with lane ^ 32; P = exp(S - max) is rounded to bf16 in registers;
//   O^T = V^T P^T  (A operand = V^T rows from LDS, B operand = P): the lane needs 8 CONSECUTIVE keys per k-step, it
//      holds keys {0-3, 8-11} (h = 0) or {4-7, 12-15} (h = 1) of every 16 -- one v_permlane32_swap per dword trades
//      the halves (cdna_hip_programming.md T12);
//   -> lane holds 4 consecutive channels of ITS query per accumulator quad: scale by 1 / sum, pack, stage 32 x 64 bf16
//      per wave in LDS, leave as 16-byte row-major stores.
typedef __bf16 uni_bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 uni_lds_bf16x4;
// ds_read_b64_tr_b16 (semantics as in gemm_tn.hip): within a 16-lane group, lane 4 j + c supplies the address of 4 contiguous bf16
// (row j of four, columns 4 c .. 4 c + 3 of a 16-column block); lane i receives column i of rows 0 .. 3
__device__ __forceinline__ u32x2 uni_tr_read(const char* p) {
    const auto v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) uni_lds_bf16x4*)p);
    return __builtin_bit_cast(u32x2, v);
}

__device__ __forceinline__ f32x16 mma_bf16(const u32x4& a, const u32x4& b, f32x16 acc) {
    union { u32x4 u; bf16x8 h; } ua, ub;
    ua.u = a; ub.u = b;
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.h, ub.h, acc, 0, 0, 0);
}

__global__ __launch_bounds__(256, 2) void uni_attn_mfma_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o, int Ttok, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ntile = (Ttok + 31) / 32, Tp = ntile * 32;
    char* sK = smem;                                   // [Tp][128 B], 16-byte chunks swizzled by (row >> 1) & 7
    char* sV = smem + 256 * 128;                       // [Tp][128 B], the same chunk swizzle
    char* sO = smem + 2 * 256 * 128;                   // [4 waves][32 q][128 B]
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int I = H * DH, ldq = 3 * I;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const bf16_t* base = qkv + (size_t)b * Ttok * ldq + h * DH;
    for (int i = tid; i < Tp * 8; i += 256) {          // K rows (zero beyond T), V transposed
        const int r = i >> 3, c = i & 7;
        u32x4 kv = {0, 0, 0, 0}, vv = {0, 0, 0, 0};
        if (r < Ttok) {
            kv = *reinterpret_cast<const u32x4*>(base + (size_t)r * ldq + I + c * 8);
            vv = *reinterpret_cast<const u32x4*>(base + (size_t)r * ldq + 2 * I + c * 8);
        }
        *reinterpret_cast<u32x4*>(sK + r * 128 + ((c ^ ((r >> 1) & 7)) << 4)) = kv;
        *reinterpret_cast<u32x4*>(sV + r * 128 + ((c ^ ((r >> 1) & 7)) << 4)) = vv;
    }
    // transposing-read geometry of this lane: group G = lane >> 4 serves channels (G & 1) * 16 .. + 16 and the keys of k half G >> 1;
    // as a supplier the lane hands over key offset j = (lane >> 2) & 3, channels 4 (lane & 3) .. + 3 of that block
    const int tr_j = (lane >> 2) & 3, tr_c = lane & 3, tr_G = lane >> 4;
    const int tr_chunk0 = (tr_G & 1) * 2 + (tr_c >> 1);          // 16-byte chunk of the row (+ 4 nt), before the swizzle
    const int tr_sub = (tr_c & 1) << 3;                          // byte inside the chunk
    __syncthreads();
    char* myO = sO + wave * 4096;
    for (int qb = wave; qb < ntile; qb += 4) {
        const int q = qb * 32 + l31;
        const bf16_t* qrow = base + (size_t)(q < Ttok ? q : Ttok - 1) * ldq;
        u32x4 qf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const u32x4*>(qrow + ks * 16 + lh * 8);
        f32x16 acc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
            if (t < ntile) {
                const int kr = t * 32 + l31;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    acc[t] = mma_bf16(*reinterpret_cast<const u32x4*>(sK + kr * 128 + (((2 * ks + lh) ^ ((kr >> 1) & 7)) << 4)), qf[ks], acc[t]);
            }
        }
        // softmax over the keys of this lane's query.  The scores stay unscaled: scale > 0, so the maximum commutes with it, and
        // exp(scale (s - max)) is ONE fma into v_exp_f32 (2^x) with c = scale log2(e).  Only the last key tile holds padding keys
        // (-inf there: 2^-inf = 0).  (Round 5: the softmax is VALU time -- ~40 k scores per workgroup -- the matrix pipes wait for.)
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (t < ntile) {
                if (t == ntile - 1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (key >= Ttok) acc[t][r] = -INFINITY;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc[t][r]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float cexp = scale * 1.44269504088896340736f, moff = -mx * cexp;
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (t < ntile) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(acc[t][r], cexp, moff));
                    acc[t][r] = e;
                    sum += e;
                }
            }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        // O^T = V^T P^T
        f32x16 ot[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) ot[nt][e] = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (t < ntile) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {           // 16 keys per MFMA k-step: 32t + 16hf ..
                    const int r0 = 8 * hf;
                    uint32_t x0 = pack_bf16x2(acc[t][r0 + 0], acc[t][r0 + 1]), x1 = pack_bf16x2(acc[t][r0 + 2], acc[t][r0 + 3]);
                    uint32_t y0 = pack_bf16x2(acc[t][r0 + 4], acc[t][r0 + 5]), y1 = pack_bf16x2(acc[t][r0 + 6], acc[t][r0 + 7]);
                    auto s0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
                    auto s1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
                    const u32x4 pf = {(uint32_t)s0[0], (uint32_t)s1[0], (uint32_t)s0[1], (uint32_t)s1[1]};      // keys base + 8*lh .. +7
                    const int key0 = t * 32 + hf * 16 + (tr_G >> 1) * 8 + tr_j;          // this lane's supplied key (first read; + 4 second)
                    const int key1 = key0 + 4;
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const int ch = nt * 4 + tr_chunk0;
                        const u32x2 v0 = uni_tr_read(sV + key0 * 128 + ((ch ^ ((key0 >> 1) & 7)) << 4) + tr_sub);
                        const u32x2 v1 = uni_tr_read(sV + key1 * 128 + ((ch ^ ((key1 >> 1) & 7)) << 4) + tr_sub);
                        ot[nt] = mma_bf16(u32x4{v0[0], v0[1], v1[0], v1[1]}, pf, ot[nt]);     // lane: channel nt*32 + l31, keys kb .. kb + 7
                    }
                }
            }
        // lane = query l31, channels nt*32 + 8g + 4lh .. +3  ->  staging row l31
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = nt * 32 + 8 * g + 4 * lh;
                *reinterpret_cast<u32x2*>(myO + l31 * 128 + d0 * 2) =
                    u32x2{pack_bf16x2(ot[nt][4 * g + 0] * inv, ot[nt][4 * g + 1] * inv), pack_bf16x2(ot[nt][4 * g + 2] * inv, ot[nt][4 * g + 3] * inv)};
            }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = u * 64 + lane, r = idx >> 3, c = idx & 7;
            const int qq = qb * 32 + r;
            const u32x4 v = *reinterpret_cast<const u32x4*>(myO + r * 128 + c * 16);
            if (qq < Ttok) *reinterpret_cast<u32x4*>(o + ((size_t)b * Ttok + qq) * I + h * DH + c * 8) = v;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

struct UniBufs {
    void* col; float* E; float* X; float* X1; void* Xn; void* QKV; void* O; void* Hid; float* cls; size_t bytes;
};

void uni_bufs(const sq_uni_config& c, int dtype, int n, char* base, UniBufs* o) {
    Arena a{base, 0};
    const size_t es = sq_dtype_size(dtype);
    const size_t G = c.img_size / PS, np_ = G * G, T = np_ + 1, D = c.dim;
    o->col = a.take((size_t)n * np_ * KP * es);
    o->E = (float*)a.take((size_t)n * np_ * D * 4);
    o->X = (float*)a.take((size_t)n * T * D * 4);
    o->X1 = (float*)a.take((size_t)n * T * D * 4);
    o->Xn = a.take((size_t)n * T * D * es);
    o->QKV = a.take((size_t)n * T * 3 * D * es);
    o->O = a.take((size_t)n * T * D * es);
    o->Hid = a.take((size_t)n * T * c.mlp_dim * es);
    o->bytes = sq_align_up(a.off, 256);
}

int check_cfg(const sq_uni_config* c) {
    SQ_REQUIRE(c != nullptr, "uni: null config");
    SQ_REQUIRE(c->dim > 0 && c->dim % 64 == 0 && c->dim <= 4096 && c->heads * DH == c->dim, "uni: dim=%d heads=%d (dim must be heads * 64, <= 4096)", c->dim, c->heads);
    SQ_REQUIRE(c->depth >= 1 && c->depth <= SQ_UNI_MAX_DEPTH, "uni: depth=%d out of [1,%d]", c->depth, SQ_UNI_MAX_DEPTH);
    SQ_REQUIRE(c->mlp_dim > 0 && c->mlp_dim % 8 == 0, "uni: mlp_dim=%d", c->mlp_dim);
    SQ_REQUIRE(c->img_size >= PS && c->img_size % PS == 0 && (c->img_size / PS) * (c->img_size / PS) + 1 <= ATT_MAXT,
               "uni: img_size=%d (multiple of 16, at most %d tokens)", c->img_size, ATT_MAXT);
    return SQ_OK;
}

}  // namespace

extern "C" int sq_uni_layout_init(const sq_uni_config* c, sq_uni_layout* out) {
    if (int e = check_cfg(c)) return e;
    SQ_REQUIRE(out != nullptr, "uni: null layout");
    const int64_t D = c->dim, M = c->mlp_dim, G = c->img_size / PS, T = G * G + 1;
    int64_t off = 0;
    auto take = [&](int64_t n) { off = (off + 7) / 8 * 8; const int64_t o = off; off += n; return o; };
    out->patch_w = take(D * KP); out->patch_b = take(D);
    out->cls = take(D); out->pos = take(T * D);
    for (int l = 0; l < SQ_UNI_MAX_DEPTH; ++l) {
        sq_uni_layer_offsets& L = out->layer[l];
        if (l >= c->depth) { L = sq_uni_layer_offsets{-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1}; continue; }
        L.ln1_g = take(D); L.ln1_b = take(D);
        L.qkv_w = take(3 * D * D); L.qkv_b = take(3 * D);
        L.proj_w = take(D * D); L.proj_b = take(D); L.ls1 = take(D);
        L.ln2_g = take(D); L.ln2_b = take(D);
        L.fc1_w = take(M * D); L.fc1_b = take(M);
        L.fc2_w = take(D * M); L.fc2_b = take(D); L.ls2 = take(D);
    }
    out->norm_g = take(D); out->norm_b = take(D);
    out->total = (off + 7) / 8 * 8;
    return SQ_OK;
}

extern "C" size_t sq_uni_workspace_bytes(const sq_uni_config* c, int dtype, int n_patches) {
    if (check_cfg(c) != SQ_OK || n_patches < 1) return 0;
    UniBufs b;
    uni_bufs(*c, dtype, n_patches, nullptr, &b);
    return b.bytes;
}

// params: fp32 flat buffer (sq_uni_layout) -- biases, LayerNorm and embeddings are read from it;
// params_exec: the GEMM weights in the compute dtype with LayerScale folded into proj / fc2 (same layout; fp32 mode may
// pass a folded fp32 copy, bf16 mode a folded bf16 copy).  bias_exec: fp32 copy of the flat buffer whose proj / fc2
// biases are folded likewise (may equal params when every ls == 1).
extern "C" int sq_uni_forward(const sq_uni_config* c, int dtype, const float* params, const void* params_exec, const float* bias_exec,
                              const uint8_t* patches_u8, const float* patches_f32_nchw, int n, float* out, void* workspace,
                              size_t workspace_bytes, sq_stream_t stream_) {
    if (int e = check_cfg(c)) return e;
    hipStream_t s = (hipStream_t)stream_;
    SQ_REQUIRE(dtype == SQ_F32 || dtype == SQ_BF16, "uni_forward: dtype %d", dtype);
    SQ_REQUIRE(params && params_exec && bias_exec && out && workspace, "uni_forward: null pointer");
    SQ_REQUIRE((patches_u8 != nullptr) != (patches_f32_nchw != nullptr), "uni_forward: give exactly one of patches_u8 / patches_f32_nchw");
    SQ_REQUIRE(n >= 1, "uni_forward: n=%d", n);
    {   // every activation is addressed through a 31-bit buffer descriptor: the widest one is [n * tokens, max(3 dim, mlp_dim)]
        const size_t tokens = (size_t)(c->img_size / PS) * (c->img_size / PS) + 1;
        const size_t widest = (size_t)n * tokens * (size_t)(c->mlp_dim > 3 * c->dim ? c->mlp_dim : 3 * c->dim) * sq_dtype_size(dtype);
        SQ_REQUIRE(widest < (1ull << 31), "uni_forward: a launch group of %d patches exceeds the 2 GiB buffer-descriptor limit (ViT-L/16 at 224: <= 1330 patches in bf16, <= 665 in fp32)", n);
    }
    sq_uni_layout lay;
    if (int e = sq_uni_layout_init(c, &lay)) return e;
    UniBufs w;
    uni_bufs(*c, dtype, n, (char*)workspace, &w);
    if (w.bytes > workspace_bytes) {
        sq_set_error("uni_forward: workspace %zu < required %zu", workspace_bytes, w.bytes);
        return SQ_ERR_WORKSPACE;
    }
    const int D = c->dim, H = c->heads, Mh = c->mlp_dim, S = c->img_size, G = S / PS, NP = G * G, T = NP + 1;
    const size_t es = sq_dtype_size(dtype);
    const bool lp = dtype == SQ_BF16;
    SQ_REQUIRE((size_t)n * T * (size_t)(Mh > 3 * D ? Mh : 3 * D) * es < (1ull << 31) && (size_t)n * NP * KP * es < (1ull << 31),
               "uni_forward: %d patches exceed the 2 GiB buffer-descriptor limit of one launch group (use sub-batches)", n);
    auto W = [&](int64_t off) { return (const void*)((const char*)params_exec + (size_t)off * es); };
    auto Wrem = [&](int64_t off) { return (size_t)(lay.total - off) * es; };
    auto Pf = [&](int64_t off) { return params + off; };
    auto Bf = [&](int64_t off) { return bias_exec + off; };
    auto grid_for = [](size_t work) { size_t nb = (work + 255) / 256; return (int)(nb > 65535 ? 65535 : (nb ? nb : 1)); };
    auto ln = [&](const float* x, size_t stride, int64_t g, int64_t b, void* y, int ydt, int R) -> int {
        const bool v8 = D == 1024 && stride % 4 == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)Pf(g) | (uintptr_t)Pf(b)) & 15) == 0;
        if (v8 && ydt == SQ_BF16) hipLaunchKernelGGL((uni_ln8_kernel<bf16_t, 2>), dim3((R + 3) / 4), dim3(256), 0, s, x, stride, Pf(g), Pf(b), (bf16_t*)y, R, 1e-6f);
        else if (v8) hipLaunchKernelGGL((uni_ln8_kernel<float, 2>), dim3((R + 3) / 4), dim3(256), 0, s, x, stride, Pf(g), Pf(b), (float*)y, R, 1e-6f);
        else if (ydt == SQ_BF16) hipLaunchKernelGGL(uni_ln_kernel<bf16_t>, dim3((R + 3) / 4), dim3(256), 0, s, x, stride, Pf(g), Pf(b), (bf16_t*)y, R, D, 1e-6f);
        else hipLaunchKernelGGL(uni_ln_kernel<float>, dim3((R + 3) / 4), dim3(256), 0, s, x, stride, Pf(g), Pf(b), (float*)y, R, D, 1e-6f);
        SQ_LAUNCH_CHECK();
        return SQ_OK;
    };
    const int Mt = n * T;

    // patch embedding: im2col (+ transform) then [n*NP, 768] x [768, D]
    if (lp) hipLaunchKernelGGL(uni_im2col_kernel<bf16_t>, dim3(grid_for((size_t)n * NP * (KP / 8))), dim3(256), 0, s, patches_u8, patches_f32_nchw, (bf16_t*)w.col, n, S);
    else hipLaunchKernelGGL(uni_im2col_kernel<float>, dim3(grid_for((size_t)n * NP * (KP / 8))), dim3(256), 0, s, patches_u8, patches_f32_nchw, (float*)w.col, n, S);
    SQ_LAUNCH_CHECK();
    {
        GemmArgs g; g.A = w.col; g.lda = KP; g.a_bytes = (size_t)n * NP * KP * es;
        g.B = W(lay.patch_w); g.ldb = KP; g.b_bytes = Wrem(lay.patch_w); g.bias = Pf(lay.patch_b);
        g.C = w.E; g.ldc = D; g.M = n * NP; g.N = D; g.K = KP;
        if (int e = sq_launch_gemm(g, dtype, s)) return e;
    }
    hipLaunchKernelGGL(uni_tokens_kernel, dim3(grid_for((size_t)Mt * D / 4)), dim3(256), 0, s, (const float4*)w.E, (const float4*)Pf(lay.cls),
                       (const float4*)Pf(lay.pos), (float4*)w.X, n, T, D / 4);
    SQ_LAUNCH_CHECK();

    const size_t att_lds = ((size_t)T * DH * 2 + (size_t)T * (DH + 1) + 4 * ATT_MAXT) * sizeof(float);
    const size_t att_mfma_lds = 2 * 256 * 128 + 4 * 4096;
    const bool valu_attn = (T + 31) / 32 * 32 > 256;   // the MFMA core's K / V images hold <= 256 keys
    static SqDevOnce attr;       // hipFuncSetAttribute is per device
    if (attr.needed()) {
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)uni_attn_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)att_mfma_lds));
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)uni_attn_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)uni_attn_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr.done();
    }
    float* X = w.X;
    float* X1 = w.X1;
    for (int l = 0; l < c->depth; ++l) {
        const sq_uni_layer_offsets& L = lay.layer[l];
        if (int e = ln(X, (size_t)D, L.ln1_g, L.ln1_b, w.Xn, dtype, Mt)) return e;
        {   // qkv = LN(x) Wqkv^T + b
            GemmArgs g; g.A = w.Xn; g.lda = D; g.a_bytes = (size_t)Mt * D * es;
            g.B = W(L.qkv_w); g.ldb = D; g.b_bytes = Wrem(L.qkv_w); g.bias = Pf(L.qkv_b);
            g.C = w.QKV; g.out_dtype = dtype; g.ldc = 3 * D; g.M = Mt; g.N = 3 * D; g.K = D;
            if (int e = sq_launch_gemm(g, dtype, s)) return e;
        }
        if (lp && !valu_attn) hipLaunchKernelGGL(uni_attn_mfma_kernel, dim3(n * H), dim3(256), att_mfma_lds, s, (const bf16_t*)w.QKV, (bf16_t*)w.O, T, H, 0.125f);
        else if (lp) hipLaunchKernelGGL(uni_attn_kernel<bf16_t>, dim3(n * H), dim3(256), att_lds, s, (const bf16_t*)w.QKV, (bf16_t*)w.O, T, H, 0.125f);
        else hipLaunchKernelGGL(uni_attn_kernel<float>, dim3(n * H), dim3(256), att_lds, s, (const float*)w.QKV, (float*)w.O, T, H, 0.125f);
        SQ_LAUNCH_CHECK();
        // The LAST block: only the class token of its output is read (forward_head with global_pool='token', num_classes=0:
        // compute_features_hdf5.py:125-129 takes feat_model(image) = norm(x)[:, 0]).  Keys and values need every token, everything
        // behind the attention core is per token: the projection, LayerNorm and the MLP run on the n class rows only (row stride
        // T D in O / X) -- 9/12 of the block's product work, 3 % of the network's, with the same per-row arithmetic.
        const bool cls_only = l == c->depth - 1;
        const int R = cls_only ? n : Mt;                          // rows behind the attention core
        const int ldr = cls_only ? T * D : D;                     // their stride in O and X
        {   // x1 = x + ls1 * (o Wp^T + bp)      (gain folded into Wp / bp)
            GemmArgs g; g.A = w.O; g.lda = ldr; g.a_bytes = (size_t)Mt * D * es;
            g.B = W(L.proj_w); g.ldb = D; g.b_bytes = Wrem(L.proj_w); g.bias = Bf(L.proj_b);
            g.res = X; g.ldres = ldr; g.C = X1; g.ldc = D; g.M = R; g.N = D; g.K = D;
            if (int e = sq_launch_gemm(g, dtype, s)) return e;
        }
        if (int e = ln(X1, (size_t)D, L.ln2_g, L.ln2_b, w.Xn, dtype, R)) return e;
        {   // h = GELU(LN(x1) W1^T + b1)
            GemmArgs g; g.A = w.Xn; g.lda = D; g.a_bytes = (size_t)R * D * es;
            g.B = W(L.fc1_w); g.ldb = D; g.b_bytes = Wrem(L.fc1_w); g.bias = Pf(L.fc1_b); g.act = SQ_ACT_GELU;
            g.C = w.Hid; g.out_dtype = dtype; g.ldc = Mh; g.M = R; g.N = Mh; g.K = D;
            if (int e = sq_launch_gemm(g, dtype, s)) return e;
        }
        {   // x = x1 + ls2 * (h W2^T + b2)        (last block: the n class rows, packed at the head of X)
            GemmArgs g; g.A = w.Hid; g.lda = Mh; g.a_bytes = (size_t)R * Mh * es;
            g.B = W(L.fc2_w); g.ldb = Mh; g.b_bytes = Wrem(L.fc2_w); g.bias = Bf(L.fc2_b);
            g.res = X1; g.ldres = D; g.C = X; g.ldc = D; g.M = R; g.N = D; g.K = Mh;
            if (int e = sq_launch_gemm(g, dtype, s)) return e;
        }
    }
    // features = LN(x)[cls] (forward_head with global_pool='token', num_classes=0): the class rows, packed by the last block
    return ln(X, (size_t)D, lay.norm_g, lay.norm_b, out, SQ_F32, n);
}
