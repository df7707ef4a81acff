// MFMA NT GEMM / implicit-GEMM convolution for gfx950 (MI355X).
//
//   C[M,N] = act(alpha * A[M,K] . B[N,K]^T + bias + rowbias + res) * GELU'(gelu_grad_of)
//
// Tile: 256 threads = 4 waves (2x2); each wave owns WTM x WTN MFMA tiles of 32x32.
// K is walked in tiles of 128 bytes per row (64 bf16 / 32 fp32).  Operand tiles go HBM/L2 -> LDS
// directly (buffer_load ... lds, 16 B per lane, no VGPR round trip): a wave's 64 lanes fill 8
// consecutive 128-B LDS rows, and the XOR swizzle that keeps ds_read_b128 conflict-free
// (16-B chunk ^= (row>>1)&7) is applied on the SOURCE address, the LDS image stays lane-linear.
// Tile t+1 is in flight while the MFMAs of tile t run (two LDS buffers, one barrier per K-tile).
// Loads go through buffer descriptors: rows past M/N, K tails and convolution padding are
// redirected to an out-of-range offset and land in LDS as zeros.
// Epilogue: accumulators are staged through LDS and leave as 16-byte row-major accesses.
// Split-K (deterministic): K-slices write fp32 partials, a second kernel sums them in slice
// order and applies the same epilogue.
//
// fp32 path: v_mfma_f32_32x32x2_f32 (exact fp32 fma chain) -- the parity mode.
// bf16 path: v_mfma_f32_32x32x16_bf16, fp32 accumulate -- the perf mode.
#include "gemm.h"
#include "gemm_epi.h"

#include <cstdio>
#include <cstdlib>

bool sq_conv_halo_eligible(const GemmArgs& a, int dtype);
int sq_launch_conv_halo(const GemmArgs& a, hipStream_t stream);
bool sq_gemm_p8_eligible(const GemmArgs& a, int dtype);
int sq_gemm_p8_shape(const GemmArgs& a, int dtype);
int sq_launch_gemm_p8(const GemmArgs& a, hipStream_t stream);
bool sq_gemm_ring_eligible(const GemmArgs& a, int dtype);
int sq_launch_gemm_ring(const GemmArgs& a, hipStream_t stream);

namespace {

constexpr uint32_t OOB = 0x80000000u;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ u32x4 lds_read128(const char* p) { return *reinterpret_cast<const u32x4*>(p); }

// 16 B per lane, HBM/L2 -> LDS without a VGPR round trip.  LDS destination = lds_base + lane*16
// (lane-linear); the source offset is per lane.  Kept in a non-template helper: the host pass of a
// kernel TEMPLATE that names this builtin directly silently drops the kernel's launch stub.
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_base, uint32_t voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_base, 16, voffset, soffset, 0, 0);
}

template <typename T> struct Mma;

template <> struct Mma<float> {
    // chunk = 4 floats: element e of lane-half h is k = 4*(2s+h)+e; A and B use the same
    // assignment, so the contraction is a permutation of the K-tile.
    static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& acc) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[e]), __uint_as_float(b[e]), acc, 0, 0, 0);
    }
};

template <> struct Mma<bf16_t> {
    static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& acc) {
        union { u32x4 u; bf16x8 h; } ua, ub;
        ua.u = a; ub.u = b;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.h, ub.h, acc, 0, 0, 0);
    }
};

template <typename T, int WTM, int WTN, bool CONV, int EPI>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const GemmArgs p) {
    constexpr int BM = 64 * WTM, BN = 64 * WTN;
    constexpr int EPC = 16 / (int)sizeof(T);   // elements per 16-byte chunk
    constexpr int BK = 8 * EPC;                // K elements per tile (128 B)
    constexpr int RA = BM / 32, RB = BN / 32;  // LDS-DMA instructions per thread per tile
    constexpr int TILE_BYTES = (BM + BN) * 128;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware bijective remap: each XCD (block id % 8) walks a contiguous run of tiles
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int nwg = tiles_m * tiles_n;
    int t;
    {
        const int b = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = b & 7, idx = b >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (t / tiles_n) * BM;
    const int n0 = (t % tiles_n) * BN;
    const int z = blockIdx.z;

    const T* Ab = reinterpret_cast<const T*>(p.A) + (long long)z * p.sA;
    const T* Bb = reinterpret_cast<const T*>(p.B) + (long long)z * p.sB;
    const size_t a_rem = p.a_bytes - (size_t)z * p.sA * sizeof(T);
    const size_t b_rem = p.b_bytes - (size_t)z * p.sB * sizeof(T);
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, (int)a_rem, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, (int)b_rem, 0x00020000);

    const int r0 = tid >> 3;                       // row inside a 32-row group (LDS image: lane-linear)
    const int gc = (tid & 7) ^ ((r0 >> 1) & 7);    // 16-B chunk of the SOURCE row this lane fetches

    // Loop-invariant byte offsets (the K-tile offset travels in the scalar soffset operand).
    uint32_t a_off[RA], b_off[RB];
    int a_ih0[RA], a_iw0[RA];
    uint32_t a_pix[RA];
    bool a_ok[RA];
#pragma unroll
    for (int j = 0; j < RA; ++j) {
        const int m = m0 + r0 + 32 * j;
        a_ok[j] = m < p.M;
        if constexpr (CONV) {
            const int ohw = p.OH * p.OW;
            const int img = m / ohw;
            const int rem = m - img * ohw;
            const int oh = rem / p.OW, ow = rem - oh * p.OW;
            a_ih0[j] = oh * p.stride - p.pad;
            a_iw0[j] = ow * p.stride - p.pad;
            a_pix[j] = (uint32_t)(img * p.H * p.W);
            a_off[j] = 0;
        } else {
            a_ih0[j] = a_iw0[j] = 0;
            a_pix[j] = 0;
            a_off[j] = a_ok[j] ? ((uint32_t)m * (uint32_t)p.lda + (uint32_t)(gc * EPC)) * (uint32_t)sizeof(T) : OOB;
        }
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        const int n = n0 + r0 + 32 * j;
        b_off[j] = n < p.N ? ((uint32_t)n * (uint32_t)p.ldb + (uint32_t)(gc * EPC)) * (uint32_t)sizeof(T) : OOB;
    }

    // Epilogue operands that do not depend on the accumulators are requested NOW, so their HBM latency
    // hides under the K loop: each thread owns one fixed 8-column chunk and ITER rows of the tile.
    constexpr int BN8 = BN / 8;
    constexpr int RPI = 256 / BN8;          // rows covered per epilogue iteration
    constexpr int ITER = BM / RPI;
    const int e_c8 = tid % BN8, e_rbase = tid / BN8;
    const int e_n = n0 + e_c8 * 8;
    const int e_cnt = min(8, p.N - e_n);
    const bool e_vec = p.vec_epi != 0 && e_cnt == 8 && p.splitk == 1;
    const bool pre16 = e_vec && p.res != nullptr && p.res_dtype == SQ_BF16;
#pragma clang diagnostic ignored "-Wunused-variable"
    u32x4 rr16[ITER];
    float bias8[8];
    if (pre16) {
        const bf16_t* r16 = reinterpret_cast<const bf16_t*>(p.res) + (long long)z * p.sRes;
#pragma unroll
        for (int u = 0; u < ITER; ++u) {
            const int m = m0 + e_rbase + u * RPI;
            rr16[u] = m < p.M ? *reinterpret_cast<const u32x4*>(r16 + (long long)m * p.ldres + e_n) : u32x4{0, 0, 0, 0};
        }
    }
    float lng[8], lnb[8];
    if constexpr ((EPI & 4) != 0) {
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.ln64_g + e_n), g1 = *reinterpret_cast<const f32x4*>(p.ln64_g + e_n + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.ln64_b + e_n), b1 = *reinterpret_cast<const f32x4*>(p.ln64_b + e_n + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { lng[e] = g0[e]; lng[4 + e] = g1[e]; lnb[e] = b0[e]; lnb[4 + e] = b1[e]; }
    }
    const bool pre_b = e_vec && p.bias != nullptr;
    if (pre_b) {
        const float* bsrc = p.bias + (long long)z * p.sBias + e_n;
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(bsrc), t1 = *reinterpret_cast<const f32x4*>(bsrc + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { bias8[e] = t0[e]; bias8[4 + e] = t1[e]; }
    }

    // tile kt -> LDS buffer buf, asynchronously (completion: vmcnt, drained by the barrier's fence)
    auto issue_loads = [&](int kt, int buf) {
        const int k0 = kt * BK;
        const bool k_ok = k0 + gc * EPC < p.K;           // only false in a ragged last K-tile
        char* sa = smem + buf * TILE_BYTES + wave * (8 * 128);
        char* sb = sa + BM * 128;
        uint32_t oa[RA], ob[RB];
        int soff;
        if constexpr (CONV) {
            const int tap = k0 / p.Cin;                  // a K-tile never straddles taps (Cin % BK == 0)
            const int cin0 = k0 - tap * p.Cin + gc * EPC;
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
            for (int j = 0; j < RA; ++j) {
                const int ih = a_ih0[j] + kh, iw = a_iw0[j] + kw;
                const bool ok = a_ok[j] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                const uint32_t off = ((a_pix[j] + (uint32_t)(ih * p.W + iw)) * (uint32_t)p.Cin + (uint32_t)cin0) * (uint32_t)sizeof(T);
                oa[j] = ok ? off : OOB;
            }
            soff = 0;
#pragma unroll
            for (int j = 0; j < RB; ++j) ob[j] = k_ok ? b_off[j] + (uint32_t)(k0 * (int)sizeof(T)) : OOB;
        } else {
            soff = k0 * (int)sizeof(T);
#pragma unroll
            for (int j = 0; j < RA; ++j) oa[j] = k_ok ? a_off[j] : OOB;
#pragma unroll
            for (int j = 0; j < RB; ++j) ob[j] = k_ok ? b_off[j] : OOB;
        }
#pragma unroll
        for (int j = 0; j < RA; ++j)
            glds16(rsA, sa + j * (32 * 128), oa[j], soff);
#pragma unroll
        for (int j = 0; j < RB; ++j)
            glds16(rsB, sb + j * (32 * 128), ob[j], soff);
    };

    f32x16 acc[WTM][WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;

    // fragment addresses inside a tile are loop-invariant: row * 128 + ((chunk ^ swz(row)) << 4)
    int fa_off[WTM][4], fb_off[WTN][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int chunk = 2 * s + lh;
#pragma unroll
        for (int i = 0; i < WTM; ++i) {
            const int row = wm * (WTM * 32) + i * 32 + l31;
            fa_off[i][s] = row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
        }
#pragma unroll
        for (int j = 0; j < WTN; ++j) {
            const int row = wn * (WTN * 32) + j * 32 + l31;
            fb_off[j][s] = BM * 128 + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
        }
    }

    // k-step s+1's fragments are fetched from LDS while the MFMAs of k-step s run (two register sets)
    auto compute = [&](int buf) {
        const char* st = smem + buf * TILE_BYTES;
        u32x4 fa[2][WTM], fb[2][WTN];
#pragma unroll
        for (int i = 0; i < WTM; ++i) fa[0][i] = lds_read128(st + fa_off[i][0]);
#pragma unroll
        for (int j = 0; j < WTN; ++j) fb[0][j] = lds_read128(st + fb_off[j][0]);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < 3) {
#pragma unroll
                for (int i = 0; i < WTM; ++i) fa[(s + 1) & 1][i] = lds_read128(st + fa_off[i][s + 1]);
#pragma unroll
                for (int j = 0; j < WTN; ++j) fb[(s + 1) & 1][j] = lds_read128(st + fb_off[j][s + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the prefetch ahead of this k-step's MFMAs
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j) Mma<T>::run(fa[s & 1][i], fb[s & 1][j], acc[i][j]);
        }
    };

    // this block's K-tile range (split-K slices are contiguous runs of K-tiles)
    const int nk_all = (p.K + BK - 1) / BK;
    const int per = (nk_all + p.splitk - 1) / p.splitk;
    const int kt_lo = blockIdx.y * per;
    const int kt_hi = min(nk_all, kt_lo + per);

    if (kt_lo < kt_hi) {
        issue_loads(kt_lo, 0);
        __syncthreads();
        for (int kt = kt_lo; kt < kt_hi; ++kt) {
            const int cur = (kt - kt_lo) & 1;
            if (kt + 1 < kt_hi) issue_loads(kt + 1, cur ^ 1);
            compute(cur);
            __syncthreads();
        }
    }

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* stage = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * (WTM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int col = wn * (WTN * 32) + j * 32 + l31;
                stage[row * BN + col] = acc[i][j][r];
            }
    __syncthreads();
    if (p.dbg & 1) return;

    constexpr int BN4 = BN / 4;
    if (p.splitk > 1) {
        // raw fp32 partial of this K-slice: ws[z][slice][M][N] (N % 4 == 0 enforced by the launcher)
        float* part = p.splitk_ws + ((size_t)z * p.splitk + blockIdx.y) * (size_t)p.M * p.N;
#pragma unroll 1
        for (int idx = tid; idx < BM * BN4; idx += 256) {
            const int row = idx / BN4, c4 = idx - row * BN4;
            const int m = m0 + row, n = n0 + c4 * 4;
            if (m < p.M && n < p.N)
                *reinterpret_cast<f32x4*>(part + (size_t)m * p.N + n) = *reinterpret_cast<const f32x4*>(stage + row * BN + c4 * 4);
        }
        return;
    }
    if (e_cnt <= 0) return;
    if (!e_vec) {           // ragged N / unaligned operands: generic element-wise path (rolled loop)
#pragma unroll 1
        for (int u = 0; u < ITER; ++u) {
            const int row = e_rbase + u * RPI;
            const int m = m0 + row;
            if (m >= p.M) continue;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(stage + row * BN + e_c8 * 8);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(stage + row * BN + e_c8 * 8 + 4);
            float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            epi_apply<EPI, sizeof(T) == 2>(p, z, m, e_n, v, e_cnt, false);
        }
        return;
    }
    // Fast path.  vmcnt counts stores as well as loads, so a global load issued after a store waits for that
    // store's write acknowledgement: bias and the bf16 residual were fetched before the K loop, and the
    // remaining optional operands (fp32 residual, row bias, GELU' source) are fetched for a whole group of
    // U rows BEFORE that group's first store.
    const float* res32 = (p.res && p.res_dtype == SQ_F32) ? reinterpret_cast<const float*>(p.res) + (long long)z * p.sRes : nullptr;
    const float* rowbias = p.rowbias ? p.rowbias + (long long)z * p.sRb : nullptr;
    const float* gg = (p.gelu_grad_of && p.gg_dtype == SQ_F32) ? reinterpret_cast<const float*>(p.gelu_grad_of) + (long long)z * p.sGg : nullptr;
    const bf16_t* gg16 = (p.gelu_grad_of && p.gg_dtype == SQ_BF16) ? reinterpret_cast<const bf16_t*>(p.gelu_grad_of) + (long long)z * p.sGg : nullptr;
    float* c32 = p.out_dtype == SQ_F32 ? reinterpret_cast<float*>(p.C) + (long long)z * p.sC : nullptr;
    bf16_t* c16p = p.out_dtype == SQ_BF16 ? reinterpret_cast<bf16_t*>(p.C) + (long long)z * p.sC : nullptr;
    bf16_t* c2 = p.C2 ? p.C2 + (long long)z * p.sC2 : nullptr;
    float* cpre = (p.Cpre && p.pre_dtype == SQ_F32) ? reinterpret_cast<float*>(p.Cpre) + (long long)z * p.sPre : nullptr;
    bf16_t* cpre16 = (p.Cpre && p.pre_dtype == SQ_BF16) ? reinterpret_cast<bf16_t*>(p.Cpre) + (long long)z * p.sPre : nullptr;
    const bool has_gg = gg != nullptr || gg16 != nullptr;
    constexpr int U = ITER < 4 ? ITER : 4;
#pragma unroll
    for (int c0 = 0; c0 < ITER; c0 += U) {
        float aux[U][8];     // gg mode: the GELU' source values; otherwise fp32 residual + row bias
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) aux[u][e] = 0.f;
        if ((EPI & 2) && has_gg) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int m = m0 + e_rbase + (c0 + u) * RPI;
                if (m < p.M) {
                    if (gg16) {
                        const u32x4 t = *reinterpret_cast<const u32x4*>(gg16 + (long long)m * p.ldgg + e_n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { aux[u][2 * e] = __uint_as_float(t[e] << 16); aux[u][2 * e + 1] = __uint_as_float(t[e] & 0xffff0000u); }
                    } else {
                        const float* src = gg + (long long)m * p.ldgg + e_n;
                        const f32x4 t0 = *reinterpret_cast<const f32x4*>(src), t1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { aux[u][e] = t0[e]; aux[u][4 + e] = t1[e]; }
                    }
                }
            }
        } else {
            if (res32) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int m = m0 + e_rbase + (c0 + u) * RPI;
                    if (m < p.M) {
                        const float* src = res32 + (long long)m * p.ldres + e_n;
                        const f32x4 t0 = *reinterpret_cast<const f32x4*>(src), t1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { aux[u][e] = t0[e]; aux[u][4 + e] = t1[e]; }
                    }
                }
            }
            if (rowbias) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int m = m0 + e_rbase + (c0 + u) * RPI;
                    if (m < p.M) {
                        const float* src = rowbias + (long long)(m / p.rows_per_group) * p.ldrb + e_n;
                        const f32x4 t0 = *reinterpret_cast<const f32x4*>(src), t1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { aux[u][e] += t0[e]; aux[u][4 + e] += t1[e]; }
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = e_rbase + (c0 + u) * RPI;
            const int m = m0 + row;
            if (m >= p.M) continue;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(stage + row * BN + e_c8 * 8);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(stage + row * BN + e_c8 * 8 + 4);
            float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = p.alpha * v[e] + (pre_b ? bias8[e] : 0.f);
            if (pre16) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[2 * e] += __uint_as_float(rr16[c0 + u][e] << 16); v[2 * e + 1] += __uint_as_float(rr16[c0 + u][e] & 0xffff0000u); }
            }
            if (!((EPI & 2) && has_gg)) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += aux[u][e];
            }
            if (cpre) {
                float* d = cpre + (long long)m * p.ldpre + e_n;
                *reinterpret_cast<f32x4*>(d) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(d + 4) = f32x4{v[4], v[5], v[6], v[7]};
            }
            if (cpre16)
                *reinterpret_cast<u32x4*>(cpre16 + (long long)m * p.ldpre + e_n) =
                    u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
            if constexpr ((EPI & 4) != 0) {
                // LayerNorm over the 64-column head group this thread's 8 columns belong to: the group is 8
                // consecutive lanes (two-pass mean / variance as nn.LayerNorm, eps 1e-5)
                float sm = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                sm = group8_sum(sm);
                const float mean = sm * (1.0f / 64.0f);
                float q = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { v[e] -= mean; q += v[e] * v[e]; }
                q = group8_sum(q);
                const float rstd = 1.0f / sqrtf(q * (1.0f / 64.0f) + 1e-5f);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] * rstd * lng[e] + lnb[e];
            }
            if ((EPI & 1) && p.act == SQ_ACT_GELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = sq_gelu<sizeof(T) == 2>(v[e]);
            } else if (p.act == SQ_ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if ((EPI & 2) && has_gg) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= sq_gelu_grad<sizeof(T) == 2>(aux[u][e]);
            }
            if (c32) {
                float* d = c32 + (long long)m * p.ldc + e_n;
                *reinterpret_cast<f32x4*>(d) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(d + 4) = f32x4{v[4], v[5], v[6], v[7]};
            }
            const u32x4 h = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
            if (c16p) *reinterpret_cast<u32x4*>(c16p + (long long)m * p.ldc + e_n) = h;
            if (c2) *reinterpret_cast<u32x4*>(c2 + (long long)m * p.ldc2 + e_n) = h;
        }
    }
}

// sums the K-slice partials in slice order, then the normal epilogue (N % 8 == 0 enforced by the launcher)
template <int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs p) {
    const int N8 = p.N / 8;
    const size_t total = (size_t)p.M * N8;
    const int z = blockIdx.z;
    const float* base = p.splitk_ws + (size_t)z * p.splitk * (size_t)p.M * p.N;
    const bool vec = p.vec_epi != 0;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(idx / N8), n = (int)(idx - (size_t)m * N8) * 8;
        const float* src = base + (size_t)m * p.N + n;
        f32x4 s0 = *reinterpret_cast<const f32x4*>(src), s1 = *reinterpret_cast<const f32x4*>(src + 4);
        for (int k = 1; k < p.splitk; ++k) {
            const float* sk = src + (size_t)k * p.M * p.N;
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(sk), t1 = *reinterpret_cast<const f32x4*>(sk + 4);
            s0[0] += t0[0]; s0[1] += t0[1]; s0[2] += t0[2]; s0[3] += t0[3];
            s1[0] += t1[0]; s1[1] += t1[1]; s1[2] += t1[2]; s1[3] += t1[3];
        }
        float v[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
        epi_apply<EPI>(p, p.ngroup ? 0 : z, m, n, v, 8, vec, nullptr, nullptr, p.ngroup ? sq_group_pick(p.gC, z) : nullptr);     // grouped TN launch: member z has its own result pointer
    }
    if (p.colsum_a) {              // TN products: K-slices of the bias gradient sit behind the C partials, [z][slice][M]
        const float* cs = p.splitk_ws + (size_t)p.batch * p.splitk * (size_t)p.M * p.N + (size_t)z * p.splitk * p.M;
        float* dst = p.ngroup ? sq_group_pick(p.gcs, z) : p.colsum_a;
        for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < p.M; m += gridDim.x * blockDim.x) {
            float s = cs[m];
            for (int k = 1; k < p.splitk; ++k) s += cs[(size_t)k * p.M + m];
            dst[m] = s;
        }
    }
}

int launch_reduce(const GemmArgs& a, hipStream_t stream) {
    size_t nb = ((size_t)a.M * (a.N / 8) + 255) / 256;
    if (nb > 2048) nb = 2048;
    const dim3 grid((int)nb, 1, a.batch), block(256);
    if (a.act == SQ_ACT_GELU || a.gelu_grad_of) hipLaunchKernelGGL(splitk_reduce_kernel<3>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(splitk_reduce_kernel<0>, grid, block, 0, stream, a);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}

template <typename T, int WTM, int WTN, bool CONV>
int launch_epi(const GemmArgs& a, dim3 grid, size_t lds, hipStream_t stream) {
    const dim3 block(256);
    if (a.splitk > 1) hipLaunchKernelGGL((gemm_nt_kernel<T, WTM, WTN, CONV, 0>), grid, block, lds, stream, a);   // partials only
    else if (a.ln64_g) {
        if constexpr (!CONV) hipLaunchKernelGGL((gemm_nt_kernel<T, WTM, WTN, false, 5>), grid, block, lds, stream, a);
    }
    else if (a.gelu_grad_of) hipLaunchKernelGGL((gemm_nt_kernel<T, WTM, WTN, CONV, 2>), grid, block, lds, stream, a);
    else if (a.act == SQ_ACT_GELU) hipLaunchKernelGGL((gemm_nt_kernel<T, WTM, WTN, CONV, 1>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((gemm_nt_kernel<T, WTM, WTN, CONV, 0>), grid, block, lds, stream, a);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}

template <typename T, int WTM, int WTN>
int launch_cfg(const GemmArgs& a, hipStream_t stream) {
    constexpr int BM = 64 * WTM, BN = 64 * WTN;
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    // two operand buffers, or -- when K fits one tile -- just one (never smaller than the fp32 epilogue stage):
    // short-K products are memory-bound and want as many blocks per CU as fit
    const int nk1 = (a.K + (int)(128 / sizeof(T)) - 1) / (int)(128 / sizeof(T));
    size_t lds = 2 * (BM + BN) * 128;
    if (nk1 == 1 && a.splitk == 1) {
        lds = (BM + BN) * 128;
        if (lds < (size_t)BM * BN * 4) lds = (size_t)BM * BN * 4;
    }
    const dim3 grid(tiles, a.splitk, a.batch);
    if (int e = a.conv ? launch_epi<T, WTM, WTN, true>(a, grid, lds, stream) : launch_epi<T, WTM, WTN, false>(a, grid, lds, stream)) return e;
    if (a.splitk > 1) return launch_reduce(a, stream);
    return SQ_OK;
}

int g_force_tile = 0, g_force_split = 0, g_use_ring = 1;        // sq_dbg_set key 6 (tests / probes): 0 turns gemm_ring.hip off
}
int g_dbg = 0;                   // shared with gemm_x3.hip
extern int g_tn_force_split, g_tn_ring;
namespace {

template <typename T>
int launch_t(const GemmArgs& a_in, hipStream_t stream) {
    GemmArgs a = a_in;
    a.dbg |= g_dbg;
    const int epc = 16 / (int)sizeof(T);
    auto blocks = [&](int bm, int bn) { return (long long)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn) * a.batch; };
    // 128x128 whenever both extents allow it (a short grid is filled by split-K below, which measured
    // faster than shrinking the tile); narrow problems take the matching 64-wide tile
    int tile;
    const bool can_split = a.splitk_ws != nullptr && a.K >= 16 * 8 * epc;
    const int nk_tiles = (a.K + 8 * epc - 1) / (8 * epc);
    if (a.N <= 64) tile = blocks(128, 64) >= 256 ? 21 : 11;
    else if (a.M <= 64) tile = blocks(64, 128) >= 256 ? 12 : 11;
    else if (nk_tiles <= 4 && blocks(128, 64) >= 1024) tile = 21;    // short-K (memory-bound) products: 48 KiB LDS -> 3 blocks/CU
    else if (blocks(128, 128) >= 256 || can_split) tile = 22;
    else tile = 11;
    if (g_force_tile) tile = g_force_tile;
    const int bm = tile / 10 * 64, bn = tile % 10 * 64;
    // split-K when the grid cannot fill the chip and K is long enough to be worth slicing
    a.splitk = 1;
    const int nk = (a.K + 8 * epc - 1) / (8 * epc);
    const long long nb = blocks(bm, bn);
    if (a.splitk_ws && a.N % 8 == 0 && nb < 256 && nk >= 4) {
        long long s = (512 + nb - 1) / nb;
        if (s > nk / 2) s = nk / 2;
        if (s > 32) s = 32;
        while (s > 1 && (size_t)s * a.M * a.N * a.batch * sizeof(float) > a.splitk_ws_bytes) --s;
        if (s > 1) a.splitk = (int)s;
    }
    if (g_force_split > 0 && a.splitk_ws && a.N % 8 == 0) a.splitk = g_force_split;
    if (g_force_tile >= 33 && g_force_split <= 0) a.splitk = 1;      // a forced special kernel (probes): its tile code is not a 64-multiple pair, the split heuristic above does not apply
    if constexpr (sizeof(T) == 2) {
        // long-K products with enough 256 x 128 tiles: three-stage ring kernel (gemm_ring.hip); sq_dbg_set(6, 0) turns it off
        // 3x3 / stride-1 convolutions with enough 256 x 128 tiles: input tile resident in LDS (conv_halo.hip)
        if (g_force_tile == 0 && a.splitk == 1 && sq_conv_halo_eligible(a, SQ_BF16)) return sq_launch_conv_halo(a, stream);
        // large plain products: 256 x 256 x 64 tile, eight phases per pair of K-tiles (gemm_p8.hip); tile 88 forces it
        if (a.splitk == 1 && a.N % 8 == 0 && !a.conv && (g_force_tile == 88 || (g_force_tile == 0 && sq_gemm_p8_eligible(a, SQ_BF16)))) return sq_launch_gemm_p8(a, stream);
        // (gemm_w4.hip -- 256 x 256 tile, four 32-deep stages -- was the third engine here until round 6: after gemm_p8.hip it still took the bf16
        // layer-4 3x3 and one strided 1x1; without it the ring kernel takes both: 218 -> 205 us and 248 -> 291 us, bf16 pipeline 80.28 vs 80.27
        // slides/s, profiles/r06_w4_ab.txt -- deleted)
        if (a.splitk == 1 && (g_force_tile == 33 || (g_force_tile == 0 && g_use_ring && sq_gemm_ring_eligible(a, SQ_BF16))) && a.N % 8 == 0)
            return sq_launch_gemm_ring(a, stream);
    }
    if (tile == 22) return launch_cfg<T, 2, 2>(a, stream);
    if (tile == 21) return launch_cfg<T, 2, 1>(a, stream);
    if (tile == 12) return launch_cfg<T, 1, 2>(a, stream);
    return launch_cfg<T, 1, 1>(a, stream);
}

}  // namespace

int sq_launch_splitk_reduce(const GemmArgs& a, hipStream_t stream) { return launch_reduce(a, stream); }

int g_x3_small_max_k = -1, g_x3_halo = -1;
extern int g_p8_sched, g_p8_group_m, g_p8_bn, g_p8_on;
extern "C" int sq_dbg_set(int key, int value) {
    if (key == 0) g_force_tile = value;
    else if (key == 1) g_dbg = value;
    else if (key == 2) g_force_split = value;
    else if (key == 6) g_use_ring = value;
    else if (key == 4) g_tn_force_split = value;
    else if (key == 15) g_tn_ring = value;            // gemm_tn.hip: ring form on / off
    else if (key == 7) g_x3_small_max_k = value;
    else if (key == 14) g_p8_on = value;             // gemm_p8.hip on / off (-1: environment)
    else if (key == 13) g_p8_bn = value;             // gemm_p8.hip: forced tile width (128 / 256)
    else if (key == 11) g_p8_group_m = value;        // gemm_p8.hip: tile rows per group of the tile walk
    else if (key == 10) g_p8_sched = value;          // gemm_p8.hip: schedule variant
    else if (key == 8) g_x3_halo = value;            // split-mode 3x3: 0 = implicit GEMM only     // split-mode product: K up to this takes the 128-row shape (-1 = default / environment)
    else return SQ_ERR_ARG;
    return SQ_OK;
}

// 16-byte epilogue accesses need every leading dimension / base / batch stride aligned
int sq_gemm_vec_epi(const GemmArgs& a) {
    auto al = [](const void* ptr, int ld, long long st, int elem) {
        return ptr == nullptr || (((uintptr_t)ptr % 16) == 0 && (ld * elem) % 16 == 0 && ((st * elem) % 16) == 0);
    };
    auto al8 = [&](const void* ptr, int ld, long long st) { return al(ptr, ld, st, 2); };   // bf16 rows: 16-byte accesses
    bool ok = al(a.bias, 4, a.sBias, 4) && al(a.rowbias, a.ldrb, a.sRb, 4) && al(a.Cpre, a.ldpre, a.sPre, a.pre_dtype == SQ_F32 ? 4 : 2) &&
              al(a.gelu_grad_of, a.ldgg, a.sGg, a.gg_dtype == SQ_F32 ? 4 : 2) && al8(a.C2, a.ldc2, a.sC2);
    ok = ok && (a.out_dtype == SQ_F32 ? al(a.C, a.ldc, a.sC, 4) : al8(a.C, a.ldc, a.sC));
    ok = ok && (a.res_dtype == SQ_F32 ? al(a.res, a.ldres, a.sRes, 4) : al8(a.res, a.ldres, a.sRes));
    return ok ? 1 : 0;
}

// Would sq_launch_gemm hand this bf16 product to gemm_p8.hip's 256 x 256 kernel?  (vis.hip asks before it sets GemmArgs::comb_w:
// only that kernel has the combiner epilogue.)
bool sq_gemm_takes_p8_256(const GemmArgs& a) {
    GemmArgs av = a;
    av.vec_epi = sq_gemm_vec_epi(a);
    av.splitk = 1;
    return g_force_tile == 0 && !a.conv && a.N % 8 == 0 && sq_gemm_p8_shape(av, SQ_BF16) == 256;
}

int sq_launch_gemm(const GemmArgs& a, int dtype, hipStream_t stream) {
    const int epc = dtype == SQ_BF16 ? 8 : 4;
    SQ_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.batch > 0, "gemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
    SQ_REQUIRE(!a.colsum_a && !a.ngroup, "gemm: colsum_a / grouped launches are TN-product features");
    SQ_REQUIRE(a.K % epc == 0 && a.ldb % epc == 0, "gemm: K=%d / ldb=%d must be multiples of %d", a.K, a.ldb, epc);
    SQ_REQUIRE(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.B & 15) == 0, "gemm: A/B must be 16-byte aligned");
    SQ_REQUIRE(a.a_bytes > 0 && a.a_bytes < (1ull << 31) && a.b_bytes > 0 && a.b_bytes < (1ull << 31),
               "gemm: operand extents must be in (0, 2 GiB): %zu %zu", a.a_bytes, a.b_bytes);
    if (a.conv) {
        SQ_REQUIRE(a.Cin % (8 * epc) == 0, "conv: Cin=%d must be a multiple of the K-tile (%d)", a.Cin, 8 * epc);
    } else {
        SQ_REQUIRE(a.lda % epc == 0, "gemm: lda=%d must be a multiple of %d", a.lda, epc);
    }
    if (dtype != SQ_BF16 && dtype != SQ_F32) {
        sq_set_error("gemm: unknown dtype %d", dtype);
        return SQ_ERR_ARG;
    }
    GemmArgs av = a;
    av.vec_epi = sq_gemm_vec_epi(a);
    int prof = -1;
    if (sq_prof_on()) {
        // algorithmic work of this launch: 2*M*N*K flops; operands read once + output written once
        const double es = dtype == SQ_BF16 ? 2.0 : 4.0;
        const double flops = 2.0 * a.M * (double)a.N * a.K * a.batch;
        const double a_elems = a.conv ? (double)a.M / (a.OH * a.OW) * a.H * a.W * a.Cin : (double)a.M * a.K;
        // every operand read once, every output written once (residual / GELU' source / extra copies included)
        double mn_bytes = a.out_dtype == SQ_BF16 ? 2.0 : 4.0;
        if (a.res) mn_bytes += a.res_dtype == SQ_BF16 ? 2.0 : 4.0;
        if (a.gelu_grad_of) mn_bytes += a.gg_dtype == SQ_F32 ? 4.0 : 2.0;
        if (a.Cpre) mn_bytes += a.pre_dtype == SQ_F32 ? 4.0 : 2.0;
        if (a.C2) mn_bytes += 2.0;
        const double bytes = (a_elems * es + (double)a.N * a.K * es) * a.batch + (double)a.M * a.N * a.batch * mn_bytes;
        char name[96];
        snprintf(name, sizeof(name), "%s_%s_M%d_N%d_K%d_b%d", a.conv ? "conv" : "gemm", dtype == SQ_BF16 ? "bf16" : "f32",
                 a.M, a.N, a.K, a.batch);
        prof = sq_prof_begin(name, flops, bytes, stream);
    }
    if (av.ln64_g) {
        SQ_REQUIRE(av.ln64_b && !av.conv && av.N % 64 == 0 && av.vec_epi && !av.gelu_grad_of && ((uintptr_t)av.ln64_g & 15) == 0 &&
                   ((uintptr_t)av.ln64_b & 15) == 0,
                   "gemm: the fused LayerNorm(64) epilogue needs N %% 64 == 0 (N=%d), 16-byte aligned operands and a plain A", av.N);
        av.splitk_ws = nullptr;            // the whole row group must be in one block: no K-slices
    }
    int rc;
    if (av.comb_w) {                       // the ViS combiner epilogue lives in gemm_p8.hip only: the caller has asked sq_gemm_takes_p8_256
        SQ_REQUIRE(dtype == SQ_BF16 && sq_gemm_takes_p8_256(a), "gemm: the combiner epilogue (comb_w) needs a bf16 product gemm_p8.hip's 256 x 256 kernel takes");
        av.splitk = 1;
        rc = sq_launch_gemm_p8(av, stream);
    } else {
        rc = dtype == SQ_BF16 ? launch_t<bf16_t>(av, stream) : launch_t<float>(av, stream);
    }
    if (prof >= 0) sq_prof_end(prof, stream);
    return rc;
}
