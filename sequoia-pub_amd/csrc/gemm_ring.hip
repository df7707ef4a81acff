// 256 x 128 block tile, 8 waves (4 x 2, 64 x 64 per wave), THREE operand stages in LDS -- the NT engine's variant for
// long-K products (3x3 convolutions, K >= 512 1x1 convolutions, the big ViS products).
//
// The 128 x 128 kernel (gemm.hip) keeps two LDS buffers and closes every K-tile with __syncthreads(): the loads of
// tile t+1 have exactly one tile's MFMA time (~0.2 us) to arrive, and hipcc's barrier drains vmcnt to 0, so every
// K-step ends up waiting for a memory round trip that only a second resident block can cover.  Here the ring is
// three tiles deep and the loop is held together by a COUNTED s_waitcnt and a raw s_barrier:
//
//     wait vmcnt(6)        this thread's 6 LDS-DMA loads of tile t have landed; tile t+1's may still be in flight
//     s_barrier            ... everybody's have; and every wave is done reading tile t-1's buffer
//     issue tile t+2       into the buffer tile t-1 occupied
//     MFMAs of tile t
//
// so a load has two tiles of MFMA time (2 x 16 MFMAs x 2 waves per SIMD ~ 2000 clocks) to arrive and never drains the
// queue.  One block per CU (144 KiB of LDS), 2 waves per SIMD.  A 256-row tile also halves the weight traffic per
// output of the 128-row tile (the weights of a 3x3 conv are re-streamed for every M tile).
// Same loader as gemm.hip: buffer_load ... lds, 16 B per lane, 128-byte rows, XOR swizzle on the source address,
// implicit-GEMM taps for convolutions, out-of-range rows / padding as zeros through the buffer descriptor.
// Epilogue: the fp32 tile is staged through the (now idle) ring, 128 KiB, and leaves as 16-byte row-major accesses;
// bias / residual / ReLU / bf16-or-fp32 output take a prefetching fast path, everything else goes through epi_apply.
#include "gemm.h"
#include "gemm_epi.h"

#include <cstdlib>

namespace {

constexpr uint32_t OOB = 0x80000000u;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_base, uint32_t voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_base, 16, voffset, soffset, 0, 0);
}
__device__ __forceinline__ u32x4 lds_read128(const char* p) { return *reinterpret_cast<const u32x4*>(p); }

constexpr int BM = 256, BN = 128, BK = 64;
constexpr int WTM = 2, WTN = 2;                 // 32 x 32 MFMA tiles per wave
constexpr int RA = BM / 64, RB = BN / 64;       // LDS-DMA instructions per thread per tile (4 + 2)
constexpr int STAGE_BYTES = (BM + BN) * 128;    // 48 KiB
constexpr int NSTAGE = 3;
constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES; // 144 KiB (the fp32 epilogue stage needs 128 KiB of it)

template <int EPI, bool CONV>
__global__ __launch_bounds__(512) void gemm_ring_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int nwg = tiles_m * tiles_n;
    int t;
    {   // each XCD (block id % 8) walks a contiguous run of tiles
        const int b = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = b & 7, idx = b >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;
    const int z = blockIdx.z;

    const bf16_t* Ab = reinterpret_cast<const bf16_t*>(p.A) + (long long)z * p.sA;
    const bf16_t* Bb = reinterpret_cast<const bf16_t*>(p.B) + (long long)z * p.sB;
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, (int)(p.a_bytes - (size_t)z * p.sA * 2), 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, (int)(p.b_bytes - (size_t)z * p.sB * 2), 0x00020000);

    const int r0 = tid >> 3;                        // row inside a 64-row round
    const int gc = (tid & 7) ^ ((r0 >> 1) & 7);     // 16-byte chunk of the SOURCE row this lane fetches
    uint32_t a_off[RA], b_off[RB];
    int a_ih0[RA], a_iw0[RA];
    uint32_t a_pix[RA];
    bool a_ok[RA];
#pragma unroll
    for (int j = 0; j < RA; ++j) {
        const int m = m0 + r0 + 64 * j;
        a_ok[j] = m < p.M;
        if constexpr (CONV) {          // implicit GEMM: row m = output pixel (img, oh, ow); taps gathered per K-tile
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
            a_off[j] = a_ok[j] ? ((uint32_t)m * (uint32_t)p.lda + (uint32_t)(gc * 8)) * 2u : OOB;
        }
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        const int n = n0 + r0 + 64 * j;
        b_off[j] = n < p.N ? ((uint32_t)n * (uint32_t)p.ldb + (uint32_t)(gc * 8)) * 2u : OOB;
    }
    auto issue_loads = [&](int kt, int buf) {
        const int k0 = kt * BK;
        const bool k_ok = k0 + gc * 8 < p.K;
        char* sa = smem + buf * STAGE_BYTES + wave * (8 * 128);
        char* sb = sa + BM * 128;
        if constexpr (CONV) {
            const int tap = k0 / p.Cin;                  // a K-tile never straddles taps (Cin % 64 == 0)
            const int cin0 = k0 - tap * p.Cin + gc * 8;
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
            for (int j = 0; j < RA; ++j) {
                const int ih = a_ih0[j] + kh, iw = a_iw0[j] + kw;
                const bool ok = a_ok[j] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                const uint32_t off = ((a_pix[j] + (uint32_t)(ih * p.W + iw)) * (uint32_t)p.Cin + (uint32_t)cin0) * 2u;
                glds16(rsA, sa + j * (64 * 128), ok ? off : OOB, 0);
            }
#pragma unroll
            for (int j = 0; j < RB; ++j) glds16(rsB, sb + j * (64 * 128), k_ok ? b_off[j] + (uint32_t)(k0 * 2) : OOB, 0);
        } else {
            const int soff = k0 * 2;
#pragma unroll
            for (int j = 0; j < RA; ++j) glds16(rsA, sa + j * (64 * 128), k_ok ? a_off[j] : OOB, soff);
#pragma unroll
            for (int j = 0; j < RB; ++j) glds16(rsB, sb + j * (64 * 128), k_ok ? b_off[j] : OOB, soff);
        }
    };

    // epilogue operands that do not depend on the accumulators are requested before the K loop (they are older than
    // every LDS-DMA load, so the counted waits below cover them for free)
    constexpr int BN8 = BN / 8;               // 16 chunks of 8 columns per row
    constexpr int RPI = 512 / BN8;            // 32 rows per epilogue iteration
    constexpr int ITER = BM / RPI;            // 8
    const int e_c8 = tid % BN8, e_rbase = tid / BN8;
    const int e_n = n0 + e_c8 * 8;
    const int e_cnt = min(8, p.N - e_n);
    const bool fast = (EPI == 0 || EPI == 1) && p.vec_epi != 0 && e_cnt == 8 && !p.rowbias && !p.Cpre && !p.gelu_grad_of && !p.ln64_g;
    float bias8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
    if (fast && p.bias) {
        const float* bsrc = p.bias + (long long)z * p.sBias + e_n;
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(bsrc), t1 = *reinterpret_cast<const f32x4*>(bsrc + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { bias8[e] = t0[e]; bias8[4 + e] = t1[e]; }
    }

    f32x16 acc[WTM][WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
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
    auto compute = [&](int buf) {
        const char* st = smem + buf * STAGE_BYTES;
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
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j) {
                    union { u32x4 u; bf16x8 h; } ua, ub;
                    ua.u = fa[s & 1][i]; ub.u = fb[s & 1][j];
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.h, ub.h, acc[i][j], 0, 0, 0);
                }
        }
    };

    const int nk = (p.K + BK - 1) / BK;
    issue_loads(0, 0);
    if (nk > 1) issue_loads(1, 1);
    int cur = 0, nxt2 = 2;                              // ring positions of tile kt and tile kt+2
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // tile kt landed (this thread's part)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                   // everybody's part; and tile kt-1's buffer is free
        if (kt + 2 < nk) issue_loads(kt + 2, nxt2);
        compute(cur);
        cur = cur == 2 ? 0 : cur + 1;
        nxt2 = nxt2 == 2 ? 0 : nxt2 + 1;
    }
    __syncthreads();                                    // all MFMAs read their fragments: the ring becomes the fp32 stage

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
    if (e_cnt <= 0) return;
    if (!fast) {
#pragma unroll 1
        for (int u = 0; u < ITER; ++u) {
            const int row = e_rbase + u * RPI;
            const int m = m0 + row;
            if (m >= p.M) continue;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(stage + row * BN + e_c8 * 8);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(stage + row * BN + e_c8 * 8 + 4);
            float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            epi_apply<EPI, true>(p, z, m, e_n, v, e_cnt, p.vec_epi != 0 && e_cnt == 8);
        }
        return;
    }
    // fast path: residual rows of a group are requested before that group's first store (vmcnt counts stores too)
    const float* res32 = (p.res && p.res_dtype == SQ_F32) ? reinterpret_cast<const float*>(p.res) + (long long)z * p.sRes : nullptr;
    const bf16_t* res16 = (p.res && p.res_dtype == SQ_BF16) ? reinterpret_cast<const bf16_t*>(p.res) + (long long)z * p.sRes : nullptr;
    float* c32 = p.out_dtype == SQ_F32 ? reinterpret_cast<float*>(p.C) + (long long)z * p.sC : nullptr;
    bf16_t* c16p = p.out_dtype == SQ_BF16 ? reinterpret_cast<bf16_t*>(p.C) + (long long)z * p.sC : nullptr;
    constexpr int U = 4;
#pragma unroll
    for (int c0 = 0; c0 < ITER; c0 += U) {
        float aux[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int e = 0; e < 8; ++e) aux[u][e] = 0.f;
            const int m = m0 + e_rbase + (c0 + u) * RPI;
            if (m < p.M) {
                if (res16) {
                    const u32x4 tt = *reinterpret_cast<const u32x4*>(res16 + (long long)m * p.ldres + e_n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { aux[u][2 * e] = __uint_as_float(tt[e] << 16); aux[u][2 * e + 1] = __uint_as_float(tt[e] & 0xffff0000u); }
                } else if (res32) {
                    const float* src = res32 + (long long)m * p.ldres + e_n;
                    const f32x4 t0 = *reinterpret_cast<const f32x4*>(src), t1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { aux[u][e] = t0[e]; aux[u][4 + e] = t1[e]; }
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
            for (int e = 0; e < 8; ++e) v[e] = (p.alpha * v[e] + bias8[e]) + aux[u][e];
            if ((EPI & 1) && p.act == SQ_ACT_GELU) {                     // same erf form as epi_apply<EPI, true>
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = sq_gelu<true>(v[e]);
            } else if (p.act == SQ_ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (c32) {
                float* d = c32 + (long long)m * p.ldc + e_n;
                *reinterpret_cast<f32x4*>(d) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(d + 4) = f32x4{v[4], v[5], v[6], v[7]};
            }
            const u32x4 packed = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
            if (c16p) *reinterpret_cast<u32x4*>(c16p + (long long)m * p.ldc + e_n) = packed;
            if (p.C2) *reinterpret_cast<u32x4*>(p.C2 + (long long)z * p.sC2 + (long long)m * p.ldc2 + e_n) = packed;   // bf16 operand copy
        }
    }
}

}  // namespace

// true when the ring variant takes the product: bf16, long K, N a multiple of the 128-column tile, and enough 256 x 128
// tiles to keep 256 CUs (one block each) busy for at least ~2 rounds
bool sq_gemm_ring_eligible(const GemmArgs& a, int dtype) {
    if (dtype != SQ_BF16 || a.splitk != 1 || a.ln64_g) return false;
    // only the epilogues the prefetching fast path covers (bias, residual, ReLU / GELU, a second bf16 copy): with one
    // block per CU nothing hides a generic epilogue -- row-bias / pre-activation-copy products measured 2x slower here
    if (a.rowbias || a.Cpre || a.gelu_grad_of) return false;
    constexpr int min_tiles = 448, min_k = 512;
    const long long tiles = (long long)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN) * a.batch;
    // strided 1x1 convolutions (the downsample branches) gain from the 256-row tile already at K = 256: the layer-2 one
    // 315 -> 265 us; plain K = 256 products with a residual epilogue lose (143 -> 182 us) and stay on gemm.hip
    const int need_k = a.conv && a.res == nullptr ? (min_k < 256 ? min_k : 256) : min_k;
    return a.K >= need_k && a.N % BN == 0 && tiles >= min_tiles;
}

namespace {
template <int EPI>
int launch_ring(const GemmArgs& a, dim3 grid, hipStream_t stream) {
    static SqDevOnce attr;       // hipFuncSetAttribute is per device
    if (attr.needed()) {
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_ring_kernel<EPI, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_ring_kernel<EPI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr.done();
    }
    if (a.conv) hipLaunchKernelGGL((gemm_ring_kernel<EPI, true>), grid, dim3(512), LDS_BYTES, stream, a);
    else hipLaunchKernelGGL((gemm_ring_kernel<EPI, false>), grid, dim3(512), LDS_BYTES, stream, a);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}
}  // namespace

int sq_launch_gemm_ring(const GemmArgs& a, hipStream_t stream) {
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    const dim3 grid(tiles, 1, a.batch);
    if (a.gelu_grad_of) return launch_ring<2>(a, grid, stream);
    if (a.act == SQ_ACT_GELU) return launch_ring<1>(a, grid, stream);
    return launch_ring<0>(a, grid, stream);
}
