// 256 x 256 block tile on FOUR waves, 128 x 128 per wave (4 x 4 MFMA tiles of 32 x 32, 256 accumulator registers in the
// AGPR half of the file, one wave per SIMD) -- the NT engine's variant for large bf16 products.
//
// Why this shape.  Every other kernel of the engine gives a wave a 64 x 64 (2 x 2 tiles) or 128 x 64 (4 x 2) patch:
// per 16-deep k-step a wave then reads 4 (6) fragments of 1 KiB from LDS for 4 (8) MFMAs.  At the full matrix rate (one
// 32x32x16 MFMA per 32 clocks and SIMD) that is 128 (96) B/clk of fragment reads per CU on top of the LDS-DMA writes of
// the operand stream -- the LDS moves 128 B/clk.  Those kernels are bound by LDS bandwidth, not by L2 or the matrix
// pipes: measured MFMA-busy 0.65 at best (profiles/r03_x3_sq_counters.json), and every variant "saturating at ~10 TB/s of
// staged bytes" (DESIGN round 2) was this limit seen from the other side.  A 128 x 128 patch reads 8 fragments for 16
// MFMAs: 64 B/clk of reads + 32 B/clk of DMA at the full rate -- the first shape with headroom.  It costs the whole
// register file (1 wave per SIMD), so the loop is software-pipelined inside the wave instead of across waves:
//   BK = 32 (64-byte LDS rows, chunk ^= (row >> 2) & 3 on the source address), FOUR stages of 32 KiB, tile kt+3 is issued
//   while tile kt is multiplied (two tiles stay in flight across the barrier: counted s_waitcnt vmcnt(16) + raw
//   s_barrier), and the eight LDS-DMA instructions of a tile are spread between the MFMAs of the k-steps.
// Epilogue: each wave stages its own 32 x 128 slabs through a private 16 KiB LDS region (no block barrier) and writes
// 512-byte (fp32) / 256-byte (bf16) row segments; bias / residual / ReLU / GELU / second bf16 copy as in gemm_ring.hip.
#include "gemm.h"
#include "gemm_epi.h"

#include <cstdlib>
#include <type_traits>

namespace {

constexpr uint32_t OOB = 0x80000000u;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_base, uint32_t voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_base, 16, voffset, soffset, 0, 0);
}
__device__ __forceinline__ u32x4 lds_read128(const char* p) { return *reinterpret_cast<const u32x4*>(p); }

constexpr int BM = 256, BN = 256, BK = 32, ROWB = 64;
constexpr int STAGE_BYTES = (BM + BN) * ROWB;    // 32 KiB
constexpr int NSTAGE = 4;
constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;  // 128 KiB (the epilogue slabs need 64 KiB of it)

// NW = 4: waves 2 x 2, 128 x 128 per wave (one wave per SIMD, accumulators in AGPRs); NW = 8: waves 2 x 4, 128 x 64 per
// wave (two waves per SIMD: the partner's MFMAs cover a wave's LDS-DMA issue and fragment reads)
template <int NW> struct W4Cfg {
    static constexpr int NT = 64 * NW;
    static constexpr int WTM = 4, WTN = NW == 4 ? 4 : 2;
    static constexpr int WCOLS = 32 * WTN;                 // columns per wave
    static constexpr int ROUND = NT / 4;                   // rows one round of LDS-DMA instructions fills
    static constexpr int RA = BM / ROUND, RB = BN / ROUND; // instructions per thread per tile (4 + 4 / 2 + 2)
    static constexpr int LOADS = RA + RB;
    static constexpr int SLAB_BYTES = 32 * WCOLS * 4;      // one wave's 32-row fp32 slab (16 / 8 KiB)
    static constexpr int C8 = WCOLS / 8;                   // 8-column chunks per slab row
    static constexpr int RPP = 64 / C8;                    // slab rows per read-out pass (4 / 8)
    static constexpr int PASSES = 32 / RPP;
};

template <int EPI, bool CONV, int NW>
__global__ __launch_bounds__(64 * NW) void gemm_w4_kernel(const GemmArgs p) {
    using Cfg = W4Cfg<NW>;
    constexpr int WTM = Cfg::WTM, WTN = Cfg::WTN, WCOLS = Cfg::WCOLS, ROUND = Cfg::ROUND, RA = Cfg::RA, RB = Cfg::RB, LOADS = Cfg::LOADS;
    constexpr int SLAB_BYTES = Cfg::SLAB_BYTES, C8 = Cfg::C8, RPP = Cfg::RPP, PASSES = Cfg::PASSES;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = NW == 4 ? wave >> 1 : wave >> 2, wn = NW == 4 ? wave & 1 : wave & 3;

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

    // loader geometry: a wave instruction fills 16 consecutive 64-byte LDS rows; NT threads = one round of NT / 4 rows
    const int r0 = tid >> 2;
    const int gc = (tid & 3) ^ ((r0 >> 2) & 3);     // 16-byte chunk of the SOURCE row this lane fetches
    uint32_t a_off[RA], b_off[RB];
    int a_ih0[RA], a_iw0[RA];
    uint32_t a_pix[RA];
    bool a_ok[RA];
#pragma unroll
    for (int j = 0; j < RA; ++j) {
        const int m = m0 + r0 + ROUND * j;
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
        const int n = n0 + r0 + ROUND * j;
        b_off[j] = n < p.N ? ((uint32_t)n * (uint32_t)p.ldb + (uint32_t)(gc * 8)) * 2u : OOB;
    }
    // one LDS-DMA instruction of tile kt: which = 0..3 the A rounds, 4..7 the B rounds (so the issue can be spread)
    int l_k0 = 0, l_kh = 0, l_kw = 0, l_cin0 = 0;
    bool l_kok = true;
    auto load_setup = [&](int kt) {                // kt may run past the last tile: those loads fetch nothing (zeros into a free stage)
        l_k0 = kt * BK;
        l_kok = l_k0 + gc * 8 < p.K;               // false in a ragged last K-tile and behind it
        if constexpr (CONV) {
            const int tap = l_k0 / p.Cin;          // a K-tile never straddles taps (Cin % 32 == 0)
            l_cin0 = l_k0 - tap * p.Cin + gc * 8;
            l_kh = tap / p.KW; l_kw = tap - l_kh * p.KW;
        }
    };
    auto load_one = [&](int which, int buf) {
        char* sa = smem + buf * STAGE_BYTES + wave * 1024;
        if (which < RA) {
            const int j = which;
            if constexpr (CONV) {
                const int ih = a_ih0[j] + l_kh, iw = a_iw0[j] + l_kw;
                const bool ok = l_kok && a_ok[j] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                const uint32_t off = ((a_pix[j] + (uint32_t)(ih * p.W + iw)) * (uint32_t)p.Cin + (uint32_t)l_cin0) * 2u;
                glds16(rsA, sa + j * (ROUND * ROWB), ok ? off : OOB, 0);
            } else {
                if (p.dbg & 64) {    // ... and K-tile-major activations, timing only
                    const int m = m0 + r0 + ROUND * j;
                    glds16(rsA, sa + j * (ROUND * ROWB), l_kok && m < p.M ? ((uint32_t)m * 32u + (uint32_t)(gc * 8)) * 2u : OOB, (l_k0 >> 5) * p.M * 64);
                } else
                glds16(rsA, sa + j * (ROUND * ROWB), l_kok ? a_off[j] : OOB, l_k0 * 2);
            }
        } else {
            const int j = which - RA;
            if (p.dbg & 32) {        // gemm_probe.py w4t: K-tile-major weights (element (n, k) at ((k / 32) * N + n) * 32 + k % 32), timing only
                const int n = n0 + r0 + ROUND * j;
                glds16(rsB, sa + BM * ROWB + j * (ROUND * ROWB), l_kok && n < p.N ? ((uint32_t)n * 32u + (uint32_t)(gc * 8)) * 2u : OOB, (l_k0 >> 5) * p.N * 64);
            } else
            glds16(rsB, sa + BM * ROWB + j * (ROUND * ROWB), l_kok ? b_off[j] : OOB, l_k0 * 2);
        }
    };

    // epilogue operands that do not depend on the accumulators are requested before the K loop
    const int e_c8 = lane % C8, e_r4 = lane / C8;        // slab read-out: C8 lanes cover a slab row, RPP rows per pass
    const int e_n = n0 + wn * WCOLS + e_c8 * 8;
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
    int fa_off[WTM], fb_off[WTN], f_sw[2];
    {
        // row = wave base (a multiple of 32) + i*32 + l31: (row >> 2) & 3 depends on l31 only
        const int sw = (l31 >> 2) & 3;
#pragma unroll
        for (int s = 0; s < 2; ++s) f_sw[s] = ((2 * s + lh) ^ sw) << 4;
#pragma unroll
        for (int i = 0; i < WTM; ++i) fa_off[i] = (wm * 128 + i * 32 + l31) * ROWB;
#pragma unroll
        for (int j = 0; j < WTN; ++j) fb_off[j] = BM * ROWB + (wn * WCOLS + j * 32 + l31) * ROWB;
    }

    const int nk = (p.K + BK - 1) / BK;
    // prologue: tiles 0, 1, 2.  Every tile slot issues its eight loads, real or empty, so the counted waits are uniform
#pragma unroll
    for (int pt = 0; pt < 3; ++pt) {
        load_setup(pt);
#pragma unroll
        for (int w = 0; w < LOADS; ++w) load_one(w, pt);
    }
    int cur = 0, nxt3 = 3;                              // ring positions of tile kt and tile kt+3
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt landed (this thread's part); tiles kt+1, kt+2 stay in flight
        if constexpr (LOADS == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();                   // everybody's part; and tile kt-1's buffer is free
        load_setup(kt + 3);
        const char* st = smem + cur * STAGE_BYTES;
        u32x4 fa[2][WTM], fb[2][WTN];
#pragma unroll
        for (int i = 0; i < WTM; ++i) fa[0][i] = lds_read128(st + fa_off[i] + f_sw[0]);
#pragma unroll
        for (int j = 0; j < WTN; ++j) fb[0][j] = lds_read128(st + fb_off[j] + f_sw[0]);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s == 0) {
#pragma unroll
                for (int i = 0; i < WTM; ++i) fa[1][i] = lds_read128(st + fa_off[i] + f_sw[1]);
#pragma unroll
                for (int j = 0; j < WTN; ++j) fb[1][j] = lds_read128(st + fb_off[j] + f_sw[1]);
            }
#pragma unroll
            for (int i = 0; i < WTM; ++i) {
                // the LDS-DMA instructions of tile kt+3 are spread over the rows of MFMAs: the issue cost hides in the MFMA shadow
                if constexpr (LOADS == 8) load_one(s * 4 + i, nxt3);
                else if ((i & 1) == 0) load_one(s * 2 + (i >> 1), nxt3);
#pragma unroll
                for (int j = 0; j < WTN; ++j) {
                    union { u32x4 u; bf16x8 h; } ua, ub;
                    ua.u = fa[s][i]; ub.u = fb[s][j];
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.h, ub.h, acc[i][j], 0, 0, 0);
                }
            }
        }
        cur = (cur + 1) & 3;
        nxt3 = (nxt3 + 1) & 3;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing empty loads have written their zeros
    __syncthreads();                                    // all fragment reads done: the ring becomes the epilogue slabs

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* slab = reinterpret_cast<float*>(smem + wave * SLAB_BYTES);
    const float* res32 = (fast && p.res && p.res_dtype == SQ_F32) ? reinterpret_cast<const float*>(p.res) + (long long)z * p.sRes : nullptr;
    const bf16_t* res16 = (fast && p.res && p.res_dtype == SQ_BF16) ? reinterpret_cast<const bf16_t*>(p.res) + (long long)z * p.sRes : nullptr;
    float* c32 = p.out_dtype == SQ_F32 ? reinterpret_cast<float*>(p.C) + (long long)z * p.sC : nullptr;
    bf16_t* c16p = p.out_dtype == SQ_BF16 ? reinterpret_cast<bf16_t*>(p.C) + (long long)z * p.sC : nullptr;
    auto slab_out = [&](auto ic) {                       // compile-time slab index: a run-time one would push the accumulators to scratch
        constexpr int i = decltype(ic)::value;
        const int mrow0 = m0 + wm * 128 + i * 32;
        // residual rows of the slab are requested before the slab is written (their latency hides behind the LDS pass)
        float aux[PASSES][8];
        if (fast) {
#pragma unroll
            for (int u = 0; u < PASSES; ++u) {
#pragma unroll
                for (int e = 0; e < 8; ++e) aux[u][e] = 0.f;
                const int m = mrow0 + u * RPP + e_r4;
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
        }
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                slab[row * WCOLS + j * 32 + l31] = acc[i][j][r];
            }
        // the slab is private to the wave: its own LDS writes are ordered before its reads (lgkmcnt), no barrier
#pragma unroll
        for (int u = 0; u < PASSES; ++u) {
            const int row = u * RPP + e_r4;
            const int m = mrow0 + row;
            if (m >= p.M || e_cnt <= 0) continue;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(slab + row * WCOLS + e_c8 * 8);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(slab + row * WCOLS + e_c8 * 8 + 4);
            float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            if (!fast) {
                epi_apply<EPI, true>(p, z, m, e_n, v, e_cnt, p.vec_epi != 0 && e_cnt == 8);
                continue;
            }
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
    };
    slab_out(std::integral_constant<int, 0>{});
    slab_out(std::integral_constant<int, 1>{});
    slab_out(std::integral_constant<int, 2>{});
    slab_out(std::integral_constant<int, 3>{});
}

}  // namespace

// true when the four-wave 256 x 256 variant takes the product: bf16, K a multiple of the 32-deep tile, enough tiles to
// give (nearly) every CU one, and -- with one wave per SIMD nothing hides a slow epilogue -- only the epilogues its
// prefetching fast path covers
bool sq_gemm_w4_eligible(const GemmArgs& a, int dtype) {
    if (dtype != SQ_BF16 || a.splitk != 1 || a.ln64_g || a.rowbias || a.Cpre || a.gelu_grad_of || !a.vec_epi) return false;
    constexpr int min_tiles = 232, min_k = 512;
    if (a.conv && a.Cin % BK) return false;
    const long long tiles = (long long)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN) * a.batch;
    return a.K % 8 == 0 && a.K >= min_k && a.N % BN == 0 && tiles >= min_tiles;
}

int g_w4_waves = -1;              // sq_dbg_set key 9 (tests / probes): 4 = the one-wave-per-SIMD form, otherwise eight waves
namespace {
template <int EPI, int NW>
int launch_w4(const GemmArgs& a, dim3 grid, hipStream_t stream) {
    static SqDevOnce attr;       // hipFuncSetAttribute is per device
    if (attr.needed()) {
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_w4_kernel<EPI, false, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_w4_kernel<EPI, true, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr.done();
    }
    if (a.conv) hipLaunchKernelGGL((gemm_w4_kernel<EPI, true, NW>), grid, dim3(64 * NW), LDS_BYTES, stream, a);
    else hipLaunchKernelGGL((gemm_w4_kernel<EPI, false, NW>), grid, dim3(64 * NW), LDS_BYTES, stream, a);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}
}  // namespace

int sq_launch_gemm_w4(const GemmArgs& a, hipStream_t stream) {
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    const dim3 grid(tiles, 1, a.batch);
    if (g_w4_waves == 4) return a.act == SQ_ACT_GELU ? launch_w4<1, 4>(a, grid, stream) : launch_w4<0, 4>(a, grid, stream);
    return a.act == SQ_ACT_GELU ? launch_w4<1, 8>(a, grid, stream) : launch_w4<0, 8>(a, grid, stream);
}
