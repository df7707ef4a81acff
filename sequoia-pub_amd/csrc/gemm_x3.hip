// Split-bf16 ("bf16x3") NT GEMM / implicit-GEMM convolution: the parity-grade fast mode.
//
// Every fp32 value v travels as TWO bf16 planes, hi = bf16(v) and lo = bf16(v - hi) (|v - hi - lo| <= 2^-17 |v|), and a
// product is three bf16 MFMAs into one fp32 accumulator:
//
//     a.b  ~=  a_hi.b_hi + a_hi.b_lo + a_lo.b_hi          (the dropped a_lo.b_lo term is <= 2^-16 of the product)
//
// bf16 x bf16 products are exact in fp32, so the only errors are the 2^-17 operand representation and the fp32
// accumulation: ~1e-5 on ResNet-50 features where plain bf16 gives 4e-3 and exact fp32 MFMA 5e-7 -- at 3/16 of the
// bf16 matrix rate instead of the 1/16 of v_mfma_f32_32x32x2_f32 (effective ceiling 833 TFLOP/s against 157).
// The reference is fp32 end to end (src/resnet.py:155-170, no autocast): this is the mode that keeps its tolerance.
//
// Kernel = the three-stage ring of gemm_ring.hip with FOUR operand sub-tiles per stage (A_hi, A_lo, B_hi, B_lo) and
// three MFMAs per fragment pair: 256 x (64 | 128) block tile, 8 waves (4 x 2), BK = 32 (64-byte LDS rows, 16-byte
// chunk ^= (row >> 2) & 3 on the SOURCE address keeps ds_read_b128 conflict-free), counted s_waitcnt + raw s_barrier.
// Per staged byte a stage carries 1.5x the MFMA work of the plain bf16 ring (both planes are staged once, used twice /
// once), which is what the load-path-bound shapes of this network want.
// Epilogue: acc -> LDS (fp32 tile) -> bias + residual (hi + lo) + ReLU -> split -> two 16-byte stores per 8 columns.
#include "gemm.h"
#include "x3_fmt.h"

#include <cstdio>
#include <cstdlib>

extern int g_x3_small_max_k;     // sq_dbg_set key 7 (tests: force one block shape)
bool sq_conv_halo_x3_eligible(const GemmArgs& a);     // conv_halo_x3.hip
int sq_launch_conv_halo_x3(const GemmArgs& a, hipStream_t stream);
extern int g_x3_halo;            // sq_dbg_set key 8: 0 = never take the halo-staged 3x3 kernel (tests), -1 = default
extern int g_dbg;                // sq_dbg_set key 1: ablation switches (tools/x3_probe.py) -- 1 no stores, 2 no global loads after the prologue, 4 no MFMA, 8 no fragment reads

namespace {

constexpr uint32_t OOB = 0x80000000u;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_base, uint32_t voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_base, 16, voffset, soffset, 0, 0);
}
__device__ __forceinline__ u32x4 lds_read128(const char* p) { return *reinterpret_cast<const u32x4*>(p); }
// 16-byte global accesses.  (A streaming (nt) policy on the residual reads and plane stores -- activations written once and read once
// by the next launch -- was an experiment switch until round 5: pipeline 34.1 / 34.4 slides/s without, 34.1 / 34.1 with; only
// chain_x3w.hip's 256-channel form gains from it and carries it unconditionally.)
__device__ __forceinline__ u32x4 ld16(const bf16_t* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st16(bf16_t* p, const u32x4& v) { *reinterpret_cast<u32x4*>(p) = v; }

constexpr int BK = 32, ROWB = 64;                 // 64-byte LDS rows (32 bf16)
constexpr int WTM = 2;

// Two block shapes.  BM = 256: 8 waves, three stages, one block per CU -- the long-K shape (3x3 convolutions, K >= 512).
// BM = 128: 4 waves, two stages (64 KiB), TWO blocks per CU -- the short-K shape: with K = 64 ... 256 a tile is two to eight
// K-steps and then an epilogue that moves as many bytes as the main loop staged (identity in, hi / lo planes out); a
// second resident block runs its main loop under it.
template <int BM_, int WTN> struct X3Cfg {
    static constexpr int BM = BM_;
    static constexpr int NT = BM_ * 2;                                 // 64 x (32 WTN) outputs per wave
    static constexpr int BN = 64 * WTN;
    static constexpr int A_BYTES = BM * ROWB;                          // per plane per stage
    static constexpr int B_BYTES = BN * ROWB;
    static constexpr int STAGE_BYTES = 2 * A_BYTES + 2 * B_BYTES;     // BM 256: 40 / 48 KiB; BM 128: 24 / 32 KiB
    static constexpr int NSTAGE = BM_ == 256 ? 3 : 2;
    static constexpr int EPI_BYTES = BM * BN * 4;                      // fp32 tile staged through the idle ring
    static constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES > EPI_BYTES ? NSTAGE * STAGE_BYTES : EPI_BYTES;
    static constexpr int ROUND = NT / 4;                               // rows one round of LDS-DMA instructions fills
    static constexpr bool SPLIT_B = BN < ROUND;                        // BM 256, BN 64: rows 0-63 of the round = hi plane, 64-127 = lo plane
    static constexpr int RB = SPLIT_B ? 1 : 2 * (BN / ROUND);          // B instructions per thread per stage
    static constexpr int LOADS = 4 + RB;                               // LDS-DMA instructions per thread per stage
};

template <int BM_, int WTN, bool CONV, bool F16, bool PP, bool DUAL = false>
__global__ __launch_bounds__(BM_ * 2, BM_ == 256 ? 1 : 2) void gemm_x3_kernel(const GemmArgs p) {
    static_assert(!DUAL || (BM_ == 128 && !CONV && !PP), "the dual form lives in the 128-row two-stage shape");
    using Cfg = X3Cfg<BM_, WTN>;
    using Fmt = X3Fmt<F16>;
    constexpr int BM = Cfg::BM, NT = Cfg::NT, A_BYTES = Cfg::A_BYTES, ROUND = Cfg::ROUND;
    constexpr int BN = Cfg::BN, B_BYTES = Cfg::B_BYTES, STAGE_BYTES = Cfg::STAGE_BYTES, NSTAGE = Cfg::NSTAGE;
    static_assert(!PP || BM_ == 256, "the ping-pong schedule needs the 8-wave block");
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
    // (round 6: co-resident tiles of the dual form as 8 x 8 squares instead of 4 x 16 strips was measured -- 424 -> 418 us on the
    // N = 2048 launch, nothing on the others, profiles/r06_dual_walk_ab.txt -- and not kept)
    const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;

    const bf16_t* Ah = reinterpret_cast<const bf16_t*>(p.A);
    const bf16_t* Bh = reinterpret_cast<const bf16_t*>(p.B);
    const auto rsAh = __builtin_amdgcn_make_buffer_rsrc((void*)Ah, 0, (int)p.a_bytes, 0x00020000);
    const auto rsAl = __builtin_amdgcn_make_buffer_rsrc((void*)(Ah + p.plA), 0, (int)p.a_bytes, 0x00020000);
    const auto rsBh = __builtin_amdgcn_make_buffer_rsrc((void*)Bh, 0, (int)p.b_bytes, 0x00020000);
    const auto rsBl = __builtin_amdgcn_make_buffer_rsrc((void*)(Bh + p.plB), 0, (int)p.b_bytes, 0x00020000);

    // loader geometry: a wave instruction fills 16 consecutive 64-byte LDS rows; NT threads = one round of NT / 4 rows
    const int r0 = tid >> 2;                            // row inside a round
    const int gc = (tid & 3) ^ ((r0 >> 2) & 3);         // 16-byte chunk of the SOURCE row this lane fetches
    uint32_t a_off[2];
    int a_ih0[2], a_iw0[2];
    uint32_t a_pix[2];
    bool a_ok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
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
    // B rows: BN / ROUND rounds per plane; SPLIT_B (8 waves, BN = 64): rows 0-63 of the single round are the hi plane, 64-127 the lo plane
    constexpr int RBP = Cfg::SPLIT_B ? 1 : BN / ROUND;  // rounds per plane
    uint32_t b_off[RBP];
    const bool b_lo_half = Cfg::SPLIT_B && r0 >= 64;    // wave-uniform (waves 4-7)
#pragma unroll
    for (int j = 0; j < RBP; ++j) {
        const int n = n0 + (Cfg::SPLIT_B ? (r0 & 63) : r0 + ROUND * j);
        b_off[j] = n >= p.N ? OOB : p.b_tiled ? ((uint32_t)n * 32u + (uint32_t)(gc * 8)) * 2u : ((uint32_t)n * (uint32_t)p.ldb + (uint32_t)(gc * 8)) * 2u;
    }
    auto issue_loads = [&](int kt, int buf) {
        const int k0 = kt * BK;
        const bool k_ok = k0 + gc * 8 < p.K;            // only false in a ragged last K-tile
        char* sa = smem + buf * STAGE_BYTES + wave * 1024;
        char* sb = sa + 2 * A_BYTES;
        if constexpr (CONV) {
            const int tap = k0 / p.Cin;                  // a K-tile never straddles taps (Cin % 32 == 0)
            const int cin0 = k0 - tap * p.Cin + gc * 8;
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ih = a_ih0[j] + kh, iw = a_iw0[j] + kw;
                const bool ok = a_ok[j] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                const uint32_t off = ok ? ((a_pix[j] + (uint32_t)(ih * p.W + iw)) * (uint32_t)p.Cin + (uint32_t)cin0) * 2u : OOB;
                glds16(rsAh, sa + j * (ROUND * ROWB), off, 0);
                glds16(rsAl, sa + A_BYTES + j * (ROUND * ROWB), off, 0);
            }
#pragma unroll
            for (int j = 0; j < RBP; ++j) {
                const uint32_t ob = k_ok ? b_off[j] + (p.b_tiled ? (uint32_t)kt * (uint32_t)p.N * 64u : (uint32_t)(k0 * 2)) : OOB;
                if constexpr (Cfg::SPLIT_B) {
                    if (b_lo_half) glds16(rsBl, sb, ob, 0); else glds16(rsBh, sb, ob, 0);
                } else {
                    glds16(rsBh, sb + j * (ROUND * ROWB), ob, 0);
                    glds16(rsBl, sb + B_BYTES + j * (ROUND * ROWB), ob, 0);
                }
            }
        } else {
            const int soff = (p.dbg & 16) ? k0 * 4 : k0 * 2;     // dbg 16 (x3_probe.py): hi / lo interleaved per 32-element block
            const int soffb = p.b_tiled ? kt * p.N * 64 : soff;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint32_t off = k_ok ? a_off[j] : OOB;
                glds16(rsAh, sa + j * (ROUND * ROWB), off, soff);
                glds16(rsAl, sa + A_BYTES + j * (ROUND * ROWB), off, soff);
            }
#pragma unroll
            for (int j = 0; j < RBP; ++j) {
                const uint32_t ob = k_ok ? b_off[j] : OOB;
                if constexpr (Cfg::SPLIT_B) {
                    if (b_lo_half) glds16(rsBl, sb, ob, soffb); else glds16(rsBh, sb, ob, soffb);
                } else {
                    glds16(rsBh, sb + j * (ROUND * ROWB), ob, soffb);
                    glds16(rsBl, sb + B_BYTES + j * (ROUND * ROWB), ob, soffb);
                }
            }
        }
    };

    // epilogue operands that do not depend on the accumulators are requested before the K loop
    constexpr int BN8 = BN / 8;
    constexpr int RPI = NT / BN8;             // rows per epilogue iteration
    constexpr int ITER = BM / RPI;            // 8 (BN = 128) / 4 (BN = 64)
    const int e_c8 = tid % BN8, e_rbase = tid / BN8;
    const int e_n = n0 + e_c8 * 8;
    const bool e_live = e_n < p.N;            // N % 8 == 0 (launcher)
    float bias8[8], scale8[8];                // per-column scale: undoes the power-of-two pre-scaling of the weight rows (fp16 planes)
#pragma unroll
    for (int e = 0; e < 8; ++e) { bias8[e] = 0.f; scale8[e] = p.alpha; }
    if (e_live && p.bias) {
        const float* bsrc = p.bias + e_n;
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(bsrc), t1 = *reinterpret_cast<const f32x4*>(bsrc + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { bias8[e] = t0[e]; bias8[4 + e] = t1[e]; }
    }
    if (e_live && p.colscale) {
        const float* ssrc = p.colscale + e_n;
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(ssrc), t1 = *reinterpret_cast<const f32x4*>(ssrc + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { scale8[e] = p.alpha * t0[e]; scale8[4 + e] = p.alpha * t1[e]; }
    }

    // 128-row shape: the identity rows of the epilogue (HBM, the launch's largest read for an expand 1x1) are requested
    // when the last K-tile's loads have been waited for, not when the epilogue needs them
    const bool res_pre = BM_ == 128 && p.res != nullptr && e_live && !(p.dbg & 64);
    constexpr int NPRE = BM_ == 128 ? ITER / 2 : 1;      // the first half of the rows (all of them spills)
    u32x4 rpre_h[NPRE], rpre_l[NPRE];
    auto prefetch_res = [&]() {
        if constexpr (BM_ == 128) {
            const bf16_t* rs = reinterpret_cast<const bf16_t*>(p.res);
#pragma unroll
            for (int it = 0; it < NPRE; ++it) {
                const int m = min(m0 + e_rbase + it * RPI, p.M - 1);      // rows past M repeat the last one (never stored)
                rpre_h[it] = ld16(rs + (long long)m * p.ldres + e_n);
                rpre_l[it] = ld16(rs + p.plRes + (long long)m * p.ldres + e_n);
            }
        }
    };

    f32x16 acc[WTM][WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
    int fa_off[WTM][2], fb_off[WTN][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int chunk = 2 * s + lh;
#pragma unroll
        for (int i = 0; i < WTM; ++i) {
            const int row = wm * (WTM * 32) + i * 32 + l31;
            fa_off[i][s] = row * ROWB + ((chunk ^ ((row >> 2) & 3)) << 4);
        }
#pragma unroll
        for (int j = 0; j < WTN; ++j) {
            const int row = wn * (WTN * 32) + j * 32 + l31;
            fb_off[j][s] = 2 * A_BYTES + row * ROWB + ((chunk ^ ((row >> 2) & 3)) << 4);
        }
    }
    auto compute = [&](int buf, f32x16 (&acc)[WTM][WTN]) {
        const char* st = smem + buf * STAGE_BYTES;
        u32x4 ah[2][WTM], al[2][WTM], bh[2][WTN], bl[2][WTN];
#pragma unroll
        for (int i = 0; i < WTM; ++i) { ah[0][i] = lds_read128(st + fa_off[i][0]); al[0][i] = lds_read128(st + fa_off[i][0] + A_BYTES); }
#pragma unroll
        for (int j = 0; j < WTN; ++j) { bh[0][j] = lds_read128(st + fb_off[j][0]); bl[0][j] = lds_read128(st + fb_off[j][0] + B_BYTES); }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s == 0) {
#pragma unroll
                for (int i = 0; i < WTM; ++i) { ah[1][i] = lds_read128(st + fa_off[i][1]); al[1][i] = lds_read128(st + fa_off[i][1] + A_BYTES); }
#pragma unroll
                for (int j = 0; j < WTN; ++j) { bh[1][j] = lds_read128(st + fb_off[j][1]); bl[1][j] = lds_read128(st + fb_off[j][1] + B_BYTES); }
            }
            __builtin_amdgcn_sched_barrier(0);
            // the two correction terms first, the leading term last; consecutive MFMAs touch different accumulators
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j) Fmt::mma(al[s][i], bh[s][j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j) Fmt::mma(ah[s][i], bl[s][j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j) Fmt::mma(ah[s][i], bh[s][j], acc[i][j]);
        }
    };

    const int nk = (p.K + BK - 1) / BK;
    issue_loads(0, 0);
    if (NSTAGE == 3 && nk > 1) issue_loads(1, 1);
    if constexpr (PP) {
        // Ping-pong schedule: the two waves of a SIMD (w and w + 4) belong to different groups that run half a K-tile apart.
        // A wave alternates a LOAD phase (all 16 fragment reads of tile t into registers, its share of tile t+2's LDS-DMA,
        // the counted wait for its share of tile t+1) and a MATRIX phase (the 24 MFMAs of tile t, registers only); one
        // block-wide barrier per phase keeps group 0 in its matrix phase exactly while group 1 loads and vice versa, so
        // the SIMD's matrix pipe has one wave feeding it at all times and address arithmetic, DMA issue and LDS latency
        // sit under the partner's MFMAs.  Hazards: tile t is read in phases 2t (group 0) and 2t+1 (group 1); its buffer
        // is overwritten by tile t+3, issued in phases 2t+2 / 2t+3 -- behind the barrier that ends phase 2t+1; a wave's
        // share of tile t+1 is waited for before the barrier that ends its load phase of tile t, i.e. two barriers before
        // the first read of tile t+1 by either group.
        const int grp = wave >> 2;
        u32x4 ah[2][WTM], al[2][WTM], bh[2][WTN], bl[2][WTN];
        auto read_frags = [&](int buf) {
            const char* st = smem + buf * STAGE_BYTES;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int i = 0; i < WTM; ++i) { ah[s][i] = lds_read128(st + fa_off[i][s]); al[s][i] = lds_read128(st + fa_off[i][s] + A_BYTES); }
#pragma unroll
                for (int j = 0; j < WTN; ++j) { bh[s][j] = lds_read128(st + fb_off[j][s]); bl[s][j] = lds_read128(st + fb_off[j][s] + B_BYTES); }
            }
        };
        auto mma_all = [&]() {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int i = 0; i < WTM; ++i)
#pragma unroll
                    for (int j = 0; j < WTN; ++j) Fmt::mma(al[s][i], bh[s][j], acc[i][j]);
#pragma unroll
                for (int i = 0; i < WTM; ++i)
#pragma unroll
                    for (int j = 0; j < WTN; ++j) Fmt::mma(ah[s][i], bl[s][j], acc[i][j]);
#pragma unroll
                for (int i = 0; i < WTM; ++i)
#pragma unroll
                    for (int j = 0; j < WTN; ++j) Fmt::mma(ah[s][i], bh[s][j], acc[i][j]);
            }
        };
        if (nk > 1) {
            if constexpr (Cfg::LOADS == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                   // tile 0 is in LDS
        if (grp == 1) __builtin_amdgcn_s_barrier();     // group 1 starts one phase later
        int cur = 0, nxt2 = 2;
        for (int kt = 0; kt < nk; ++kt) {
            // ---- load phase
            if (!(p.dbg & 8) || kt == 0) read_frags(cur);
            if (kt + 2 < nk) {
                if (!(p.dbg & 2)) issue_loads(kt + 2, nxt2);
                if constexpr (Cfg::LOADS == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // this thread's share of tile kt+1 landed
                else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- matrix phase
            __builtin_amdgcn_s_setprio(1);
            if (!(p.dbg & 4)) mma_all();
            else {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
#pragma unroll
                    for (int i = 0; i < WTM; ++i) asm volatile("" :: "v"(ah[s][i]), "v"(al[s][i]));
#pragma unroll
                    for (int j = 0; j < WTN; ++j) asm volatile("" :: "v"(bh[s][j]), "v"(bl[s][j]));
                }
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (grp == 0 || kt + 1 < nk) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            cur = cur == NSTAGE - 1 ? 0 : cur + 1;
            nxt2 = nxt2 == NSTAGE - 1 ? 0 : nxt2 + 1;
        }
    } else if constexpr (NSTAGE == 3) {
        int cur = 0, nxt2 = 2;                              // ring positions of tile kt and tile kt+2
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) {                              // tile kt landed (this thread's part); tile kt+1 may be in flight
                if constexpr (Cfg::LOADS == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();                   // everybody's part; and tile kt-1's buffer is free
            if (kt + 2 < nk) issue_loads(kt + 2, nxt2);
            compute(cur, acc);
            cur = cur == NSTAGE - 1 ? 0 : cur + 1;
            nxt2 = nxt2 == NSTAGE - 1 ? 0 : nxt2 + 1;
        }
    } else {
        // two stages: tile kt+1 streams in under the MFMAs of tile kt; the co-resident block covers the wait
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                   // tile kt landed everywhere; tile kt-1's buffer is free
            if (kt + 1 < nk) issue_loads(kt + 1, (kt + 1) & 1);
            else if (res_pre) prefetch_res();           // behind the last counted wait: lands under the last tile's MFMAs and the stage dump
            compute(kt & 1, acc);
        }
        if constexpr (DUAL) {
            // ---- second product through the same two stages: identity = gather(A2) . B2^T (the downsample branch), joined
            // with the first one in registers exactly as the two launches would be: the identity takes the stored planes'
            // rounding, y = relu((acc * s3 + b3) + join(split(accd * sd + bd))) -- the downsample tensor is neither written nor read
            __syncthreads();                            // every wave has read the first product's last tile
            const bf16_t* A2h = reinterpret_cast<const bf16_t*>(p.A2);
            const bf16_t* B2h = reinterpret_cast<const bf16_t*>(p.B2);
            const auto rsXh = __builtin_amdgcn_make_buffer_rsrc((void*)A2h, 0, (int)p.a2_bytes, 0x00020000);
            const auto rsXl = __builtin_amdgcn_make_buffer_rsrc((void*)(A2h + p.plA2), 0, (int)p.a2_bytes, 0x00020000);
            const auto rsDh = __builtin_amdgcn_make_buffer_rsrc((void*)B2h, 0, (int)p.b2_bytes, 0x00020000);
            const auto rsDl = __builtin_amdgcn_make_buffer_rsrc((void*)(B2h + p.plB), 0, (int)p.b2_bytes, 0x00020000);
            uint32_t x_off[2], d_off[RBP];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int m = m0 + r0 + ROUND * j;
                const int ohw = p.dOH * p.dOW;
                const int img = m / ohw, rem = m - img * ohw, oh = rem / p.dOW, ow = rem - oh * p.dOW;
                const uint32_t pix = (uint32_t)((img * p.dH + oh * p.dstride) * p.dW + ow * p.dstride);
                x_off[j] = m < p.M ? (pix * (uint32_t)p.lda2 + (uint32_t)(gc * 8)) * 2u : OOB;
            }
#pragma unroll
            for (int j = 0; j < RBP; ++j) {
                const int n = n0 + (Cfg::SPLIT_B ? (r0 & 63) : r0 + ROUND * j);
                d_off[j] = n >= p.N ? OOB : p.b_tiled ? ((uint32_t)n * 32u + (uint32_t)(gc * 8)) * 2u : ((uint32_t)n * (uint32_t)p.ldb2 + (uint32_t)(gc * 8)) * 2u;
            }
            auto issue2 = [&](int kt, int buf) {
                char* sa = smem + buf * STAGE_BYTES + wave * 1024;
                char* sb = sa + 2 * A_BYTES;
                const int soff = kt * BK * 2, soffb = p.b_tiled ? kt * p.N * 64 : soff;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    glds16(rsXh, sa + j * (ROUND * ROWB), x_off[j], soff);
                    glds16(rsXl, sa + A_BYTES + j * (ROUND * ROWB), x_off[j], soff);
                }
#pragma unroll
                for (int j = 0; j < RBP; ++j) {
                    if constexpr (Cfg::SPLIT_B) {
                        if (b_lo_half) glds16(rsDl, sb, d_off[j], soffb); else glds16(rsDh, sb, d_off[j], soffb);
                    } else {
                        glds16(rsDh, sb + j * (ROUND * ROWB), d_off[j], soffb);
                        glds16(rsDl, sb + B_BYTES + j * (ROUND * ROWB), d_off[j], soffb);
                    }
                }
            };
            f32x16 accd[WTM][WTN];
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) accd[i][j][e] = 0.f;
            const int nk2 = p.K2 / BK;
            issue2(0, 0);
            for (int kt = 0; kt < nk2; ++kt) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (kt + 1 < nk2) issue2(kt + 1, (kt + 1) & 1);
                compute(kt & 1, accd);
            }
#pragma unroll
            for (int j = 0; j < WTN; ++j) {
                const int col = min(n0 + wn * (WTN * 32) + j * 32 + l31, p.N - 1);
                const float s3 = p.colscale ? p.alpha * p.colscale[col] : p.alpha, b3 = p.bias ? p.bias[col] : 0.f;
                const float sd = p.colscale2 ? p.colscale2[col] : 1.f, bd = p.bias2 ? p.bias2[col] : 0.f;
#pragma unroll
                for (int i = 0; i < WTM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const float d0 = sd * accd[i][j][r] + bd, d1 = sd * accd[i][j][r + 1] + bd;
                        const uint32_t h = Fmt::pack2(d0, d1);
                        const uint32_t l = Fmt::rest2(d0, d1, h);
                        acc[i][j][r] = x3_relu((s3 * acc[i][j][r] + b3) + Fmt::sum_lo(h, l));
                        acc[i][j][r + 1] = x3_relu((s3 * acc[i][j][r + 1] + b3) + Fmt::sum_hi(h, l));
                    }
            }
        }
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
    if (!e_live) return;

    const bf16_t* resh = p.res ? reinterpret_cast<const bf16_t*>(p.res) : nullptr;
    bf16_t* ch = p.out_dtype != SQ_F32 ? reinterpret_cast<bf16_t*>(p.C) : nullptr;
    float* c32 = p.out_dtype == SQ_F32 ? reinterpret_cast<float*>(p.C) : nullptr;
    constexpr int U = 4;
    // with the first rows already in registers, the remaining ones are requested now and land under the first group's work
    u32x4 nh[U], nl[U];
    const bool res_nxt = res_pre && ITER == 2 * U && NPRE == U;
    if (res_nxt) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = min(m0 + e_rbase + (U + u) * RPI, p.M - 1);
            nh[u] = ld16(resh + (long long)m * p.ldres + e_n);
            nl[u] = ld16(resh + p.plRes + (long long)m * p.ldres + e_n);
        }
    }
#pragma unroll
    for (int c0 = 0; c0 < ITER; c0 += U) {
        // residual rows of a group are requested before that group's first store (vmcnt counts stores too)
        u32x4 rh[U], rl[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            rh[u] = u32x4{0, 0, 0, 0}; rl[u] = u32x4{0, 0, 0, 0};
            const int m = m0 + e_rbase + (c0 + u) * RPI;
            if constexpr (BM_ == 128) {
                if (res_pre && c0 + u < NPRE) { rh[u] = rpre_h[c0 + u]; rl[u] = rpre_l[c0 + u]; continue; }
                if (res_nxt && c0 == U) { rh[u] = nh[u]; rl[u] = nl[u]; continue; }
            }
            if (resh && m < p.M) {
                rh[u] = ld16(resh + (long long)m * p.ldres + e_n);
                rl[u] = ld16(resh + p.plRes + (long long)m * p.ldres + e_n);
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
            if constexpr (!DUAL) {                 // (dual form: the stage already holds the finished values)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = scale8[e] * v[e] + bias8[e];
            if (resh) {
                float idn[8];
                x3_join8<F16>(rh[u], rl[u], idn);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += idn[e];
            }
            if (p.act == SQ_ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = x3_relu(v[e]);
            }
            }
            if (p.dbg & 1) continue;
            if (c32) {
                float* d = c32 + (long long)m * p.ldc + e_n;
                *reinterpret_cast<f32x4*>(d) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(d + 4) = f32x4{v[4], v[5], v[6], v[7]};
            }
            if (ch) {
                u32x4 hi, lo;
                x3_split8<F16>(v, hi, lo);
                st16(ch + (long long)m * p.ldc + e_n, hi);
                st16(ch + p.plC + (long long)m * p.ldc + e_n, lo);
            }
        }
    }
}

template <int BM_, int WTN, bool F16, bool PP>
int launch_x3(const GemmArgs& a, hipStream_t stream) {
    using Cfg = X3Cfg<BM_, WTN>;
    static SqDevOnce attr;       // hipFuncSetAttribute is per device
    if (attr.needed()) {
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_x3_kernel<BM_, WTN, false, F16, PP>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_x3_kernel<BM_, WTN, true, F16, PP>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
        attr.done();
    }
    const int tiles = ((a.M + BM_ - 1) / BM_) * ((a.N + Cfg::BN - 1) / Cfg::BN);
    const dim3 grid(tiles), block(Cfg::NT);
    if (a.conv) hipLaunchKernelGGL((gemm_x3_kernel<BM_, WTN, true, F16, PP>), grid, block, Cfg::LDS_BYTES, stream, a);
    else hipLaunchKernelGGL((gemm_x3_kernel<BM_, WTN, false, F16, PP>), grid, block, Cfg::LDS_BYTES, stream, a);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}

template <bool F16>
int launch_x3_dual(const GemmArgs& a, hipStream_t stream) {
    using Cfg = X3Cfg<128, 2>;
    static SqDevOnce attr;       // hipFuncSetAttribute is per device
    if (attr.needed()) {
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_x3_kernel<128, 2, false, F16, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
        attr.done();
    }
    const int tiles = ((a.M + 127) / 128) * (a.N / Cfg::BN);
    hipLaunchKernelGGL((gemm_x3_kernel<128, 2, false, F16, false, true>), dim3(tiles), dim3(Cfg::NT), Cfg::LDS_BYTES, stream, a);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}

template <int BM_, bool PP>
int launch_x3_fmt(const GemmArgs& a, hipStream_t stream) {
    if (a.x3_f16) return a.N % 128 == 0 ? launch_x3<BM_, 2, true, PP>(a, stream) : launch_x3<BM_, 1, true, PP>(a, stream);
    return a.N % 128 == 0 ? launch_x3<BM_, 2, false, PP>(a, stream) : launch_x3<BM_, 1, false, PP>(a, stream);
}

}  // namespace

// C = act(alpha * A.B^T + bias + res): A, B (and res, and C unless out_dtype == SQ_F32) as hi / lo bf16 planes;
// a.A / a.B / a.res / a.C point at the hi plane, the lo plane sits plA / plB / plRes / plC ELEMENTS behind it.
int sq_launch_gemm_x3(const GemmArgs& a_in, hipStream_t stream) {
    GemmArgs a = a_in;
    a.dbg |= g_dbg;
    SQ_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.batch == 1, "gemm_x3: empty or batched problem M=%d N=%d K=%d batch=%d", a.M, a.N, a.K, a.batch);
    SQ_REQUIRE(a.K % 8 == 0 && a.ldb % 8 == 0 && a.N % 8 == 0, "gemm_x3: K=%d / ldb=%d / N=%d must be multiples of 8", a.K, a.ldb, a.N);
    SQ_REQUIRE(a.plA != 0 && a.plB != 0 && (a.plA & 7) == 0 && (a.plB & 7) == 0, "gemm_x3: operand plane strides must be non-zero multiples of 8 elements");
    SQ_REQUIRE(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.B & 15) == 0 && ((uintptr_t)a.C & 15) == 0, "gemm_x3: A/B/C must be 16-byte aligned");
    SQ_REQUIRE(a.a_bytes > 0 && a.a_bytes < (1ull << 31) && a.b_bytes > 0 && a.b_bytes < (1ull << 31),
               "gemm_x3: operand plane extents must be in (0, 2 GiB): %zu %zu", a.a_bytes, a.b_bytes);
    SQ_REQUIRE(a.out_dtype == SQ_F32 || ((a.out_dtype == SQ_BF16X3 || a.out_dtype == SQ_F16X3) && a.plC != 0 && (a.plC & 7) == 0), "gemm_x3: output is fp32 or hi/lo planes (plC)");
    SQ_REQUIRE(a.ldc % 8 == 0 && (!a.res || ((a.res_dtype == SQ_BF16X3 || a.res_dtype == SQ_F16X3) && a.ldres % 8 == 0 && a.plRes != 0 && (a.plRes & 7) == 0 && ((uintptr_t)a.res & 15) == 0)),
               "gemm_x3: ldc / residual planes must allow 16-byte accesses");
    SQ_REQUIRE(!a.rowbias && !a.Cpre && !a.gelu_grad_of && !a.ln64_g && !a.C2 && (a.act == SQ_ACT_NONE || a.act == SQ_ACT_RELU),
               "gemm_x3: only bias / residual / ReLU epilogues");
    SQ_REQUIRE((!a.bias || ((uintptr_t)a.bias & 15) == 0) && (!a.colscale || ((uintptr_t)a.colscale & 15) == 0), "gemm_x3: bias / colscale must be 16-byte aligned");
    SQ_REQUIRE(!a.b_tiled || a.K % BK == 0, "gemm_x3: K-tile-major weights need K %% %d == 0 (K=%d)", BK, a.K);
    if (a.conv) SQ_REQUIRE(a.Cin % BK == 0, "conv_x3: Cin=%d must be a multiple of the K-tile (%d)", a.Cin, BK);
    else SQ_REQUIRE(a.lda % 8 == 0, "gemm_x3: lda=%d must be a multiple of 8", a.lda);
    const bool dual = a.A2 != nullptr;
    if (dual) {
        SQ_REQUIRE(!a.conv && !a.res && a.act == SQ_ACT_RELU && a.N % 128 == 0 && a.out_dtype != SQ_F32, "gemm_x3 dual: plain product, no residual, ReLU, N %% 128 == 0, plane output");
        SQ_REQUIRE(a.B2 && a.K2 > 0 && a.K2 % BK == 0 && a.lda2 % 8 == 0 && a.plA2 != 0 && (a.plA2 & 7) == 0 && ((uintptr_t)a.A2 & 15) == 0 && ((uintptr_t)a.B2 & 15) == 0,
                   "gemm_x3 dual: second operand pair (K2=%d)", a.K2);
        SQ_REQUIRE(a.a2_bytes > 0 && a.a2_bytes < (1ull << 31) && a.b2_bytes > 0 && a.b2_bytes < (1ull << 31) && (a.b_tiled || a.ldb2 % 8 == 0), "gemm_x3 dual: operand extents");
        SQ_REQUIRE(a.dOH > 0 && a.dOW > 0 && a.M % (a.dOH * a.dOW) == 0 && a.dstride >= 1 && (a.dOH - 1) * a.dstride < a.dH && (a.dOW - 1) * a.dstride < a.dW,
                   "gemm_x3 dual: gather geometry %dx%d -> %dx%d stride %d", a.dH, a.dW, a.dOH, a.dOW, a.dstride);
    }
    int prof = -1;
    if (sq_prof_on() && dual) {
        char name[96];
        snprintf(name, sizeof(name), "dual_%s_M%d_N%d_K%d_K%d", a.x3_f16 ? "f16x3" : "bf16x3", a.M, a.N, a.K, a.K2);
        prof = sq_prof_begin(name, 2.0 * a.M * (double)a.N * (a.K + a.K2), ((double)a.M * (a.K + a.K2) + (double)a.N * (a.K + a.K2)) * 4.0 + (double)a.M * a.N * 4.0, stream);
    } else if (sq_prof_on()) {
        const double flops = 2.0 * a.M * (double)a.N * a.K;      // algorithmic (fp32-equivalent) work; the kernel issues 3x that in bf16 MFMAs
        const double a_elems = a.conv ? (double)a.M / (a.OH * a.OW) * a.H * a.W * a.Cin : (double)a.M * a.K;
        const double bytes = (a_elems + (double)a.N * a.K) * 4.0 + (double)a.M * a.N * (4.0 + (a.res ? 4.0 : 0.0));
        char name[96];
        snprintf(name, sizeof(name), "%s_%s_M%d_N%d_K%d", a.conv ? "conv" : "gemm", a.x3_f16 ? "f16x3" : "bf16x3", a.M, a.N, a.K);
        prof = sq_prof_begin(name, flops, bytes, stream);
    }
    // products with K up to this take the 128-row, two-blocks-per-CU shape (sq_dbg_set key 7 overrides: probes; 512 ... 2048 measured level in the pipeline)
    const int small_max_k = g_x3_small_max_k >= 0 ? g_x3_small_max_k : 256;
    const int rc = dual ? (a.x3_f16 ? launch_x3_dual<true>(a, stream) : launch_x3_dual<false>(a, stream))
                 : (g_x3_halo != 0 && sq_conv_halo_x3_eligible(a)) ? sq_launch_conv_halo_x3(a, stream)
                 : (a.K <= small_max_k || (a.N <= 64 && small_max_k > 0)) ? launch_x3_fmt<128, false>(a, stream)
                 : launch_x3_fmt<256, true>(a, stream);
    if (prof >= 0) sq_prof_end(prof, stream);
    return rc;
}

// C-ABI entry (include/sequoia_hip.h): one split-bf16 linear layer / convolution on caller-owned hi / lo planes
extern "C" int sq_linear_x3(int fmt, const void* A_hi, const void* A_lo, int lda, const void* W_hi, const void* W_lo, int ldw, const float* bias,
                            const float* colscale, const void* res_hi, const void* res_lo, int ldres, int act, void* C_hi, void* C_lo, float* C_f32, int ldc,
                            int M, int N, int K, const int* conv_geom, void* stream) {
    SQ_REQUIRE(A_hi && A_lo && W_hi && W_lo && ((C_hi && C_lo) != (C_f32 != nullptr)), "linear_x3: null pointer (give C_hi + C_lo or C_f32)");
    SQ_REQUIRE((res_hi == nullptr) == (res_lo == nullptr), "linear_x3: residual needs both planes");
    SQ_REQUIRE(fmt == 0 || fmt == 1, "linear_x3: fmt %d (0 = bf16 planes, 1 = fp16 planes)", fmt);
    GemmArgs g;
    g.x3_f16 = fmt; g.colscale = colscale;
    const int xdt = fmt ? SQ_F16X3 : SQ_BF16X3;
    g.M = M; g.N = N; g.K = K;
    g.A = A_hi; g.plA = (const bf16_t*)A_lo - (const bf16_t*)A_hi; g.lda = lda;
    g.B = W_hi; g.plB = (const bf16_t*)W_lo - (const bf16_t*)W_hi; g.ldb = ldw;
    g.b_bytes = ((size_t)(N - 1) * ldw + K) * 2;
    if (conv_geom) {        // {n_img, H, W, Cin, OH, OW, KW, stride, pad}: NHWC input, K = KH*KW*Cin tap-major, M = n_img*OH*OW
        g.conv = 1; g.H = conv_geom[1]; g.W = conv_geom[2]; g.Cin = conv_geom[3]; g.OH = conv_geom[4]; g.OW = conv_geom[5];
        g.KW = conv_geom[6]; g.stride = conv_geom[7]; g.pad = conv_geom[8];
        SQ_REQUIRE(M == conv_geom[0] * g.OH * g.OW && g.Cin > 0 && K % g.Cin == 0, "linear_x3: inconsistent convolution geometry");
        g.a_bytes = (size_t)conv_geom[0] * g.H * g.W * g.Cin * 2;
    } else {
        g.a_bytes = ((size_t)(M - 1) * lda + K) * 2;
    }
    g.bias = bias; g.act = act;
    if (res_hi) { g.res = res_hi; g.plRes = (const bf16_t*)res_lo - (const bf16_t*)res_hi; g.ldres = ldres; g.res_dtype = xdt; }
    if (C_f32) { g.C = C_f32; g.out_dtype = SQ_F32; }
    else { g.C = C_hi; g.plC = (bf16_t*)C_lo - (bf16_t*)C_hi; g.out_dtype = xdt; }
    g.ldc = ldc;
    return sq_launch_gemm_x3(g, (hipStream_t)stream);
}
