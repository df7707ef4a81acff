// 3x3 / stride 1 / pad 1 convolution with the INPUT TILE RESIDENT in LDS, split ("x3") modes -- the parity-grade twin of
// conv_halo.hip.  Every value is a hi + lo pair of 16-bit planes (x3_fmt.h), a product is three MFMAs.
//
// Measured on the implicit-GEMM form (gemm_x3.hip, tools/x3_probe.py, layer-3 shape 98000 x 256 x 2304): 361 us, of which
// the MFMAs alone need 139 us and the L2 -> LDS staging alone 222 us (2.7 GB at ~12 TB/s: every tap re-gathers the same
// activation rows) -- the kernel is bound by the bytes the CUs load, not by the matrix pipes.  Here the tile is 256
// CONSECUTIVE flat pixels of the NHWC activation, so the rows all nine taps need are the contiguous range
// [p0 - W - 1, p0 + 256 + W + 1): per 32-channel block both planes are copied to LDS once (one linear LDS-DMA stream) and
// the MFMA A fragments of tap (dy, dx) are read at row offset dy*W + dx; image borders are a 9-bit validity mask per
// output pixel.  Only the weights still stream per K-step:
//   per 256 x 128 output tile and 32-channel block:   A 40 KB (once) + B 9 x 16 KB     instead of     9 x (32 + 16) KB.
// 8 waves (4 x 2, 64 x 64 per wave), 24 MFMAs per wave and step.  LDS: two input-block buffers (hi + lo, the next block's
// rows arrive while the current one is multiplied) + a four-stage ring of weight tiles (three in flight, counted vmcnt +
// raw s_barrier) = 144 KiB, one block per CU.  Maps 32 ... 63 wide (the 56 x 56 stage) stage 384 rows per block; its
// 64-channel output uses a 256 x 64 tile (64 x 32 per wave) whose weight tile is one LDS-DMA instruction for both planes.
// 64-byte LDS rows, 16-byte chunk ^= (row >> 2) & 3 on the SOURCE address.
// K is walked channel-block-major: the fp32 accumulation order differs from the tap-major implicit GEMM (fp32 rounding).
#include "gemm.h"
#include "x3_fmt.h"

#include <cstdlib>
#include <type_traits>

namespace {

constexpr uint32_t OOB = 0x80000000u;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_base, uint32_t voffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_base, 16, voffset, 0, 0, 0);
}
__device__ __forceinline__ u32x4 lds128(const char* p) { return *reinterpret_cast<const u32x4*>(p); }

constexpr int BM = 256, ROWB = 64, CB = 32;
constexpr int NB = 4;                             // weight ring stages

// WTN: 32-column MFMA tiles per wave (BN = 64 WTN: 128 for the 28 x 28 ... 7 x 7 stages, 64 for the 56 x 56 stage);
// HROWS: staged input rows per block, >= 256 + 2 W + 2 (320: maps up to 31 wide, 384: up to 63 wide)
template <int WTN, int HROWS> struct HxCfg {
    static constexpr int BN = 64 * WTN;
    static constexpr int HP_BYTES = HROWS * ROWB;          // one plane of one input block: 20 / 24 KiB
    static constexpr int HALO_BYTES = 2 * HP_BYTES;        // hi plane, then lo plane
    static constexpr int BP_BYTES = BN * ROWB;             // one plane of one weight tile [BN n][32 k]: 8 / 4 KiB
    static constexpr int BT_BYTES = 2 * BP_BYTES;
    static constexpr int RING = 2 * HALO_BYTES + NB * BT_BYTES;
    static constexpr int EPI = BM * BN * 4;
    static constexpr int LDS_BYTES = RING > EPI ? RING : EPI;     // 144 KiB (320, 128) ... 160 KiB (384, 128); 128 KiB (384, 64)
    static constexpr int HL = HALO_BYTES / 16 / 512;       // 5 / 6 LDS-DMA instructions per thread and input block
    static constexpr int PSLOTS = HP_BYTES / 16;           // 16-byte slots per plane: 20 / 24 wave instructions
    static constexpr int WL = WTN;                         // LDS-DMA instructions per thread and weight tile (BN = 64: both planes in one)
};

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int WTN, int HROWS, bool F16, bool PP>
__global__ __launch_bounds__(512) void conv_halo_x3_kernel(const GemmArgs p) {
    using Fmt = X3Fmt<F16>;
    using Cfg = HxCfg<WTN, HROWS>;
    constexpr int BN = Cfg::BN, HP_BYTES = Cfg::HP_BYTES, HALO_BYTES = Cfg::HALO_BYTES, BP_BYTES = Cfg::BP_BYTES, BT_BYTES = Cfg::BT_BYTES;
    constexpr int HL = Cfg::HL, PSLOTS = Cfg::PSLOTS, WL = Cfg::WL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const HB = smem;                        // [2][HALO_BYTES]
    char* const WR = smem + 2 * HALO_BYTES;       // [NB][BT_BYTES]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    const int tiles_n = p.N / BN, tiles_m = (p.M + BM - 1) / BM;
    const int nwg = tiles_m * tiles_n;
    int t;
    {
        const int b = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = b & 7, idx = b >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int p0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;
    const int W = p.W, HWp = p.H * p.W, C = p.Cin;
    const int ncb = (p.dbg & 4) ? 0 : (C / CB);        // dbg 4 (x3_probe.py halo): no K loop -- what a tile costs around it
    const int halo0 = p0 - W - 1;
    const int halo_slots = (BM + 2 * W + 2) * 4;  // per plane

    const uint16_t* Ah = reinterpret_cast<const uint16_t*>(p.A);
    const uint16_t* Bh = reinterpret_cast<const uint16_t*>(p.B);
    const auto rsAh = __builtin_amdgcn_make_buffer_rsrc((void*)Ah, 0, (int)p.a_bytes, 0x00020000);
    const auto rsAl = __builtin_amdgcn_make_buffer_rsrc((void*)(Ah + p.plA), 0, (int)p.a_bytes, 0x00020000);
    const auto rsBh = __builtin_amdgcn_make_buffer_rsrc((void*)Bh, 0, (int)p.b_bytes, 0x00020000);
    const auto rsBl = __builtin_amdgcn_make_buffer_rsrc((void*)(Bh + p.plB), 0, (int)p.b_bytes, 0x00020000);

    // input block: rows [halo0, halo0 + HROWS) x channels [32 cb, 32 cb + 32), hi plane then lo plane.  Instruction u of a wave
    // covers slots [u*512 + wave*64, +64); a plane is a whole number of wave instructions, so the plane is wave-uniform
    // (HROWS 320: u < 2 hi, u > 2 lo, u == 2 hi for waves 0-3 and lo for waves 4-7).
    auto issue_halo = [&](int cb, int buf) {
#pragma unroll
        for (int u = 0; u < HL; ++u) {
            const int q = u * 512 + tid;
            const bool lo = u * 512 + wave * 64 >= PSLOTS;       // wave-uniform
            const int s = lo ? q - PSLOTS : q;
            const int row = s >> 2, c = (s & 3) ^ ((row >> 2) & 3);
            const int px = halo0 + row;
            const bool ok = s < halo_slots && px >= 0 && px < p.M;
            const uint32_t off = ok ? ((uint32_t)px * (uint32_t)C + (uint32_t)(cb * CB + c * 8)) * 2u : OOB;
            char* dst = HB + buf * HALO_BYTES + u * 8192 + wave * 1024;
            if (lo) glds16(rsAl, dst, off); else glds16(rsAh, dst, off);
        }
    };
    // weight tile [n0 + n][tap*C + cb*32 + 0..31], hi plane then lo plane.  BN = 128: one instruction per plane;
    // BN = 64: one instruction for both (waves 0-3 the hi plane, waves 4-7 the lo plane)
    const int br0 = (tid >> 2) & (BN - 1), bc = (tid & 3) ^ ((br0 >> 2) & 3);
    const uint32_t b_row = p.b_tiled ? (uint32_t)(n0 + br0) * 32u + (uint32_t)(bc * 8) : (uint32_t)(n0 + br0) * (uint32_t)p.ldb + (uint32_t)(bc * 8);
    auto issue_wtile = [&](int cb, int tap, int stage) {
        const int k0 = tap * C + cb * CB;
        const uint32_t off = (b_row + (p.b_tiled ? (uint32_t)(k0 >> 5) * (uint32_t)p.N * 32u : (uint32_t)k0)) * 2u;
        char* dst = WR + stage * BT_BYTES + wave * 1024;
        if constexpr (WTN == 2) {
            glds16(rsBh, dst, off);
            glds16(rsBl, dst + BP_BYTES, off);
        } else {
            if (wave >= 4) glds16(rsBl, dst, off); else glds16(rsBh, dst, off);
        }
    };

    // this lane's two output pixels: halo rows of the centre tap and tap validity
    int jc[2];
    uint32_t mask[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ml = wm * 64 + i * 32 + l31;
        const int px = p0 + ml;
        jc[i] = ml + W + 1;
        uint32_t mk = 0;
        if (px < p.M) {
            const int rem = px % HWp, r = rem / W, c = rem - r * W;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int rr = r + tp / 3 - 1, cc = c + tp % 3 - 1;
                if (rr >= 0 && rr < p.H && cc >= 0 && cc < W) mk |= 1u << tp;
            }
        }
        mask[i] = mk;
    }
    int fb_off[WTN][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < WTN; ++j) {
            const int row = wn * (32 * WTN) + j * 32 + l31;
            fb_off[j][s] = row * ROWB + (((2 * s + lh) ^ ((row >> 2) & 3)) << 4);
        }

    // epilogue operands (bias, per-channel scale of the pre-scaled fp16 weight rows) before the loop
    constexpr int BN8 = BN / 8, RPI = 512 / BN8, ITER = BM / RPI;
    const int e_c8 = tid % BN8, e_rbase = tid / BN8;
    const int e_n = n0 + e_c8 * 8;
    float bias8[8], scale8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { bias8[e] = 0.f; scale8[e] = p.alpha; }
    if (p.bias) {
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(p.bias + e_n), t1 = *reinterpret_cast<const f32x4*>(p.bias + e_n + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { bias8[e] = t0[e]; bias8[4 + e] = t1[e]; }
    }
    if (p.colscale) {
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(p.colscale + e_n), t1 = *reinterpret_cast<const f32x4*>(p.colscale + e_n + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { scale8[e] = p.alpha * t0[e]; scale8[4 + e] = p.alpha * t1[e]; }
    }

    f32x16 acc[2][WTN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int ntile = ncb * 9;                    // weight tiles of this output tile, g = cb * 9 + tap
    // prologue: input block 0, weight tiles 0, 1, 2
    issue_halo(0, 0);
    issue_wtile(0, 0, 0);
    if (ntile > 1) issue_wtile(0, 1, 1);
    if (ntile > 2) issue_wtile(0, 2, 2);

    // Ping-pong schedule (as gemm_x3.hip): waves 0-3 and 4-7 -- one of each per SIMD -- run half a step apart.  A wave
    // alternates a LOAD phase (the 16 fragment reads of step g with the border masks applied, its share of weight tile g+3
    // and, at tap 4, of the next input block, the counted wait for everything step g+1 reads) and a MATRIX phase (24 MFMAs
    // on registers); one block-wide barrier per phase.  Weight tile g is read in phases 2g / 2g+1 and overwritten by tile
    // g+4, issued in phases 2g+2 / 2g+3; an input block's buffer is rewritten nine steps after its last read.
    const int grp = PP ? wave >> 2 : 0;
    u32x4 ah[2][2], al[2][2], bh[2][WTN], bl[2][WTN];
    auto load_phase = [&](int cb, auto tapc) {
        constexpr int TAP = decltype(tapc)::value;
        const int g = cb * 9 + TAP;
        const char* hb = HB + (cb & 1) * HALO_BYTES;
        const char* wt = WR + (g & 3) * BT_BYTES;
        const int off = (TAP / 3 - 1) * W + (TAP % 3 - 1);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = jc[i] + off;
            const char* arow = hb + j * ROWB;
            const int asw = (j >> 2) & 3;
            const bool aok = (mask[i] >> TAP) & 1u;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int co = ((2 * s + lh) ^ asw) << 4;
                ah[s][i] = lds128(arow + co);
                al[s][i] = lds128(arow + HP_BYTES + co);
                if (!aok) { ah[s][i] = u32x4{0, 0, 0, 0}; al[s][i] = u32x4{0, 0, 0, 0}; }
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < WTN; ++j) { bh[s][j] = lds128(wt + fb_off[j][s]); bl[s][j] = lds128(wt + BP_BYTES + fb_off[j][s]); }
        if constexpr (TAP == 4) {
            if (cb + 1 < ncb) issue_halo(cb + 1, (cb + 1) & 1);       // in FRONT of this step's weight tile
            else {                                                     // keep the per-step load count uniform
#pragma unroll
                for (int u = 0; u < HL; ++u) glds16(rsAh, HB + ((cb + 1) & 1) * HALO_BYTES + u * 8192 + wave * 1024, OOB);
            }
        }
        {
            const int g3 = g + 3;
            if (g3 < ntile) issue_wtile(g3 / 9, g3 % 9, g3 & 3);
        }
        // everything step g+1 reads has landed (this thread's part).  Issued after weight tile g+1: tiles g+2, g+3 and the
        // next input block when it went out at this step or the one before (it is issued in front of its step's tile)
        if (g + 3 < ntile) wait_vm<2 * WL + ((TAP == 4 || TAP == 5) ? HL : 0)>();
        else if (g + 2 < ntile) wait_vm<WL>();
        else wait_vm<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto matrix_phase = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            // the two correction terms first, the leading term last; consecutive MFMAs touch different accumulators
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j) Fmt::mma(al[s][i], bh[s][j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j) Fmt::mma(ah[s][i], bl[s][j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j) Fmt::mma(ah[s][i], bh[s][j], acc[i][j]);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto step = [&](int cb, auto tapc) {
        constexpr int TAP = decltype(tapc)::value;
        if constexpr (PP) {
            load_phase(cb, tapc);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            matrix_phase();
            if (grp == 0 || cb * 9 + TAP + 1 < ntile) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        } else {
            // all eight waves in step, ONE barrier per step: wait for weight tile g (issued three steps ago; behind it two
            // newer tiles and, when it went out in one of the last two steps, the next input block), then issue, read, multiply
            const int g = cb * 9 + TAP;
            if (g + 2 < ntile) wait_vm<2 * WL + ((TAP == 5 || TAP == 6) ? HL : 0)>();
            else if (g + 1 < ntile) wait_vm<WL>();
            else wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            if constexpr (TAP == 4) {
                if (cb + 1 < ncb) issue_halo(cb + 1, (cb + 1) & 1);
                else {
#pragma unroll
                    for (int u = 0; u < HL; ++u) glds16(rsAh, HB + ((cb + 1) & 1) * HALO_BYTES + u * 8192 + wave * 1024, OOB);
                }
            }
            {
                const int g3 = g + 3;
                if (g3 < ntile) issue_wtile(g3 / 9, g3 % 9, g3 & 3);
            }
            const char* hb = HB + (cb & 1) * HALO_BYTES;
            const char* wt = WR + (g & 3) * BT_BYTES;
            const int off = (TAP / 3 - 1) * W + (TAP % 3 - 1);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int j = jc[i] + off;
                const char* arow = hb + j * ROWB;
                const int asw = (j >> 2) & 3;
                const bool aok = (mask[i] >> TAP) & 1u;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int co = ((2 * s + lh) ^ asw) << 4;
                    ah[s][i] = lds128(arow + co);
                    al[s][i] = lds128(arow + HP_BYTES + co);
                    if (!aok) { ah[s][i] = u32x4{0, 0, 0, 0}; al[s][i] = u32x4{0, 0, 0, 0}; }
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < WTN; ++j) { bh[s][j] = lds128(wt + fb_off[j][s]); bl[s][j] = lds128(wt + BP_BYTES + fb_off[j][s]); }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < WTN; ++j) Fmt::mma(al[s][i], bh[s][j], acc[i][j]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < WTN; ++j) Fmt::mma(ah[s][i], bl[s][j], acc[i][j]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < WTN; ++j) Fmt::mma(ah[s][i], bh[s][j], acc[i][j]);
            }
        }
    };
    if constexpr (PP) {
        // input block 0 and weight tile 0 in LDS (older than tiles 1, 2)
        if (ntile > 2) wait_vm<2 * WL>(); else if (ntile > 1) wait_vm<WL>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (grp == 1) __builtin_amdgcn_s_barrier();       // group 1 starts one phase later
    }
    for (int cb = 0; cb < ncb; ++cb) {
        step(cb, std::integral_constant<int, 0>{});
        step(cb, std::integral_constant<int, 1>{});
        step(cb, std::integral_constant<int, 2>{});
        step(cb, std::integral_constant<int, 3>{});
        step(cb, std::integral_constant<int, 4>{});
        step(cb, std::integral_constant<int, 5>{});
        step(cb, std::integral_constant<int, 6>{});
        step(cb, std::integral_constant<int, 7>{});
        step(cb, std::integral_constant<int, 8>{});
    }
    __syncthreads();

    // epilogue: fp32 tile through LDS, colscale / bias (+ ReLU), hi / lo planes (or fp32 rows) out
    float* stage = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int col = wn * (32 * WTN) + j * 32 + l31;
                stage[row * BN + col] = acc[i][j][r];
            }
    __syncthreads();
    float* c32 = p.out_dtype == SQ_F32 ? reinterpret_cast<float*>(p.C) : nullptr;
    uint16_t* ch = p.out_dtype != SQ_F32 ? reinterpret_cast<uint16_t*>(p.C) : nullptr;
#pragma unroll
    for (int u = 0; u < ITER; ++u) {
        const int row = e_rbase + u * RPI;
        const int m = p0 + row;
        if (m >= p.M || (p.dbg & 1)) continue;             // dbg 1: no epilogue work per row, no stores
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(stage + row * BN + e_c8 * 8);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(stage + row * BN + e_c8 * 8 + 4);
        float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[e] = scale8[e] * v[e] + bias8[e];
            if (p.act == SQ_ACT_RELU) v[e] = x3_relu(v[e]);
        }
        if (c32) {
            float* d = c32 + (long long)m * p.ldc + e_n;
            *reinterpret_cast<f32x4*>(d) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(d + 4) = f32x4{v[4], v[5], v[6], v[7]};
        }
        if (ch) {
            u32x4 hi, lo;
            x3_split8<F16>(v, hi, lo);
            *reinterpret_cast<u32x4*>(ch + (long long)m * p.ldc + e_n) = hi;
            *reinterpret_cast<u32x4*>(ch + p.plC + (long long)m * p.ldc + e_n) = lo;
        }
    }
}

}  // namespace

// 3x3 / stride 1 / pad 1 split-mode argument sets this kernel takes over (checked by sq_launch_gemm_x3 after its own
// argument validation): maps up to 63 wide, Cin a multiple of 32, N a multiple of 64, bias / ReLU epilogue only
bool sq_conv_halo_x3_eligible(const GemmArgs& a) {
    if (!a.conv) return false;
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("SQ_CONV_HALO");
        on = (e && e[0] == '0') ? 0 : 1;
    }
    if (!on) return false;
    if (a.KW != 3 || a.stride != 1 || a.pad != 1 || a.H != a.OH || a.W != a.OW || a.K != 9 * a.Cin) return false;
    if (a.Cin % CB || a.N % 64 || a.W > 63 || a.W < 3 || a.ldb % 8) return false;
    return a.res == nullptr;
}

namespace {
template <int WTN, int HROWS>
int launch_halo(const GemmArgs& a, hipStream_t stream) {
    using Cfg = HxCfg<WTN, HROWS>;
    static SqDevOnce attr;       // hipFuncSetAttribute is per device
    if (attr.needed()) {
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)conv_halo_x3_kernel<WTN, HROWS, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)conv_halo_x3_kernel<WTN, HROWS, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)conv_halo_x3_kernel<WTN, HROWS, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)conv_halo_x3_kernel<WTN, HROWS, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
        attr.done();
    }
    const int tiles = ((a.M + BM - 1) / BM) * (a.N / Cfg::BN);
    // all eight waves in step for the 64-column tile (12 MFMAs per step and wave: the load phase is the longer one and the
    // ping-pong buys nothing -- 56 x 56 stage 478 vs 489 us)
    const bool lockstep = WTN == 1;
    if (a.x3_f16 && lockstep) hipLaunchKernelGGL((conv_halo_x3_kernel<WTN, HROWS, true, false>), dim3(tiles), dim3(512), Cfg::LDS_BYTES, stream, a);
    else if (a.x3_f16) hipLaunchKernelGGL((conv_halo_x3_kernel<WTN, HROWS, true, true>), dim3(tiles), dim3(512), Cfg::LDS_BYTES, stream, a);
    else if (lockstep) hipLaunchKernelGGL((conv_halo_x3_kernel<WTN, HROWS, false, false>), dim3(tiles), dim3(512), Cfg::LDS_BYTES, stream, a);
    else hipLaunchKernelGGL((conv_halo_x3_kernel<WTN, HROWS, false, true>), dim3(tiles), dim3(512), Cfg::LDS_BYTES, stream, a);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}
}  // namespace

int sq_launch_conv_halo_x3(const GemmArgs& a, hipStream_t stream) {
    const bool wide = a.N % 128 == 0, small_map = a.W <= 31;
    if (wide) return small_map ? launch_halo<2, 320>(a, stream) : launch_halo<2, 384>(a, stream);
    return small_map ? launch_halo<1, 320>(a, stream) : launch_halo<1, 384>(a, stream);
}
