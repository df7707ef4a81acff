// Split-mode ("x3") chain of the 28 x 28 and 14 x 14 stages: one launch does, for a tile of 32 NW consecutive pixels,
//     y   = relu(t2 . w3^T * s3 + b3 + identity)          src/resnet.py:83-91   (conv3 / bn3, += identity, relu)
//     t1' = relu(y . w1'^T * s1' + b1')                   src/resnet.py:75-77 of the NEXT block (conv1 / bn1 / relu)
// on hi / lo planes of 16-bit values (x3_fmt.h), three MFMAs per product -- the counterpart of chain_x3.hip (56 x 56 stage)
// for C = 128 (y has 512 channels) and C = 256 (1024 channels).
//
// Why: unfused, y (4 C channels x 4 bytes per pixel) is written by the expand launch and read back by the next block's
// reduce launch -- 1.6 GB per 1000 patches at 28 x 28, 0.8 GB at 14 x 14, per bottleneck.  Here the reduce is fed from
// LDS: per pixel t2 (4 C bytes) and the identity (16 C) are read, y (16 C) and t1' (4 C) written.
//
// Construction (the transposed-product scheme of chain256.hip, re-budgeted for 4-byte elements and three MFMAs per product):
//   * wave w owns pixels [32 w, 32 w + 32) of the tile; products are transposed (A = weights, B = pixels: lane = pixel), so
//     a lane's accumulators are runs of 4 consecutive channels of ITS pixel -- the epilogue needs no cross-lane traffic and
//     writes y as 8-byte pieces into the wave's private [32 px][64 B] hi / lo image, which is at once the B operand of the
//     second product and the source of the 16-byte global stores;
//   * the wave's t2 rows never touch LDS: the C / 16 k-step fragments of both planes are loaded once into registers
//     (C = 256: 128 VGPRs -- with the 128 accumulator registers of t1' this is why that form runs ONE wave per SIMD with
//     the 512-register budget, NW = 4; C = 128: 64 + 64 registers, NW = 8, two waves per SIMD);
//   * y is produced in slices of 32 channels: P1 = t2 . w3[slice]^T (3 C / 16 MFMAs per wave), epilogue (bias, identity
//     -- LDS-DMA'd one slice ahead into the wave's image of the other parity -- ReLU, split, in place), then
//     P2: t1' += y[slice] . w1'[:, slice]^T (3 * 2 * C / 32 MFMAs);
//   * weights stream through a three-slot ring of 128 C-byte units (w3 rows of the slice: [C / 32 k-tiles][32 n][64 B] x 2
//     planes; w1' columns of the slice: [C n][64 B] x 2 planes; 16-byte chunk ^= (row >> 2) & 3 on the SOURCE address), two
//     units in flight, counted s_waitcnt vmcnt + raw s_barrier: two barriers per slice;
//   * biases and per-channel scales sit in LDS (ds_read: outside the vmcnt bookkeeping).
// Same K order, same MFMA order per accumulator (a_lo.b_hi, a_hi.b_lo, a_hi.b_hi per 16-deep step, steps ascending) and the
// same epilogue arithmetic as the two gemm_x3.hip launches it replaces: bit-identical results (tests/test_gpu_x3.py).
#include "gemm.h"
#include "x3_fmt.h"

#include <cstdio>
#include <type_traits>

extern int g_dbg;                // sq_dbg_set key 1 (gemm.hip)

namespace {

constexpr uint32_t OOB = 0x80000000u;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_base, uint32_t voffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_base, 16, voffset, 0, 0, 0);
}
__device__ __forceinline__ u32x4 lds128(const char* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ u32x2 lds64(const char* p) { return *reinterpret_cast<const u32x2*>(p); }
__device__ __forceinline__ void st_lds64(char* p, u32x2 v) { *reinterpret_cast<u32x2*>(p) = v; }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct ChainWArgs {
    const uint16_t* t2; long long plT2;          // [P, C] planes
    const uint16_t* res; long long plRes;        // identity [P, 4 C]
    uint16_t* y; long long plY;                  // [P, 4 C]
    uint16_t* t1n; long long plT1n;              // [P, N2]
    const uint16_t* w3; const uint16_t* w1n; long long plW;     // [4 C, C] / [N2, 4 C]; lo plane plW elements behind
    const float* b3; const float* cs3; const float* b1n; const float* cs1n;
    int P, tiles, tiled;
    uint32_t w3_bytes, w1n_bytes;                // descriptor extents of one weight plane
    int dbg;                                     // ablation switches (tools/chainw_probe.py): 1 no global stores, 2 no identity reads, 4 no first product, 8 no second product, 16 no weight stream, 32 no epilogue
};

template <int C, int N2, int NW> struct ChainWCfg {
    static constexpr int N1 = 4 * C, NS = N1 / 32, KS1 = C / 16, NT2 = N2 / 32, T = NW * 64, PX = NW * 32;
    static constexpr int UNIT = 128 * C, HALF = UNIT / 2, LPT = UNIT / (T * 16);      // ring unit (both planes), loads per thread per unit
    static constexpr int RING = 3 * UNIT;
    static constexpr int XYW = 4096;             // one wave's image of one slice: [32 px][64 B] hi | [32 px][64 B] lo
    static constexpr int XY = NW * 3 * XYW;      // three slices per wave: identity arriving, y being made, y being multiplied / stored
    static constexpr int BIAS_FLOATS = 2 * N1 + 2 * N2;        // b3 | cs3 | b1' | cs1'
    static constexpr int LDS_BYTES = RING + XY + BIAS_FLOATS * 4;
    static constexpr int IDL = 4;                // identity loads per thread per slice
    static_assert(N2 == C, "the w1' unit (128 N2 bytes) must equal the w3 unit (128 C bytes)");
    static_assert(UNIT % (T * 16) == 0 && LPT >= 1 && (4 * C) % T == 0, "a plane of a unit must be a whole number of block-wide LDS-DMA rounds");
    static_assert(32 * N2 * 2 * NW <= RING + XY, "t1' staging (one plane per wave) must fit the dead ring");
};

template <int C, int N2, int NW, bool F16, bool DBG>
__global__ __launch_bounds__(NW * 64, NW / 4) void chain_x3w_kernel(const ChainWArgs p) {
    using Cfg = ChainWCfg<C, N2, NW>;
    const int dbg = DBG ? p.dbg : 0;              // the ablation switches exist in the DBG instantiation only (they split the phases' basic blocks)
    using Fmt = X3Fmt<F16>;
    constexpr int N1 = Cfg::N1, NS = Cfg::NS, KS1 = Cfg::KS1, NT2 = Cfg::NT2, T = Cfg::T;
    constexpr int UNIT = Cfg::UNIT, HALF = Cfg::HALF, LPT = Cfg::LPT, XYW = Cfg::XYW, IDL = Cfg::IDL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const RG = smem;
    char* const XYB = smem + Cfg::RING;
    float* const BS = reinterpret_cast<float*>(smem + Cfg::RING + Cfg::XY);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int swz = (l31 >> 2) & 3;
    int t;
    {   // each XCD (block id % 8) walks a contiguous run of tiles
        const int b = blockIdx.x, q = p.tiles >> 3, r = p.tiles & 7, xcd = b & 7, idx = b >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int p0 = t * Cfg::PX + wave * 32;       // this wave's first pixel

    const uint32_t act_bytes = (uint32_t)p.P * N1 * 2u;
    const auto rsRh = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, (int)act_bytes, 0x00020000);
    const auto rsRl = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res + p.plRes), 0, (int)act_bytes, 0x00020000);
    const auto rsYh = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)act_bytes, 0x00020000);
    const auto rsYl = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + p.plY), 0, (int)act_bytes, 0x00020000);
    const auto rsTh = __builtin_amdgcn_make_buffer_rsrc((void*)p.t1n, 0, (int)((uint32_t)p.P * N2 * 2u), 0x00020000);
    const auto rsTl = __builtin_amdgcn_make_buffer_rsrc((void*)(p.t1n + p.plT1n), 0, (int)((uint32_t)p.P * N2 * 2u), 0x00020000);
    const auto rsW3h = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3, 0, (int)p.w3_bytes, 0x00020000);
    const auto rsW3l = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w3 + p.plW), 0, (int)p.w3_bytes, 0x00020000);
    const auto rsW1h = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1n, 0, (int)p.w1n_bytes, 0x00020000);
    const auto rsW1l = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w1n + p.plW), 0, (int)p.w1n_bytes, 0x00020000);

    // ---- ring unit U in consumption order: U0 = w3 rows of slice 0, odd U = w3 rows of slice (U + 1) / 2, even U >= 2 = w1' columns
    // of slice U / 2 - 1 (U = 2 NS - 1 does not exist); slot U % 3.
    // 16-byte slot Q = round * T + tid of the unit image (hi plane, then lo plane, 4 C slots each).
    auto issue_unit = [&](int U) {
        if (dbg & 16) return;                     // (ablation: no weight stream)
        char* dst = RG + (U % 3) * UNIT + wave * 1024;
        const int u = U == 0 ? 0 : (U & 1) ? 0 : 1;               // 0: w3 rows of slice s, 1: w1' columns of slice s
        const int s = U == 0 ? 0 : (U & 1) ? (U + 1) / 2 : U / 2 - 1;
#pragma unroll
        for (int r = 0; r < LPT; ++r) {
            constexpr int RPP = 4 * C / T;        // rounds per plane: the plane of a round is a compile-time fact (a descriptor picked by a
            const bool lo = r >= RPP;             // per-lane condition would be wrapped in a waterfall loop, guide T20)
            const int q = (r % RPP) * T + tid;    // 16-byte slot of the plane image
            const int row = q >> 2, cl = (q & 3) ^ ((row >> 2) & 3);
            uint32_t off;
            if (u == 0) {                         // w3: image row = (k-tile kt, n): kt = row / 32, n = row % 32
                const int kt = row >> 5, n = s * 32 + (row & 31);
                off = p.tiled ? (uint32_t)(((kt * N1 + n) * 32 + cl * 8) * 2) : (uint32_t)((n * C + kt * 32 + cl * 8) * 2);
            } else {                              // w1': image row = n in [0, N2), k-tile = s
                off = p.tiled ? (uint32_t)(((s * N2 + row) * 32 + cl * 8) * 2) : (uint32_t)((row * N1 + s * 32 + cl * 8) * 2);
            }
            if (u == 0) { if (lo) glds16(rsW3l, dst + r * (T * 16), off); else glds16(rsW3h, dst + r * (T * 16), off); }
            else { if (lo) glds16(rsW1l, dst + r * (T * 16), off); else glds16(rsW1h, dst + r * (T * 16), off); }
        }
    };
    // identity of slice s -> this wave's image of parity s & 1 (IDL loads per thread)
    auto issue_identity = [&](int s) {
        char* dst = XYB + (wave * 3 + s % 3) * XYW;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = u * 64 + lane, row = i >> 2, cl = (i & 3) ^ ((row >> 2) & 3);
            const int pr = p0 + row;
            const uint32_t off = (pr < p.P && !(dbg & 2)) ? ((uint32_t)pr * N1 + (uint32_t)(s * 32 + cl * 8)) * 2u : OOB;
            glds16(rsRh, dst + u * 1024, off);
            glds16(rsRl, dst + 2048 + u * 1024, off);
        }
    };

    issue_identity(0);
    issue_unit(0);
    asm volatile("" ::: "memory");

    // this wave's 32 pixels of t2 as the K-step operands of the first product (lane = pixel l31, K half lh), both planes
    u32x4 xh[KS1], xl[KS1];
    {
        const int pr = p0 + l31;
        const bool ok = pr < p.P;
        const uint16_t* row = p.t2 + (size_t)(ok ? pr : 0) * C + lh * 8;
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            xh[ks] = *reinterpret_cast<const u32x4*>(row + ks * 16);
            xl[ks] = *reinterpret_cast<const u32x4*>(row + p.plT2 + ks * 16);
            if (!ok) { xh[ks] = u32x4{0u, 0u, 0u, 0u}; xl[ks] = u32x4{0u, 0u, 0u, 0u}; }
        }
    }
    for (int i = tid; i < N1; i += T) { BS[i] = p.b3[i]; BS[N1 + i] = p.cs3 ? p.cs3[i] : 1.f; }
    for (int i = tid; i < N2; i += T) { BS[2 * N1 + i] = p.b1n[i]; BS[2 * N1 + N2 + i] = p.cs1n ? p.cs1n[i] : 1.f; }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // everything requested so far has landed (this thread's part)

    f32x16 acc2[NT2];
#pragma unroll
    for (int i = 0; i < NT2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[i][e] = 0.f;
    f32x16 accy;

    // ---- P1(s): y^T slice [32 ch][32 px] = w3[32 s ..][:] . t2^T, from ring unit U
    auto P1 = [&](int U) {
#pragma unroll
        for (int e = 0; e < 16; ++e) accy[e] = 0.f;
        if (dbg & 4) return;
        const char* ub = RG + (U % 3) * UNIT + l31 * 64;
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            const int off = (ks >> 1) * 2048 + (((2 * (ks & 1) + lh) ^ swz) << 4);
            const u32x4 wh = lds128(ub + off), wl = lds128(ub + HALF + off);
            Fmt::mma(wh, xl[ks], accy);       // activation lo . weight hi
            Fmt::mma(wl, xh[ks], accy);       // activation hi . weight lo
            Fmt::mma(wh, xh[ks], accy);       // activation hi . weight hi
        }
    };
    // ---- epilogue group g of slice s: lane (px = l31, half lh) holds channels 32 s + 8 g + 4 lh + (0..3); identity in, y out (in place)
    auto EPI = [&](int s, int g) {
        if (dbg & 32) return;                     // (ablation: no epilogue)
        char* a = XYB + (wave * 3 + s % 3) * XYW + l31 * 64 + ((g ^ swz) << 4) + 8 * lh;
        const int cb = s * 32 + 8 * g + 4 * lh;
        const f32x4 b = *reinterpret_cast<const f32x4*>(BS + cb), sc = *reinterpret_cast<const f32x4*>(BS + N1 + cb);
        const u32x2 ih = lds64(a), il = lds64(a + 2048);
        float v[4];
        v[0] = x3_relu((sc[0] * accy[4 * g + 0] + b[0]) + (Fmt::lo_f(ih[0]) + Fmt::lo_f(il[0])));
        v[1] = x3_relu((sc[1] * accy[4 * g + 1] + b[1]) + (Fmt::hi_f(ih[0]) + Fmt::hi_f(il[0])));
        v[2] = x3_relu((sc[2] * accy[4 * g + 2] + b[2]) + (Fmt::lo_f(ih[1]) + Fmt::lo_f(il[1])));
        v[3] = x3_relu((sc[3] * accy[4 * g + 3] + b[3]) + (Fmt::hi_f(ih[1]) + Fmt::hi_f(il[1])));
        const uint32_t h0 = Fmt::pack2(v[0], v[1]), h1 = Fmt::pack2(v[2], v[3]);
        const uint32_t l0 = Fmt::pack2(v[0] - Fmt::lo_f(h0), v[1] - Fmt::hi_f(h0));
        const uint32_t l1 = Fmt::pack2(v[2] - Fmt::lo_f(h1), v[3] - Fmt::hi_f(h1));
        st_lds64(a, u32x2{h0, h1});
        st_lds64(a + 2048, u32x2{l0, l1});
    };
    // ---- second phase of an iteration: the epilogue of slice se (EPI_ON) under the MFMAs of
    // P2(sp): t1'^T [N2][32 px] += w1'[:, 32 sp ..] . y^T slice sp, from ring unit U (P2_ON).  ONE basic block (no run-time switch
    // inside), and a sched_group_barrier program that deals the epilogue's VALU / LDS work between the MFMAs: with one wave per
    // SIMD nothing else can fill the matrix pipe while the epilogue runs, and nothing else runs under the MFMAs.
    auto PHASE2 = [&](auto epi_on, auto p2_on, int se, int sp, int U) {
        constexpr bool EPI_ON = decltype(epi_on)::value, P2_ON = decltype(p2_on)::value;
        const bool p2 = P2_ON && !(dbg & 8);
        const char* ub = RG + (U % 3) * UNIT + l31 * 64;
        const char* xyp = XYB + (wave * 3 + sp % 3) * XYW + l31 * 64;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
            const int co = ((2 * ks2 + lh) ^ swz) << 4;
            u32x4 yh, yl, wh[NT2], wl[NT2];
            if (P2_ON) {
                yh = lds128(xyp + co); yl = lds128(xyp + 2048 + co);
#pragma unroll
                for (int nt = 0; nt < NT2; ++nt) { wh[nt] = lds128(ub + nt * 2048 + co); wl[nt] = lds128(ub + HALF + nt * 2048 + co); }
            }
            if (EPI_ON) EPI(se, 2 * ks2);
            if (P2_ON && (!DBG || p2)) {
#pragma unroll
                for (int nt = 0; nt < NT2; ++nt) Fmt::mma(wh[nt], yl, acc2[nt]);
            }
            if (EPI_ON) EPI(se, 2 * ks2 + 1);
            if (P2_ON && (!DBG || p2)) {
#pragma unroll
                for (int nt = 0; nt < NT2; ++nt) Fmt::mma(wl[nt], yh, acc2[nt]);
#pragma unroll
                for (int nt = 0; nt < NT2; ++nt) Fmt::mma(wh[nt], yh, acc2[nt]);
            }
        }
        if constexpr (EPI_ON && P2_ON && !DBG) {
            // the pipeline the scheduler is asked for: a burst of fragment reads, then per MFMA a few epilogue VALU ops and a read
            constexpr int NMFMA = 6 * NT2, VPM = NT2 == 8 ? 3 : 6;
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
            for (int i = 0; i < NMFMA; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (i % (NMFMA / 4) == NMFMA / 4 - 1) __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // y slice s -> global: the wave's image read linearly, 64-byte runs per pixel and plane
    auto STORE_Y = [&](int s) {
        if (dbg & 1) return;
        const char* xy = XYB + (wave * 3 + s % 3) * XYW;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = u * 64 + lane, row = i >> 2, cl = (i & 3) ^ ((row >> 2) & 3);
            const int pr = p0 + row;
            const uint32_t off = pr < p.P ? ((uint32_t)pr * N1 + (uint32_t)(s * 32 + cl * 8)) * 2u : OOB;
            const u32x4 vh = lds128(xy + i * 16), vl = lds128(xy + 2048 + i * 16);
            __builtin_amdgcn_raw_buffer_store_b128(vh, rsYh, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(vl, rsYl, off, 0, 0);
        }
    };

    // Software pipeline, one slice deep: iteration s = { P1(s) } { epilogue(s) under P2(s-1) }.  Ring units in consumption order:
    // U0 = w3(0), U(2s-1) = w3(s), U(2s) = w1'(s-1); a phase waits for its unit, crosses one barrier, requests the unit two
    // ahead into the slot the barrier just freed.  Per-thread order of the vector-memory operations in the steady state:
    //   top(s): [y stores s-1] identity s+1 (IDL) unit 2s+1 (LPT) | mid(s): unit 2s+2 (LPT) | top(s+1): ...
    // loads retire in order among themselves; stores only make a counted wait stricter.
    // ---- iteration 0
    __builtin_amdgcn_s_barrier();                 // biases and everybody's part of U0 are in LDS
    issue_identity(1);
    issue_unit(1);
    asm volatile("" ::: "memory");
    P1(0);
    issue_unit(2);                                // fresh slot
    asm volatile("" ::: "memory");
    PHASE2(std::true_type{}, std::false_type{}, 0, 0, 0);
    for (int s = 1; s < NS; ++s) {
        // ---- top: U(2s-1) = w3(s) has landed (this thread's part); U(2s) may be in flight
        wait_vm<LPT>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();             // everybody's part; every wave finished P2(s-2): U(2s-2)'s slot is free
        STORE_Y(s - 1);
        asm volatile("" ::: "memory");
        if (s + 1 < NS) { issue_identity(s + 1); issue_unit(2 * s + 1); }
        asm volatile("" ::: "memory");
        P1(2 * s - 1);
        // ---- mid: U(2s) = w1'(s-1) has landed; identity s+1 and U(2s+1) may be in flight
        if (s + 1 < NS) wait_vm<LPT + IDL>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();             // everybody's part of U(2s); U(2s-1) consumed by every wave
        issue_unit(2 * s + 2);                    // = w1'(s), into U(2s-1)'s slot
        asm volatile("" ::: "memory");
        PHASE2(std::true_type{}, std::true_type{}, s, s - 1, 2 * s);
    }
    // ---- tail: P2 of the last slice
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    STORE_Y(NS - 1);
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    PHASE2(std::false_type{}, std::true_type{}, 0, NS - 1, 2 * NS);

    // ---- t1' = relu(acc2 * s1' + b1') -> planes, one plane at a time through this wave's piece of the dead ring:
    // [32 px][N2 x 2 B], 16-byte chunk ^= px & (chunks per row - 1)
    __syncthreads();
    {
        constexpr int ROWB = N2 * 2, NCH = ROWB / 16;
        char* const mine = smem + wave * (32 * ROWB);
        char* const myrow = mine + l31 * ROWB;
        const int sw = l31 & (NCH - 1);
        u32x2 lo_keep[NT2][4];
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cb = 32 * nt + 8 * g + 4 * lh;
                const f32x4 b = *reinterpret_cast<const f32x4*>(BS + 2 * N1 + cb), sc = *reinterpret_cast<const f32x4*>(BS + 2 * N1 + N2 + cb);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = x3_relu(sc[e] * acc2[nt][4 * g + e] + b[e]);
                const uint32_t h0 = Fmt::pack2(v[0], v[1]), h1 = Fmt::pack2(v[2], v[3]);
                lo_keep[nt][g] = u32x2{Fmt::pack2(v[0] - Fmt::lo_f(h0), v[1] - Fmt::hi_f(h0)), Fmt::pack2(v[2] - Fmt::lo_f(h1), v[3] - Fmt::hi_f(h1))};
                st_lds64(myrow + (((nt * 4 + g) ^ sw) << 4) + 8 * lh, u32x2{h0, h1});
            }
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            if (pl == 1) {
#pragma unroll
                for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) st_lds64(myrow + (((nt * 4 + g) ^ sw) << 4) + 8 * lh, lo_keep[nt][g]);
            }
            if (!(dbg & 1)) {
#pragma unroll
                for (int u = 0; u < (32 * ROWB) / 1024; ++u) {
                    const int i = u * 64 + lane, row = i / NCH, cl = (i % NCH) ^ (row & (NCH - 1));
                    const int pr = p0 + row;
                    const uint32_t off = pr < p.P ? ((uint32_t)pr * N2 + (uint32_t)(cl * 8)) * 2u : OOB;
                    const u32x4 v = lds128(mine + i * 16);
                    if (pl == 0) __builtin_amdgcn_raw_buffer_store_b128(v, rsTh, off, 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(v, rsTl, off, 0, 0);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the plane's reads are done before the next plane overwrites the stage
        }
    }
}

template <int C, int N2, int NW, bool F16>
int launch_chainw(const ChainWArgs& a, hipStream_t stream) {
    using Cfg = ChainWCfg<C, N2, NW>;
    static SqDevOnce attr;       // hipFuncSetAttribute is per device
    if (attr.needed()) {
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)chain_x3w_kernel<C, N2, NW, F16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)chain_x3w_kernel<C, N2, NW, F16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
        attr.done();
    }
    if (a.dbg) hipLaunchKernelGGL((chain_x3w_kernel<C, N2, NW, F16, true>), dim3(a.tiles), dim3(Cfg::T), Cfg::LDS_BYTES, stream, a);
    else hipLaunchKernelGGL((chain_x3w_kernel<C, N2, NW, F16, false>), dim3(a.tiles), dim3(Cfg::T), Cfg::LDS_BYTES, stream, a);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}

}  // namespace

// Which (C, next width) pairs the launch covers: the plain bottlenecks of layer 2 (128 -> 512 -> 128) and layer 3 (256 -> 1024 -> 256)
bool sq_chain_x3w_eligible(int c, int n2) { return (c == 128 || c == 256) && n2 == c; }

// t2 [P, C], res / y [P, 4 C], t1n [P, n2] as hi / lo planes (pl* = elements between the planes);
// w3 [4 C, C] and w1n [n2, 4 C] planes plW apart (row-major, or K-tile-major when w_tiled), biases / per-channel scales fp32 (scales may be null).
// w3_bytes / w1n_bytes: bytes from the pointer to the end of one weight plane's allocation (descriptor extent).
int sq_launch_chain_x3w(int f16, int c, const uint16_t* t2, long long plT2, const uint16_t* res, long long plRes, uint16_t* y, long long plY,
                        uint16_t* t1n, long long plT1n, int n2, const uint16_t* w3, const uint16_t* w1n, long long plW, size_t w3_bytes, size_t w1n_bytes,
                        const float* b3, const float* cs3, const float* b1n, const float* cs1n, long long P, int w_tiled, hipStream_t stream) {
    SQ_REQUIRE(sq_chain_x3w_eligible(c, n2), "chain_x3w: C=%d next width %d (128 -> 128 or 256 -> 256)", c, n2);
    SQ_REQUIRE(P > 0 && P * 4 * c * 2 < (1ll << 31), "chain_x3w: %lld pixels exceed the 2 GiB descriptor limit", P);
    SQ_REQUIRE(t2 && res && y && t1n && w3 && w1n && b3 && b1n, "chain_x3w: null pointer");
    SQ_REQUIRE(w3_bytes >= (size_t)4 * c * c * 2 && w1n_bytes >= (size_t)n2 * 4 * c * 2, "chain_x3w: weight extents");
    SQ_REQUIRE((((uintptr_t)t2 | (uintptr_t)res | (uintptr_t)y | (uintptr_t)t1n | (uintptr_t)w3 | (uintptr_t)w1n) & 15) == 0 &&
               ((plT2 | plRes | plY | plT1n | plW) & 7) == 0, "chain_x3w: planes must be 16-byte aligned");
    ChainWArgs a;
    a.t2 = t2; a.plT2 = plT2; a.res = res; a.plRes = plRes; a.y = y; a.plY = plY; a.t1n = t1n; a.plT1n = plT1n;
    a.w3 = w3; a.w1n = w1n; a.plW = plW; a.b3 = b3; a.cs3 = cs3; a.b1n = b1n; a.cs1n = cs1n;
    a.P = (int)P; a.tiled = w_tiled; a.dbg = g_dbg;
    auto clamp = [](size_t b) { return (uint32_t)(b < 0x7fffffffu ? b : 0x7fffffffu); };
    a.w3_bytes = clamp(w3_bytes); a.w1n_bytes = clamp(w1n_bytes);
    const int px = c == 256 ? 128 : 256;
    a.tiles = (int)((P + px - 1) / px);
    int prof = -1;
    if (sq_prof_on()) {
        char name[96];
        snprintf(name, sizeof(name), "chainw_%s_c%d_cn%d_P%lld", f16 ? "f16x3" : "bf16x3", c, n2, P);
        prof = sq_prof_begin(name, 2.0 * P * (4.0 * c * c + 4.0 * c * n2), (double)P * 4.0 * (c + 4 * c + 4 * c + n2), stream);
    }
    int rc;
    if (c == 256) rc = f16 ? launch_chainw<256, 256, 4, true>(a, stream) : launch_chainw<256, 256, 4, false>(a, stream);
    else rc = f16 ? launch_chainw<128, 128, 8, true>(a, stream) : launch_chainw<128, 128, 8, false>(a, stream);
    if (prof >= 0) sq_prof_end(prof, stream);
    return rc;
}

// Probe / test entry (tests/test_gpu_x3.py, tools/chainw_probe.py): one launch on caller-provided planes (row-major or K-tile-major weights)
extern "C" int sq_dbg_chain_x3w(int f16, int c, long long P, const void* t2_hi, const void* t2_lo, const void* res_hi, const void* res_lo,
                                void* y_hi, void* y_lo, void* t1n_hi, void* t1n_lo, const void* w3_hi, const void* w3_lo,
                                const void* w1n_hi, const void* w1n_lo, const float* b3, const float* cs3, const float* b1n, const float* cs1n,
                                int w_tiled, void* stream) {
    const uint16_t* t2 = (const uint16_t*)t2_hi; const uint16_t* res = (const uint16_t*)res_hi;
    uint16_t* y = (uint16_t*)y_hi; uint16_t* t1n = (uint16_t*)t1n_hi;
    const uint16_t* w3 = (const uint16_t*)w3_hi; const uint16_t* w1 = (const uint16_t*)w1n_hi;
    const long long plW = (const uint16_t*)w3_lo - w3;
    SQ_REQUIRE((const uint16_t*)w1n_lo - w1 == plW, "dbg_chain_x3w: both weights must share the plane distance");
    const size_t w3b = (size_t)4 * c * c * 2, w1b = (size_t)c * 4 * c * 2;
    return sq_launch_chain_x3w(f16, c, t2, (const uint16_t*)t2_lo - t2, res, (const uint16_t*)res_lo - res, y, (uint16_t*)y_lo - y, t1n, (uint16_t*)t1n_lo - t1n,
                               c, w3, w1, plW, w3b, w1b, b3, cs3, b1n, cs1n, P, w_tiled, (hipStream_t)stream);
}
