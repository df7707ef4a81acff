// Split-mode ("x3") chain of the 28 x 28 and 14 x 14 stages: one launch does, for a tile of 128 consecutive pixels,
//     y   = relu(t2 . w3^T * s3 + b3 + identity)          src/resnet.py:83-91   (conv3 / bn3, += identity, relu)
//     t1' = relu(y . w1'^T * s1' + b1')                   src/resnet.py:75-77 of the NEXT block (conv1 / bn1 / relu)
// on hi / lo planes of 16-bit values (x3_fmt.h), three MFMAs per product -- the counterpart of chain_x3.hip (56 x 56 stage)
// for C = 128 (y has 512 channels) and C = 256 (1024 channels).
//
// Why: unfused, y (4 C channels x 4 bytes per pixel) is written by the expand launch and read back by the next block's
// reduce launch -- 1.6 GB per 1000 patches at 28 x 28, 0.8 GB at 14 x 14, per bottleneck.  Here the reduce is fed from
// LDS: per pixel t2 (4 C bytes) and the identity (16 C) are read, y (16 C) and t1' (4 C) written.
//
// Construction: products are transposed (A = weights, B = pixels: lane = pixel), so a lane's accumulators are runs of 4
// consecutive channels of ITS pixel -- the epilogue needs no cross-lane traffic and writes y as 8-byte pieces into a
// [32 px][64 B] hi / lo image that is at once the B operand of the second product and the source of 16-byte global stores.
// y is produced in slices of 32 channels.  Eight waves = 4 pixel groups of 32 x TWO ROLES, two waves per SIMD:
//   * wave A(pg): holds the group's t2 rows as register fragments (both planes, C / 16 k-steps), computes
//     P1: y^T slice = w3[slice] . t2^T and the epilogue (bias, scale, identity, ReLU, split, in place), and issues EVERY load
//     of the block -- weight units, identity tiles (LDS-DMA, one to three steps ahead) -- so its vmcnt sees loads only, which
//     retire in order: counted waits are exact;
//   * wave B(pg): accumulates P2: t1'^T += w1'[:, slice] . y^T slice one slice behind, and issues EVERY store (y slices, t1')
//     -- it never waits on vmcnt, so HBM write latency stalls nobody (a store in front of a counted wait makes the wait
//     stricter: measured +1.3 us per slice in the one-role form of this kernel, profiles/r05_chainw_probe_one_role_form.txt).
//   One s_barrier per step hands over y slices, ring slots and identity buffers; a step is one slice (C = 128) or half a
//   slice (C = 256: P1 over half of K, P2 over one of its two k-steps), i.e. 24 + 24 MFMAs per SIMD pair either way.
//   A's epilogue VALU work runs under B's MFMAs.
//   * weights stream through a ring of six 16 KiB units (w3 part: [4 k-tiles][32 n][64 B] x 2 planes; w1' part: [C n][32 or
//     64 B] x 2 planes; 16-byte chunks XOR-swizzled on the SOURCE address), requested two steps ahead;
//   * biases and per-channel scales sit in LDS.
// Same K order, same MFMA order per accumulator (a_lo.b_hi, a_hi.b_lo, a_hi.b_hi per 16-deep step, steps ascending) and the
// same epilogue arithmetic as the two gemm_x3.hip launches it replaces: bit-identical results (tests/test_gpu_x3.py).
#include "gemm.h"
#include "x3_fmt.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

extern int g_dbg;                // sq_dbg_set key 1 (gemm.hip)

namespace {

constexpr uint32_t OOB = 0x80000000u;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_base, uint32_t voffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_base, 16, voffset, 0, 0, 0);
}
__device__ __forceinline__ void glds16_nt(__amdgpu_buffer_rsrc_t rsrc, char* lds_base, uint32_t voffset) {      // streaming (read-once) data
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_base, 16, voffset, 0, 0, 2);
}
__device__ __forceinline__ u32x4 lds128(const char* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ u32x2 lds64(const char* p) { return *reinterpret_cast<const u32x2*>(p); }
__device__ __forceinline__ void st_lds64(char* p, u32x2 v) { *reinterpret_cast<u32x2*>(p) = v; }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// n (wave-uniform, even, <= 16): waits until at most n vector-memory operations are outstanding
__device__ __forceinline__ void wait_vm_n(int n) {
    switch (n >> 1) {
        case 0: wait_vm<0>(); break;
        case 1: wait_vm<2>(); break;
        case 2: wait_vm<4>(); break;
        case 3: wait_vm<6>(); break;
        case 4: wait_vm<8>(); break;
        case 5: wait_vm<10>(); break;
        case 6: wait_vm<12>(); break;
        case 7: wait_vm<14>(); break;
        default: wait_vm<16>(); break;
    }
}

struct ChainWArgs {
    const uint16_t* t2; long long plT2;          // [P, C] planes
    const uint16_t* res; long long plRes;        // identity [P, 4 C]
    uint16_t* y; long long plY;                  // [P, 4 C]
    uint16_t* t1n; long long plT1n;              // [P, N2]
    const uint16_t* w3; const uint16_t* w1n; long long plW;     // [4 C, C] / [N2, 4 C]; lo plane plW elements behind
    const float* b3; const float* cs3; const float* b1n; const float* cs1n;
    int P, tiles, tiled;
    uint32_t w3_bytes, w1n_bytes;                // descriptor extents of one weight plane
    int dbg;                                     // ablation switches (tools/chainw_probe.py): 1 no global stores, 2 no identity reads, 4 no first product, 8 no second product, 16 no weight stream, 32 no epilogue, 64 flip the nt policy, 128 nothing (selects the DBG instantiation)
};

template <int C, int N2_> struct ChainWCfg {
    static constexpr int N2 = N2_, N1 = 4 * C, NS = N1 / 32, KS1 = C / 16, NT2 = N2 / 32;
    static constexpr int H = N2 / 128;            // steps per slice: the w1' part of a step is [N2 n][32 / H k] x 2 planes = 16 KiB
    static constexpr int KSH = KS1 / H;           // k-steps of P1 per step (8; 4 for C = 128 with a 256-wide next reduce)
    static constexpr int KTL = C / H / 32;        // k-tiles of the w3 part of a step
    static constexpr int W3R = KTL / 2;           // block-wide LDS-DMA rounds per plane of the w3 part (A waves: 256 threads x 16 B)
    static constexpr int NW3 = 2 * W3R;           // w3 pieces per A thread per step (the w1' part: 4)
    static constexpr int STEPS = H * NS + H;      // A works in steps [0, H NS), B in [H, H NS + H)
    static constexpr int T = 512, PX = 128;
    static constexpr int UNIT = 16384, UH = 8192; // ring unit: both planes / one plane
    static constexpr int RING = 6 * UNIT;         // units of steps i, i+1, i+2: two each (w3 part, w1' part)
    static constexpr int XYW = 4096;              // one pixel group's image of one slice: [32 px][64 B] hi | [32 px][64 B] lo
    static constexpr int XY = 4 * 3 * XYW;        // three slices per group: identity arriving, y being made, y being multiplied / stored
    static constexpr int BIAS_FLOATS = 2 * N1 + 2 * N2;        // b3 | cs3 | b1' | cs1'
    static constexpr int LDS_BYTES = RING + XY + BIAS_FLOATS * 4;
    static_assert((C == 128 && (N2 == 128 || N2 == 256)) || (C == 256 && N2 == 256), "C -> next width: 128 -> 128 | 256 (28 x 28 stage), 256 -> 256 (14 x 14 stage)");
    static_assert(32 * (C / H) * 4 <= UNIT && N2 * (32 / H) * 4 == UNIT && KTL >= 2, "unit sizes");
    static_assert(32 * N2 * 2 * 4 <= RING, "t1' staging (one plane per pixel group) must fit the dead ring");
};

template <int C, int N2T, bool F16, bool DBG>
__global__ __launch_bounds__(512, 2) void chain_x3w_kernel(const ChainWArgs p) {
    using Cfg = ChainWCfg<C, N2T>;
    using Fmt = X3Fmt<F16>;
    constexpr int N1 = Cfg::N1, N2 = Cfg::N2, NS = Cfg::NS, KSH = Cfg::KSH, NT2 = Cfg::NT2, H = Cfg::H, STEPS = Cfg::STEPS;
    constexpr int UNIT = Cfg::UNIT, UH = Cfg::UH, XYW = Cfg::XYW;
    const int dbg = DBG ? p.dbg : 0;              // the ablation switches exist in the DBG instantiation only
    // Streaming policy (nt) for the read-once identity tiles and the write-once y lines: keeps the re-read weights in L2.  Measured
    // (tools/chainw_probe.py, profiles/r05_chainw_probe_roles_nt.txt): C = 256, whole-line stores: 856 -> 785 us; C = 128, whose y leaves in
    // half lines: 1093 -> 1282 us -- so only the 256-channel form uses it (dbg 64 flips the choice in the DBG instantiation)
    const bool nt = (H == 2) != ((dbg & 64) != 0);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const RG = smem;
    char* const XYB = smem + Cfg::RING;
    float* const BS = reinterpret_cast<float*>(smem + Cfg::RING + Cfg::XY);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wave & 3;                      // pixel group; waves 0-3: role A, waves 4-7: role B
    const int l31 = lane & 31, lh = lane >> 5;
    const int swz = (l31 >> 2) & 3;
    int t;
    {   // each XCD (block id % 8) walks a contiguous run of tiles
        const int b = blockIdx.x, q = p.tiles >> 3, r = p.tiles & 7, xcd = b & 7, idx = b >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int p0 = t * Cfg::PX + pg * 32;         // this pixel group's first pixel
    char* const myXY = XYB + pg * 3 * XYW;

    // biases / scales -> LDS (all threads), visible after the first barrier
    {
        constexpr int R1 = N1 / 512, R2 = (N2 + 511) / 512;
        float tb[R1], ts[R1];
#pragma unroll
        for (int r = 0; r < R1; ++r) { tb[r] = p.b3[r * 512 + tid]; ts[r] = p.cs3 ? p.cs3[r * 512 + tid] : 1.f; }
#pragma unroll
        for (int r = 0; r < R1; ++r) { BS[r * 512 + tid] = tb[r]; BS[N1 + r * 512 + tid] = ts[r]; }
#pragma unroll
        for (int r = 0; r < R2; ++r) {
            const int i = r * 512 + tid;
            if (i < N2) { BS[2 * N1 + i] = p.b1n[i]; BS[2 * N1 + N2 + i] = p.cs1n ? p.cs1n[i] : 1.f; }
        }
    }

    if (wave < 4) {
        // =========================================== role A ===========================================
        const int ta = tid;                       // 0 .. 255
        const uint32_t act_bytes = (uint32_t)p.P * N1 * 2u;
        const auto rsRh = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, (int)act_bytes, 0x00020000);
        const auto rsRl = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res + p.plRes), 0, (int)act_bytes, 0x00020000);
        const auto rsW3h = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3, 0, (int)p.w3_bytes, 0x00020000);
        const auto rsW3l = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w3 + p.plW), 0, (int)p.w3_bytes, 0x00020000);
        const auto rsW1h = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1n, 0, (int)p.w1n_bytes, 0x00020000);
        const auto rsW1l = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w1n + p.plW), 0, (int)p.w1n_bytes, 0x00020000);

        // ---- the two units of step i -> ring slots 2 (i % 3) and 2 (i % 3) + 1, one LDS-DMA request (piece) at a time so that the
        // requests can be dealt between the MFMAs of a step (an LDS-DMA request costs the issuing wave ~100 cycles of issue time).
        // Pieces 0-3: w3 part (slice i / H, k-tiles [4 (i % H), +4): image row = (local k-tile, n)), hi plane rounds 0-1, lo plane 2-3.
        // Pieces 4-7: w1' part (step i - H of role B: slice (i - H) / H, k range 32 / H wide at 16 ((i - H) % H): image row = n, 64-byte
        // (H = 1) or 32-byte (H = 2) rows).  unit_pieces(i): how many of them exist (per thread: one load each).
        auto unit_has3 = [&](int i) { return !(dbg & 16) && i < H * NS; };
        auto unit_has1 = [&](int i) { return !(dbg & 16) && i >= H && i < STEPS; };
        constexpr int NW3 = Cfg::NW3, W3R = Cfg::W3R, KTL = Cfg::KTL, NPC = NW3 + 4;      // pieces per thread per step
        auto issue_piece = [&](int i, int r) {        // r compile-time after unrolling
            char* base = RG + (i % 3) * (2 * UNIT) + wave * 1024;
            if (r < NW3) {
                if (!unit_has3(i)) return;
                const int s = i / H, h = i % H;
                const int pl = r / W3R, q = (r % W3R) * 256 + ta;                 // (the plane of a round is a compile-time fact: guide T20)
                const int row = q >> 2, cl = (q & 3) ^ ((row >> 2) & 3);
                const int kt = h * KTL + (row >> 5), nn = s * 32 + (row & 31);
                const uint32_t off = p.tiled ? (uint32_t)(((kt * N1 + nn) * 32 + cl * 8) * 2) : (uint32_t)((nn * C + kt * 32 + cl * 8) * 2);
                if (pl == 0) glds16(rsW3h, base + (r % W3R) * 4096, off); else glds16(rsW3l, base + UH + (r % W3R) * 4096, off);
            } else {
                if (!unit_has1(i)) return;
                const int j = i - H, s = j / H, h = j % H, rr = r - NW3;
                const int q = (rr & 1) * 256 + ta;
                uint32_t off;
                if constexpr (H == 1) {
                    const int row = q >> 2, cl = (q & 3) ^ ((row >> 2) & 3);
                    off = p.tiled ? (uint32_t)(((s * N2 + row) * 32 + cl * 8) * 2) : (uint32_t)((row * N1 + s * 32 + cl * 8) * 2);
                } else {
                    const int row = q >> 1, cl = (q & 1) ^ ((row >> 3) & 1);
                    off = p.tiled ? (uint32_t)(((s * N2 + row) * 32 + h * 16 + cl * 8) * 2) : (uint32_t)((row * N1 + s * 32 + h * 16 + cl * 8) * 2);
                }
                if (rr < 2) glds16(rsW1h, base + UNIT + rr * 4096, off); else glds16(rsW1l, base + UNIT + rr * 4096, off);
            }
        };
        auto issue_units = [&](int i) {
#pragma unroll
            for (int r = 0; r < NPC; ++r) issue_piece(i, r);
        };
        // identity of slice s -> this group's image s % 3: 4 loads per thread, piece r = (16-row half u = r >> 1, plane r & 1)
        auto identity_piece = [&](int s, int r) {
            char* dst = myXY + (s % 3) * XYW;
            const int u = r >> 1;
            const int e = u * 64 + lane, row = e >> 2, cl = (e & 3) ^ ((row >> 2) & 3);
            const int pr = p0 + row;
            const uint32_t off = (pr < p.P && !(dbg & 2)) ? ((uint32_t)pr * N1 + (uint32_t)(s * 32 + cl * 8)) * 2u : OOB;
            if (nt) { if (r & 1) glds16_nt(rsRl, dst + 2048 + u * 1024, off); else glds16_nt(rsRh, dst + u * 1024, off); }
            else { if (r & 1) glds16(rsRl, dst + 2048 + u * 1024, off); else glds16(rsRh, dst + u * 1024, off); }
        };
        auto issue_identity = [&](int s) {
#pragma unroll
            for (int r = 0; r < 4; ++r) identity_piece(s, r);
        };
        issue_identity(0);
        issue_units(0);
        issue_units(1);
        asm volatile("" ::: "memory");
        // this group's 32 pixels of t2 as the K-step operands of the first product (lane = pixel l31, K half lh), both planes
        u32x4 xh[Cfg::KS1], xl[Cfg::KS1];
        {
            const int pr = p0 + l31;
            const bool ok = pr < p.P;
            const uint16_t* row = p.t2 + (size_t)(ok ? pr : 0) * C + lh * 8;
#pragma unroll
            for (int ks = 0; ks < Cfg::KS1; ++ks) {
                if (nt) {       // (read once: streaming policy like the identity tiles)
                    xh[ks] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(row + ks * 16));
                    xl[ks] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(row + p.plT2 + ks * 16));
                } else {
                    xh[ks] = *reinterpret_cast<const u32x4*>(row + ks * 16);
                    xl[ks] = *reinterpret_cast<const u32x4*>(row + p.plT2 + ks * 16);
                }
                if (!ok) { xh[ks] = u32x4{0u, 0u, 0u, 0u}; xl[ks] = u32x4{0u, 0u, 0u, 0u}; }
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // everything requested so far has landed (this thread's part)
        __builtin_amdgcn_s_barrier();             // biases, the units of steps 0 and 1, identity 0

        f32x16 accy;
        int prev_id = 0;                          // identity loads requested in the previous step
        for (int s = 0; s <= NS; ++s) {
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const int i = s * H + h;
                // requests of this step: the units of step i + 2 (their slots held step i - 1's, consumed before the barrier just passed),
                // and at the first step of a slice the identity of slice s + 1 (its image held y(s - 2), last read in step i - 1) --
                // dealt one per k-step between the MFMAs below
                // (the identity goes out at the LAST step of a slice, behind the unit pieces: loads retire in order, and an HBM read in
                // front of the L2-resident weight pieces would hold their count up)
                const bool want_id = h == H - 1 && s + 1 < NS;
                const int k = (unit_has3(i + 2) ? NW3 : 0) + (unit_has1(i + 2) ? 4 : 0) + (want_id ? 4 : 0);
                if (s < NS && !(dbg & 4)) {
                    // ---- P1, K part h: y^T slice [32 ch][32 px] (+)= w3[32 s ..][K part] . t2^T
                    if (h == 0) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) accy[e] = 0.f;
                    }
                    const char* ub = RG + (i % 3) * (2 * UNIT) + l31 * 64;
                    u32x4 wh = lds128(ub + ((lh ^ swz) << 4)), wl = lds128(ub + UH + ((lh ^ swz) << 4));
#pragma unroll
                    for (int ks = 0; ks < KSH; ++ks) {
                        u32x4 nh = wh, nl = wl;
                        if (ks + 1 < KSH) {
                            const int off = ((ks + 1) >> 1) * 2048 + (((2 * ((ks + 1) & 1) + lh) ^ swz) << 4);
                            nh = lds128(ub + off); nl = lds128(ub + UH + off);
                        }
                        Fmt::mma(wh, xl[h * KSH + ks], accy);       // activation lo . weight hi
                        Fmt::mma(wl, xh[h * KSH + ks], accy);       // activation hi . weight lo
                        Fmt::mma(wh, xh[h * KSH + ks], accy);       // activation hi . weight hi
                        {   // this k-step's share of the step's requests: the unit pieces first, the identity pieces behind them
                            constexpr int PPK = (NPC + 4 + KSH - 1) / KSH;
#pragma unroll
                            for (int j = ks * PPK; j < (ks + 1) * PPK; ++j) {
                                if (j < NPC) issue_piece(i + 2, j);
                                else if (j < NPC + 4 && want_id) identity_piece(s + 1, j - NPC);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        wh = nh; wl = nl;
                    }
                } else {
                    if (h == 0 && s < NS) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) accy[e] = 0.f;
                    }
                    issue_units(i + 2);
                    if (want_id) issue_identity(s + 1);
                }
                asm volatile("" ::: "memory");
                // the units of step i + 1 have landed (requested in step i - 1, in front of that step's identity pieces).  H = 2: the
                // identity requested in step i - 1 may stay in flight (its epilogue is two steps away); H = 1: it is this step's
                wait_vm_n(k + (H == 2 ? prev_id : 0));
                prev_id = want_id ? 4 : 0;
                if (s < NS && h == H - 1 && !(dbg & 32)) {
                    // ---- epilogue: lane (px = l31, half lh) holds channels 32 s + 8 g + 4 lh + (0..3); identity in, y out (in place)
                    char* xrow = myXY + (s % 3) * XYW + l31 * 64;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        char* a = xrow + ((g ^ swz) << 4) + 8 * lh;
                        const int cb = s * 32 + 8 * g + 4 * lh;
                        const f32x4 b = *reinterpret_cast<const f32x4*>(BS + cb), sc = *reinterpret_cast<const f32x4*>(BS + N1 + cb);
                        const u32x2 ih = lds64(a), il = lds64(a + 2048);
                        float v[4];
                        v[0] = x3_relu((sc[0] * accy[4 * g + 0] + b[0]) + Fmt::sum_lo(ih[0], il[0]));
                        v[1] = x3_relu((sc[1] * accy[4 * g + 1] + b[1]) + Fmt::sum_hi(ih[0], il[0]));
                        v[2] = x3_relu((sc[2] * accy[4 * g + 2] + b[2]) + Fmt::sum_lo(ih[1], il[1]));
                        v[3] = x3_relu((sc[3] * accy[4 * g + 3] + b[3]) + Fmt::sum_hi(ih[1], il[1]));
                        const uint32_t h0 = Fmt::pack2(v[0], v[1]), h1 = Fmt::pack2(v[2], v[3]);
                        const uint32_t l0 = Fmt::rest2(v[0], v[1], h0);
                        const uint32_t l1 = Fmt::rest2(v[2], v[3], h1);
                        st_lds64(a, u32x2{h0, h1});
                        st_lds64(a + 2048, u32x2{l0, l1});
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
    } else {
        // =========================================== role B ===========================================
        const uint32_t act_bytes = (uint32_t)p.P * N1 * 2u;
        const auto rsYh = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)act_bytes, 0x00020000);
        const auto rsYl = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + p.plY), 0, (int)act_bytes, 0x00020000);
        const auto rsTh = __builtin_amdgcn_make_buffer_rsrc((void*)p.t1n, 0, (int)((uint32_t)p.P * N2 * 2u), 0x00020000);
        const auto rsTl = __builtin_amdgcn_make_buffer_rsrc((void*)(p.t1n + p.plT1n), 0, (int)((uint32_t)p.P * N2 * 2u), 0x00020000);
        f32x16 acc2[NT2];
#pragma unroll
        for (int i = 0; i < NT2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[i][e] = 0.f;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // (the bias writes)
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int h = 0; h < H; ++h) __builtin_amdgcn_s_barrier();        // steps 0 .. H-1: nothing to multiply yet
        for (int s = 0; s < NS; ++s) {
            const char* xy = myXY + (s % 3) * XYW;
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const int i = (s + 1) * H + h;
                // y -> global at the first step of a slice, dealt between the MFMA batches (a store request costs issue time like an LDS-DMA
                // request).  H = 1: slice s, 64-byte runs per pixel and plane (the group's image read linearly).  H = 2: at odd s the PAIR
                // (s - 1, s) as 128-byte runs -- whole cache lines: a half-line store makes the L2 fetch the line first.
                const bool st = h == 0 && !(dbg & 1) && (H == 1 || (s & 1));
                constexpr int NSP = H == 1 ? 4 : 8;      // store pieces: H = 1: (16-row half, plane); H = 2: (plane, 8-row quarter)
                u32x4 sv[4];
                auto store_off = [&](int r) -> uint32_t {
                    if constexpr (H == 1) {
                        const int e = (r >> 1) * 64 + lane, row = e >> 2, cl = (e & 3) ^ ((row >> 2) & 3);
                        const int pr = p0 + row;
                        return pr < p.P ? ((uint32_t)pr * N1 + (uint32_t)(s * 32 + cl * 8)) * 2u : OOB;
                    } else {
                        const int row = (r & 3) * 8 + (lane >> 3), c8 = lane & 7;
                        const int pr = p0 + row;
                        return pr < p.P ? ((uint32_t)pr * N1 + (uint32_t)((s - 1) * 32 + c8 * 8)) * 2u : OOB;
                    }
                };
                auto store_read = [&](int r) {            // piece r -> sv[r & 3]
                    if (!st) return;
                    if constexpr (H == 1) {
                        const int e = (r >> 1) * 64 + lane;
                        sv[r & 3] = lds128(xy + (r & 1) * 2048 + e * 16);
                    } else {
                        const int row = (r & 3) * 8 + (lane >> 3), c8 = lane & 7;
                        const char* img = myXY + ((c8 >> 2) ? s % 3 : (s - 1) % 3) * XYW;
                        sv[r & 3] = lds128(img + (r >> 2) * 2048 + row * 64 + (((c8 & 3) ^ ((row >> 2) & 3)) << 4));
                    }
                };
                auto store_piece = [&](int r) {
                    if (!st) return;
                    const bool lo = H == 1 ? (r & 1) : (r >> 2);
                    if (nt) {
                        if (lo) __builtin_amdgcn_raw_buffer_store_b128(sv[r & 3], rsYl, store_off(r), 0, 2);
                        else __builtin_amdgcn_raw_buffer_store_b128(sv[r & 3], rsYh, store_off(r), 0, 2);
                    } else {
                        if (lo) __builtin_amdgcn_raw_buffer_store_b128(sv[r & 3], rsYl, store_off(r), 0, 0);
                        else __builtin_amdgcn_raw_buffer_store_b128(sv[r & 3], rsYh, store_off(r), 0, 0);
                    }
                };
#pragma unroll
                for (int r = 0; r < 4; ++r) store_read(r);
                if (!(dbg & 8)) {
                    // ---- P2: t1'^T [N2][32 px] += w1'[:, k part] . y^T slice, k part = the step's 16 (H = 2) or 32 (H = 1) channels
                    const char* ub = RG + (i % 3) * (2 * UNIT) + UNIT;
#pragma unroll
                    for (int kk = 0; kk < 2 / H; ++kk) {
                        const int ks2 = H == 2 ? h : kk;
                        const int co = ((2 * ks2 + lh) ^ swz) << 4;
                        const u32x4 yh = lds128(xy + l31 * 64 + co), yl = lds128(xy + 2048 + l31 * 64 + co);
                        u32x4 wh[NT2], wl[NT2];
#pragma unroll
                        for (int nt = 0; nt < NT2; ++nt) {
                            const char* wp = H == 2 ? ub + (nt * 32 + l31) * 32 + ((lh ^ ((l31 >> 3) & 1)) << 4) : ub + (nt * 32 + l31) * 64 + co;
                            wh[nt] = lds128(wp); wl[nt] = lds128(wp + UH);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int nt = 0; nt < NT2; ++nt) Fmt::mma(wh[nt], yl, acc2[nt]);
                        if (kk == 0) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) store_piece(r);
                            if (NSP == 8) {
#pragma unroll
                                for (int r = 4; r < 8; ++r) store_read(r);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int nt = 0; nt < NT2; ++nt) Fmt::mma(wl[nt], yh, acc2[nt]);
                        if (kk == 0 && NSP == 8) {
#pragma unroll
                            for (int r = 4; r < 8; ++r) store_piece(r);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int nt = 0; nt < NT2; ++nt) Fmt::mma(wh[nt], yh, acc2[nt]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) store_piece(r);
                    if (NSP == 8) {
#pragma unroll
                        for (int r = 4; r < 8; ++r) store_read(r);
#pragma unroll
                        for (int r = 4; r < 8; ++r) store_piece(r);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
        __syncthreads();                          // (matches role A's) the ring is dead
        // ---- t1' = relu(acc2 * s1' + b1') -> planes, one plane at a time through this group's piece of the dead ring:
        // [32 px][N2 x 2 B], 16-byte chunk ^= px & (chunks per row - 1)
        constexpr int ROWB = N2 * 2, NCH = ROWB / 16;
        char* const mine = smem + pg * (32 * ROWB);
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
                lo_keep[nt][g] = u32x2{Fmt::rest2(v[0], v[1], h0), Fmt::rest2(v[2], v[3], h1)};
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
                    const int e = u * 64 + lane, row = e / NCH, cl = (e % NCH) ^ (row & (NCH - 1));
                    const int pr = p0 + row;
                    const uint32_t off = pr < p.P ? ((uint32_t)pr * N2 + (uint32_t)(cl * 8)) * 2u : OOB;
                    const u32x4 v = lds128(mine + e * 16);
                    if (nt && (dbg & 256)) {      // (experiment: t1' with the streaming policy too)
                        if (pl == 0) __builtin_amdgcn_raw_buffer_store_b128(v, rsTh, off, 0, 2);
                        else __builtin_amdgcn_raw_buffer_store_b128(v, rsTl, off, 0, 2);
                    } else {
                        if (pl == 0) __builtin_amdgcn_raw_buffer_store_b128(v, rsTh, off, 0, 0);
                        else __builtin_amdgcn_raw_buffer_store_b128(v, rsTl, off, 0, 0);
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the plane's reads are done before the next plane overwrites the stage
        }
        return;
    }
    __syncthreads();                              // role A: matches role B's barrier in front of the t1' stage
}

template <int C, int N2, bool F16>
int launch_chainw(const ChainWArgs& a, hipStream_t stream) {
    using Cfg = ChainWCfg<C, N2>;
    static SqDevOnce attr;       // hipFuncSetAttribute is per device
    if (attr.needed()) {
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)chain_x3w_kernel<C, N2, F16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)chain_x3w_kernel<C, N2, F16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
        attr.done();
    }
    if (a.dbg) hipLaunchKernelGGL((chain_x3w_kernel<C, N2, F16, true>), dim3(a.tiles), dim3(Cfg::T), Cfg::LDS_BYTES, stream, a);
    else hipLaunchKernelGGL((chain_x3w_kernel<C, N2, F16, false>), dim3(a.tiles), dim3(Cfg::T), Cfg::LDS_BYTES, stream, a);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}

}  // namespace

// Which (C, next width) pairs the launch covers: the plain bottlenecks of layer 2 (128 -> 512 -> 128) and layer 3 (256 -> 1024 -> 256)
bool sq_chain_x3w_eligible(int c, int n2) { return (c == 128 && (n2 == 128 || n2 == 256)) || (c == 256 && n2 == 256); }

// t2 [P, C], res / y [P, 4 C], t1n [P, n2] as hi / lo planes (pl* = elements between the planes);
// w3 [4 C, C] and w1n [n2, 4 C] planes plW apart (row-major, or K-tile-major when w_tiled), biases / per-channel scales fp32 (scales may be null).
// w3_bytes / w1n_bytes: bytes from the pointer to the end of one weight plane's allocation (descriptor extent).
int sq_launch_chain_x3w(int f16, int c, const uint16_t* t2, long long plT2, const uint16_t* res, long long plRes, uint16_t* y, long long plY,
                        uint16_t* t1n, long long plT1n, int n2, const uint16_t* w3, const uint16_t* w1n, long long plW, size_t w3_bytes, size_t w1n_bytes,
                        const float* b3, const float* cs3, const float* b1n, const float* cs1n, long long P, int w_tiled, hipStream_t stream) {
    SQ_REQUIRE(sq_chain_x3w_eligible(c, n2), "chain_x3w: C=%d next width %d (128 -> 128 | 256, 256 -> 256)", c, n2);
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
    a.tiles = (int)((P + 127) / 128);
    int prof = -1;
    if (sq_prof_on()) {
        char name[96];
        snprintf(name, sizeof(name), "chainw_%s_c%d_cn%d_P%lld", f16 ? "f16x3" : "bf16x3", c, n2, P);
        prof = sq_prof_begin(name, 2.0 * P * (4.0 * c * c + 4.0 * c * n2), (double)P * 4.0 * (c + 4 * c + 4 * c + n2), stream);
    }
    int rc;
    if (c == 256) rc = f16 ? launch_chainw<256, 256, true>(a, stream) : launch_chainw<256, 256, false>(a, stream);
    else if (n2 == 256) rc = f16 ? launch_chainw<128, 256, true>(a, stream) : launch_chainw<128, 256, false>(a, stream);
    else rc = f16 ? launch_chainw<128, 128, true>(a, stream) : launch_chainw<128, 128, false>(a, stream);
    if (prof >= 0) sq_prof_end(prof, stream);
    return rc;
}

// Probe / test entry (tests/test_gpu_x3.py, tools/chainw_probe.py): one launch on caller-provided planes (row-major or K-tile-major weights)
extern "C" int sq_dbg_chain_x3w(int f16, int c, int n2, long long P, const void* t2_hi, const void* t2_lo, const void* res_hi, const void* res_lo,
                                void* y_hi, void* y_lo, void* t1n_hi, void* t1n_lo, const void* w3_hi, const void* w3_lo,
                                const void* w1n_hi, const void* w1n_lo, const float* b3, const float* cs3, const float* b1n, const float* cs1n,
                                int w_tiled, void* stream) {
    const uint16_t* t2 = (const uint16_t*)t2_hi; const uint16_t* res = (const uint16_t*)res_hi;
    uint16_t* y = (uint16_t*)y_hi; uint16_t* t1n = (uint16_t*)t1n_hi;
    const uint16_t* w3 = (const uint16_t*)w3_hi; const uint16_t* w1 = (const uint16_t*)w1n_hi;
    const long long plW = (const uint16_t*)w3_lo - w3;
    SQ_REQUIRE((const uint16_t*)w1n_lo - w1 == plW, "dbg_chain_x3w: both weights must share the plane distance");
    const size_t w3b = (size_t)4 * c * c * 2, w1b = (size_t)n2 * 4 * c * 2;
    return sq_launch_chain_x3w(f16, c, t2, (const uint16_t*)t2_lo - t2, res, (const uint16_t*)res_lo - res, y, (uint16_t*)y_lo - y, t1n, (uint16_t*)t1n_lo - t1n,
                               n2, w3, w1, plW, w3b, w1b, b3, cs3, b1n, cs1n, P, w_tiled, (hipStream_t)stream);
}
