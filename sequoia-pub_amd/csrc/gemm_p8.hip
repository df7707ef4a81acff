// 256 x 256 x 64 block tile, eight waves (2 x 4, 128 x 64 per wave), 16x16x32 MFMAs, TWO LDS buffers of four 16 KiB half-tiles,
// eight phases per pair of K-tiles -- the NT engine's kernel for large bf16 products (UNI ViT-L/16, the spatial path, ViS
// inference at large batch, the ResNet's plain 1x1 products in bf16 mode).
//
// Why another shape (gemm_w4.hip is 256 x 256 x 32, four stages, 32x32x16 MFMAs): a 32-deep K-tile is a 64-byte row -- every
// operand request touches HALF a 128-byte line and the other half is wanted one stage later, when L1 has dropped it
// (K-tile-major operands took gemm_w4 from 829 to 1170 TF on 8192^3, addressing only).  A 64-deep tile
// makes every row a whole line with the operands left row-major, at the price of 128 KiB for TWO buffers; with two buffers
// the load -> use distance comes from splitting the tile instead: a K-tile is four half-tiles (A even / odd 64-row groups,
// B even / odd 32-column groups), each phase reads ONE half-tile's fragments, multiplies ONE 64 x 32 quadrant of the wave's
// patch (16 MFMAs) and re-stages the half-tile whose last read lies TWO phases back -- so every request has six phases
// (~2000 clocks) to land before the single counted wait of a K-tile:
//
//   phase  reads (ds_read_b128)         multiplies     stages (2 LDS-DMA instructions per thread)
//   1      B_0 (4), A_0 (8) of tile t   A_0 x B_0      A_1 of tile t+1  (slot read last in phase 3 of tile t-1)
//   2      B_1 (4)                      A_0 x B_1      B_0 of tile t+1  (read last in phase 4 of tile t-1)
//   3      A_1 (8)                      A_1 x B_1      A_0 of tile t+2  (read last in phase 1)
//   4      B_0 (4)                      A_1 x B_0      B_1 of tile t+2  (read last in phase 2);  s_waitcnt vmcnt(4): tile t+1 has landed
//
// The two groups of four waves (one wave of each per SIMD) run ONE BARRIER apart (waves 4-7 take an extra s_barrier in front
// of the loop, waves 0-3 one behind it): while one group issues its fragment reads and LDS-DMA, the other group's MFMAs own
// the matrix pipes (s_setprio 1 around them), and vice versa.  Never a vmcnt(0) inside the loop.  The fragments are waited for
// BEHIND the phase barrier (their latency overlaps the barrier wait); that is safe because of the two-phase distance: a reader's
// lgkmcnt(0) sits in front of its multiply, and the earliest re-stage of that slot by the other group comes a barrier later.
// (Refilling one phase after the last read with the wait in front of the barrier -- three half-tiles in flight, vmcnt(6) --
// measured the same, +-1 %.)
// LDS image: 1 KiB sub-tiles of [16 rows][32 k], byte ^= ((byte >> 9) & 1) << 5 inside a sub-tile (rows 8-15 swap their
// 32-byte halves): conflict-free 16-lane groups for the 16x16x32 fragment reads; with LDS-DMA the permutation sits on the
// SOURCE address.  Half-tile A_a = block rows with (m >> 6) & 1 == a, B_b = columns with (n >> 5) & 1 == b, so that a wave's
// 128 x 64 patch is contiguous in C although each of its quadrants lives in its own half-tile.
// Tile walk: inside an XCD's contiguous run, 8 tile rows per group, column by column (kernel comment) -- operand panels stay in L2.
// Results are bit-equal to the engine's 32x32x16 kernels (tests/test_gpu_gemm.py).
// Epilogue: per-wave 16 x 64 fp32 slabs through LDS, 16-byte accesses; bias, residual, pre-activation copy, LayerNorm(64), ReLU / GELU,
// GELU' multiply, bf16 copy -- every operand requested before the slab is written.
// PERSISTENT form (one block per CU; block i of the 32 an XCD runs takes tiles i, i + 32, ... of the XCD's run): with K = 1024 a tile's
// 256 x 256 results are a third of its time when every CU stores them in the same few microseconds (tools/gemm_probe.py p8a:
// 50432 x 4096 x 1024 with fp32 results 534 us, 367 us without the stores = an HBM write burst at 4.9 TB/s).  The persistent
// block requests the NEXT tile's first operands before it writes the current tile's results (the slabs live in the 32 KiB
// the two buffers leave free): their latency hides behind the epilogue, +2..5 % on the K = 1024 shapes.  The burst itself
// stays: the next tile's first wait is a counted vmcnt, and a counted wait cannot tell loads from stores -- on gfx950 they
// retire OUT OF ORDER with respect to each other (tools/vmcnt_probe.hip: 90 % of the lanes see a sentinel when a wave
// waits vmcnt(NS) for a load issued in front of NS stores), so it has to cover the stores too.
// Round 5 (tools/gemm_probe.py p8e / p8d, profiles/r05_gemm_p8_probe_*): of the ~90 us the epilogues of 204800 x 1024 x 1024 (bf16
// results) cost, 30 are read-out and arithmetic, 27 the issue of the stores (measured with every tile landing on tile (0, 0)) and
// 30 their HBM traffic.  Keeping the packed rows of a tile's lower half in registers (32 VGPRs: 237 in all, no spills) and
// storing one row group per K-tile under the NEXT tile's K loop was built, bit-equal -- and slower: 443 -> 459 us (50432 x 4096 x
// 1024: 418 -> 417; K = 4096: 396 -> 401): the counted wait of every K-tile then has stores in front of it.  Removed.
#include "gemm.h"
#include "gemm_epi.h"

#include <cstdlib>
#include <type_traits>

namespace {

constexpr uint32_t OOB = 0x80000000u;
typedef __attribute__((address_space(3))) void lds_void;
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_base, uint32_t voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_base, 16, voffset, soffset, 0, 0);
}
__device__ __forceinline__ u32x4 lds_read128(const char* p) { return *reinterpret_cast<const u32x4*>(p); }

constexpr int BM = 256, BK = 64, NT = 512;
constexpr int HALF_A = 128 * BK * 2;             // 16 KiB: 128 rows x 64 k
constexpr int SLAB_BYTES = 16 * 64 * 4;          // one wave's 16 x 64 fp32 slab
// BNT = 256: waves 2 x 4, 128 x 64 per wave, B half-tiles of 128 columns, 128 KiB for the two buffers.
// BNT = 128: waves 4 x 2, 64 x 64 per wave, B half-tiles of 64 columns (one LDS-DMA instruction per thread), 96 KiB: the shape
// for products whose 256 x 256 tiling cannot fill the chip (the ViS training step: 6400 x 1024 x 1024 = 100 tiles of 256 x 256
// but 200 of 256 x 128 -- one per CU, all in one round).
template <int BNT> struct P8Cfg {
    static constexpr int WROWS = BNT == 256 ? 2 : 4, WCOLS = 8 / WROWS;
    static constexpr int MF = 256 / WROWS / 16;          // m-fragments per wave (8 | 4); n-fragments: always 4 (64 columns)
    static constexpr int MA = MF / 2;                    // m-fragments per A half-tile and wave (4 | 2)
    static constexpr int SA = MA * 16;                   // rows of a wave's A sub-tile (64 | 32)
    static constexpr int HALF_B = (BNT / 2) * BK * 2;    // 16 | 8 KiB
    static constexpr int LB = BNT == 256 ? 2 : 1;        // LDS-DMA instructions per thread and B half-tile
    static constexpr int BUF = 2 * HALF_A + 2 * HALF_B;  // A_0 A_1 B_0 B_1: 64 | 48 KiB
    static constexpr int LDS = 2 * BUF;
    static constexpr int WAIT = 2 + LB;                  // loads that may stay in flight behind the wait of phase 4: A_0 and B_1 of the K-tile after next
};

template <int EPI, bool PERSIST, bool DBG, int BNT>
__global__ __launch_bounds__(NT) void gemm_p8_kernel(const GemmArgs p) {
    using Cfg = P8Cfg<BNT>;
    constexpr int WCOLS = Cfg::WCOLS, MF = Cfg::MF, MA = Cfg::MA, SA = Cfg::SA, HALF_B = Cfg::HALF_B, BUF_BYTES = Cfg::BUF, LB = Cfg::LB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WCOLS, wc = wave % WCOLS;
    const int grp = wave >> 2;           // the two groups of four waves (one wave of each per SIMD) that run one barrier apart
    const int dbg = DBG ? p.dbg : 0;     // ablation switches (tools/gemm_probe.py p8a): 1 no stores, 2 no LDS-DMA inside the loop, 4 no MFMA, 8 no fragment reads

    const int tiles_n = (p.N + BNT - 1) / BNT, tiles_m = (p.M + BM - 1) / BM;
    const int nwg = tiles_m * tiles_n;
    const int z = PERSIST ? 0 : blockIdx.z;
    // each XCD (block id % 8) owns a contiguous run of tiles ...
    const int xcd = blockIdx.x & 7;
    auto run_start = [&](int x) { const int q = nwg >> 3, r = nwg & 7; return x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q; };
    auto run_count = [&](int x) { return (nwg >> 3) + (x < (nwg & 7) ? 1 : 0); };
    // ... and inside the run takes the tiles in groups of `gm` tile rows, column by column: the 32 blocks an XCD runs at a time
    // then cover gm row panels x 32 / gm column panels instead of one row panel x 32 column panels -- per K-tile
    // (gm + 32 / gm) x 32 KiB of distinct operand bytes instead of 33 x 32 KiB, so most staging requests hit the XCD's L2
    // (38 TB/s, tools/stage_probe.hip) instead of going to the memory-side cache (~10 TB/s)
    int m0 = 0, n0 = 0;
    auto tile_coords = [&](int t, int& m0_, int& n0_) {
        const int gm = p.tile_group_m > 0 ? p.tile_group_m : 1;
        const int per_group = gm * tiles_n;
        const int g = t / per_group, first_m = g * gm;
        const int gsz = min(tiles_m - first_m, gm);
        const int in_g = t - g * per_group;
        m0_ = (first_m + in_g % gsz) * BM;
        n0_ = (in_g / gsz) * BNT;
    };
    // persistent form: block (xcd, i) of the G blocks an XCD runs takes tiles i, i + G, i + 2 G, ... of the XCD's run -- the order
    // the dispatcher would have handed them out in
    const int G = PERSIST ? (int)(gridDim.x >> 3) : 1;
    int idx = blockIdx.x >> 3;
    if (idx >= run_count(xcd)) return;
    const int t_cur = run_start(xcd) + idx;
    tile_coords(t_cur, m0, n0);

    const bf16_t* Ab = reinterpret_cast<const bf16_t*>(p.A) + (long long)z * p.sA;
    const bf16_t* Bb = reinterpret_cast<const bf16_t*>(p.B) + (long long)z * p.sB;
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, (int)(p.a_bytes - (size_t)z * p.sA * 2), 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, (int)(p.b_bytes - (size_t)z * p.sB * 2), 0x00020000);

    // ---- staging geometry.  A half-tile (128 LDS rows): wave w stages row group w (16 rows), both 32-deep halves -- two
    // instructions that together request the whole 128-byte line of each row.  B half-tile: the same for 128 rows (BNT = 256);
    // 64 rows (BNT = 128) are eight 1 KiB sub-tiles, wave w stages sub-tile w = (row group w >> 1, k half w & 1).
    // Lane l lands at byte 16 l of the 1 KiB sub-tile = LDS row l >> 2, PHYSICAL chunk l & 3; it must hold LOGICAL chunk
    // (l & 3) ^ 2 [rows 8-15].  Half-tile A_a = the rows whose SA-row block index is = a mod 2 (SA = a wave's sub-tile height);
    // B_b = the columns whose 32-column block index is = b mod 2: a wave's patch is contiguous in C.
    const int s_lr = lane >> 2;
    const int s_ck = (lane & 3) ^ ((lane >> 5) << 1);
    const int s_RA = wave * 16 + s_lr;
    const int s_RB = (LB == 2 ? wave : wave >> 1) * 16 + s_lr;
    const int s_khB = wave & 1;                                            // BNT = 128: the k half this wave stages of a B half-tile
    uint32_t a_src[2], b_src[2];
    auto set_src = [&](int m0_, int n0_) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = m0_ + (s_RA / SA) * (2 * SA) + h * SA + (s_RA % SA);
            const int n = n0_ + (s_RB >> 5) * 64 + h * 32 + (s_RB & 31);
            a_src[h] = m < p.M ? ((uint32_t)m * (uint32_t)p.lda + (uint32_t)(s_ck * 8)) * 2u : OOB;
            b_src[h] = n < p.N ? ((uint32_t)n * (uint32_t)p.ldb + (uint32_t)(s_ck * 8)) * 2u : OOB;
        }
    };
    set_src(m0, n0);
    const int nk = (p.K + BK - 1) / BK;
    // slot: 0 A_0, 1 A_1, 2 B_0, 3 B_1.  K-tiles behind the last one (and the ragged end of K: K % 8 == 0, whole chunks) fetch
    // nothing -- the instruction is still issued so that the counted waits stay uniform.
    bool in_loop = false;
    auto stage = [&](int kt, int slot, int buf) {
        if ((dbg & 2) && in_loop) return;
        const int k0 = kt * BK;
        if (slot < 2) {
            char* dst = smem + buf * BUF_BYTES + slot * HALF_A + wave * 2048;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                const bool ok = kt < nk && k0 + kh * 32 + s_ck * 8 < p.K;
                glds16(rsA, dst + kh * 1024, ok ? a_src[slot] : OOB, (k0 + kh * 32) * 2);
            }
        } else if constexpr (LB == 2) {
            char* dst = smem + buf * BUF_BYTES + 2 * HALF_A + (slot - 2) * HALF_B + wave * 2048;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                const bool ok = kt < nk && k0 + kh * 32 + s_ck * 8 < p.K;
                glds16(rsB, dst + kh * 1024, ok ? b_src[slot - 2] : OOB, (k0 + kh * 32) * 2);
            }
        } else {
            char* dst = smem + buf * BUF_BYTES + 2 * HALF_A + (slot - 2) * HALF_B + wave * 1024;
            const bool ok = kt < nk && k0 + s_khB * 32 + s_ck * 8 < p.K;
            glds16(rsB, dst, ok ? b_src[slot - 2] : OOB, (k0 + s_khB * 32) * 2);
        }
    };
    // first operands of a tile: K-tile 0 complete, then what the steady state would have issued during "K-tile -1"
    auto stage_first = [&]() {
        stage(0, 0, 0); stage(0, 2, 0); stage(0, 3, 0); stage(0, 1, 0);
        stage(1, 0, 1); stage(1, 3, 1);                            // A_0, B_1 of K-tile 1
    };

    // ---- fragment geometry (16x16x32: lane = (row l & 15, k group l >> 4), 8 consecutive k = 16 bytes)
    const int f_r = lane & 15, f_kg = lane >> 4;
    const int f_byte = (f_r * 64 + f_kg * 16) ^ ((f_r >> 3) << 5);
    // A_a m-fragment ii: LDS rows wr * SA + ii * 16 -> row group wr * MA + ii;  B_b n-fragment jj (0..1): row group wc * 2 + jj
    const int fa_base = (wr * MA) * 2048 + f_byte, fb_base = 2 * HALF_A + (wc * 2) * 2048 + f_byte;

    f32x4v acc[MF][4];
    u32x4 fa[MA][2] = {}, fb[2][2] = {};
    auto read_a = [&](int buf, int a) {
        if (dbg & 8) return;
        const char* s = smem + buf * BUF_BYTES + a * HALF_A + fa_base;
#pragma unroll
        for (int ii = 0; ii < MA; ++ii)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fa[ii][ks] = lds_read128(s + ii * 2048 + ks * 1024);
    };
    auto read_b = [&](int buf, int b) {
        if (dbg & 8) return;
        const char* s = smem + buf * BUF_BYTES + b * HALF_B + fb_base;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fb[jj][ks] = lds_read128(s + jj * 2048 + ks * 1024);
    };
    auto mma = [&](auto ac, auto bc) {
        constexpr int a = decltype(ac)::value, b = decltype(bc)::value;
        if (dbg & 4) {
#pragma unroll
            for (int ii = 0; ii < MA; ++ii) asm volatile("" ::"v"(fa[ii][0]), "v"(fa[ii][1]));
            asm volatile("" ::"v"(fb[0][0]), "v"(fb[0][1]), "v"(fb[1][0]), "v"(fb[1][1]));
            return;
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int ii = 0; ii < MA; ++ii)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    union { u32x4 u; bf16x8 h; } ua, ub;
                    ua.u = fa[ii][ks]; ub.u = fb[jj][ks];
                    acc[a * MA + ii][b * 2 + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.h, ub.h, acc[a * MA + ii][b * 2 + jj], 0, 0, 0);
                }
        __builtin_amdgcn_s_setprio(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // one phase's ending: the barrier that lets the other group run its load segment; the fragments are waited for BEHIND it
    // (their latency overlaps the barrier wait -- safe because a slot is re-staged TWO phases after its last read); the
    // multiply; the barrier that hands the pipes to the other group
#define P8_SYNC_MMA(AC, BC)                                          \
    __builtin_amdgcn_sched_barrier(0);                               \
    __builtin_amdgcn_s_barrier();                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               \
    __builtin_amdgcn_sched_barrier(0);                               \
    mma(AC{}, BC{});                                                 \
    __builtin_amdgcn_sched_barrier(0);                               \
    __builtin_amdgcn_s_barrier();                                    \
    __builtin_amdgcn_sched_barrier(0)

    // four phases of K-tile kt in buffer `buf` (compile-time).  Last reads of a K-tile's slots: A_0 phase 1, B_1 phase 2, A_1
    // phase 3, B_0 phase 4; each is refilled two phases later: two half-tiles in flight behind the wait of phase 4.
    auto tile_phases = [&](int kt, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        read_b(buf, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(buf, 0);
        stage(kt + 1, 1, buf ^ 1);
        P8_SYNC_MMA(I0, I0);
        read_b(buf, 1);
        stage(kt + 1, 2, buf ^ 1);
        P8_SYNC_MMA(I0, I1);
        read_a(buf, 1);
        stage(kt + 2, 0, buf);
        P8_SYNC_MMA(I1, I1);
        read_b(buf, 0);
        stage(kt + 2, 3, buf);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Cfg::WAIT) : "memory");      // everything up to B_0 of K-tile kt + 1 has landed (this wave's part)
        P8_SYNC_MMA(I1, I0);
    };

    // ---- epilogue geometry.  C/D of the 16x16 MFMA: col = lane & 15, row = 4 (lane >> 4) + r.  Wave patch: MF m-fragments of
    // 16 rows from m0 + wr * (16 MF), columns n0 + wc*64 .. +64; slab = one m-fragment x 64 columns, private to the wave: 4 KiB.
    float* slab = reinterpret_cast<float*>(smem + (PERSIST ? 2 * BUF_BYTES : 0) + wave * SLAB_BYTES);
    const int e_c8 = lane & 7, e_r8 = lane >> 3;         // read-out: 8 lanes cover a slab row (64 columns), 8 rows per pass
    auto epilogue = [&](int m0_, int n0_) {
        const int e_n = n0_ + wc * 64 + e_c8 * 8;
        const int e_cnt = min(8, p.N - e_n);
        const bool fast = p.vec_epi != 0 && e_cnt == 8 && !p.rowbias && ((EPI & 4) != 0) == (p.ln64_g != nullptr) &&
            ((EPI & 2) != 0 || !p.gelu_grad_of);
        float bias8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
        if (fast && p.bias) {
            const float* bsrc = p.bias + (long long)z * p.sBias + e_n;
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(bsrc), t1 = *reinterpret_cast<const f32x4*>(bsrc + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { bias8[e] = t0[e]; bias8[4 + e] = t1[e]; }
        }
        // EPI & 4: per-head LayerNorm(64) + affine between the bias / residual and the activation (src/tformer_lin.py:20-21: the f
        // projection's local_norm): a slab row IS one head -- the wave's 64 columns -- and lives in 8 consecutive lanes
        float lng[8], lnb[8];
        if constexpr ((EPI & 4) != 0) {
            if (fast) {
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.ln64_g + e_n), g1 = *reinterpret_cast<const f32x4*>(p.ln64_g + e_n + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.ln64_b + e_n), b1 = *reinterpret_cast<const f32x4*>(p.ln64_b + e_n + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { lng[e] = g0[e]; lng[4 + e] = g1[e]; lnb[e] = b0[e]; lnb[4 + e] = b1[e]; }
            }
        }
        const float* res32 = (fast && p.res && p.res_dtype == SQ_F32) ? reinterpret_cast<const float*>(p.res) + (long long)z * p.sRes : nullptr;
        const bf16_t* res16 = (fast && p.res && p.res_dtype == SQ_BF16) ? reinterpret_cast<const bf16_t*>(p.res) + (long long)z * p.sRes : nullptr;
        const float* gg32 = ((EPI & 2) && fast && p.gelu_grad_of && p.gg_dtype == SQ_F32) ? reinterpret_cast<const float*>(p.gelu_grad_of) + (long long)z * p.sGg : nullptr;
        const bf16_t* gg16 = ((EPI & 2) && fast && p.gelu_grad_of && p.gg_dtype == SQ_BF16) ? reinterpret_cast<const bf16_t*>(p.gelu_grad_of) + (long long)z * p.sGg : nullptr;
        float* c32 = p.out_dtype == SQ_F32 ? reinterpret_cast<float*>(p.C) + (long long)z * p.sC : nullptr;
        bf16_t* c16p = p.out_dtype == SQ_BF16 ? reinterpret_cast<bf16_t*>(p.C) + (long long)z * p.sC : nullptr;
        float* cpre32 = (p.Cpre && p.pre_dtype == SQ_F32) ? reinterpret_cast<float*>(p.Cpre) + (long long)z * p.sPre : nullptr;
        bf16_t* cpre16 = (p.Cpre && p.pre_dtype == SQ_BF16) ? reinterpret_cast<bf16_t*>(p.Cpre) + (long long)z * p.sPre : nullptr;
        // EPI & 8: the ViS combiner on the slab (GemmArgs::comb_w).  B fragments of this wave's head: n-fragment jj (16 outputs) x k-step
        // ks (32 of the 64 local inputs), lane (n = f_r, k group f_kg) -- 8 KiB of L2-resident weights per wave and tile
        u32x4 wcf[4][2] = {};
        if constexpr ((EPI & 8) != 0) {
            const bf16_t* wsrc = reinterpret_cast<const bf16_t*>(p.comb_w) + (size_t)((n0_ + wc * 64) >> 6) * (64 * 128);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) wcf[jj][ks] = *reinterpret_cast<const u32x4*>(wsrc + (16 * jj + f_r) * 128 + 32 * ks + 8 * f_kg);
        }
        auto ld8 = [&](const float* s32, const bf16_t* s16, long long off, float (&d)[8]) {
            if (s16) {
                const u32x4 tt = *reinterpret_cast<const u32x4*>(s16 + off);
#pragma unroll
                for (int e = 0; e < 4; ++e) { d[2 * e] = __uint_as_float(tt[e] << 16); d[2 * e + 1] = __uint_as_float(tt[e] & 0xffff0000u); }
            } else if (s32) {
                const f32x4 t0 = *reinterpret_cast<const f32x4*>(s32 + off), t1 = *reinterpret_cast<const f32x4*>(s32 + off + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { d[e] = t0[e]; d[4 + e] = t1[e]; }
            }
        };
        auto slab_out = [&](auto ic) {                       // compile-time m-fragment index: a run-time one would push the accumulators to scratch
            constexpr int i = decltype(ic)::value;
            const int mrow0 = m0_ + wr * (MF * 16) + i * 16;
            float aux[2][8], gsrc[2][8];
            if (fast) {                                      // residual / GELU' source rows are requested before the slab is written
#pragma unroll
                for (int u = 0; u < 2; ++u) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { aux[u][e] = 0.f; gsrc[u][e] = 0.f; }
                    const int m = mrow0 + u * 8 + e_r8;
                    if (m < p.M) {
                        ld8(res32, res16, (long long)m * p.ldres + e_n, aux[u]);
                        if constexpr ((EPI & 2) != 0) ld8(gg32, gg16, (long long)m * p.ldgg + e_n, gsrc[u]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) slab[(f_kg * 4 + r) * 64 + j * 16 + f_r] = acc[i][j][r];
            // the slab is private to the wave: its own LDS writes are ordered before its reads (lgkmcnt), no barrier
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int row = u * 8 + e_r8;
                const int m = mrow0 + row;
                if (m >= p.M || e_cnt <= 0 || (dbg & 1)) continue;
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(slab + row * 64 + e_c8 * 8);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(slab + row * 64 + e_c8 * 8 + 4);
                float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                if (!fast) {
                    epi_apply<EPI & 3, true>(p, z, m, e_n, v, e_cnt, p.vec_epi != 0 && e_cnt == 8);
                    continue;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (p.alpha * v[e] + bias8[e]) + aux[u][e];
                if (cpre32) {                                 // the pre-activation value the backward pass wants (f32 or bf16)
                    float* d = cpre32 + (long long)m * p.ldpre + e_n;
                    *reinterpret_cast<f32x4*>(d) = f32x4{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(d + 4) = f32x4{v[4], v[5], v[6], v[7]};
                }
                if (cpre16)
                    *reinterpret_cast<u32x4*>(cpre16 + (long long)m * p.ldpre + e_n) =
                        u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
                if constexpr ((EPI & 4) != 0) {          // two-pass mean / variance as nn.LayerNorm, eps 1e-5 (the arithmetic of gemm.hip's epilogue)
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
                if ((EPI & 1) && p.act == SQ_ACT_GELU) {                     // same erf form as epi_apply<EPI, true>
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = sq_gelu<true>(v[e]);
                } else if (p.act == SQ_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if constexpr ((EPI & 2) != 0) {
                    if (gg32 || gg16) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] *= sq_gelu_grad<true>(gsrc[u][e]);
                    }
                }
                const u32x4 packed = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
                if (DBG && (dbg & 64)) {                      // ablation: the whole epilogue but its global stores (tools/gemm_probe.py p8e)
                    asm volatile("" ::"v"(packed), "v"(v[0]), "v"(v[7]));
                    continue;
                }
                if (DBG && (dbg & 128)) {                     // ablation: every tile's results land on tile (0, 0): the stores without their HBM traffic
                    *reinterpret_cast<u32x4*>(c16p + (long long)(m - m0_) * p.ldc + (e_n - n0_)) = packed;
                    continue;
                }
                if (c32) {
                    float* d = c32 + (long long)m * p.ldc + e_n;
                    *reinterpret_cast<f32x4*>(d) = f32x4{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(d + 4) = f32x4{v[4], v[5], v[6], v[7]};
                }
                // (non-temporal stores for bf16 results: -3 ... -6 % in the probe, nothing in the applications -- the next kernel re-reads the
                // tensor, part of it from the caches such a store bypasses; removed in round 5)
                if (c16p) *reinterpret_cast<u32x4*>(c16p + (long long)m * p.ldc + e_n) = packed;
                if (p.C2) *reinterpret_cast<u32x4*>(p.C2 + (long long)z * p.sC2 + (long long)m * p.ldc2 + e_n) = packed;   // bf16 operand copy
            }
        };
        // one m-fragment through the combiner: LayerNorm(64) + GELU rows -> bf16 A tile on the slab (two [16][32] sub-tiles in the
        // operand buffers' format) -> 8 MFMAs against the head's fragments -> fp32 slab again -> + Cs row, GELU, 16-byte stores.
        // The slab is private to the wave and LDS operations of a wave execute in order: no barrier anywhere.
        auto slab_out_comb = [&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int mrow0 = m0_ + wr * (MF * 16) + i * 16;
            char* const sb = reinterpret_cast<char*>(slab);
            f32x4 rb0[2], rb1[2];                    // the slide's Cs row of this lane's 8 columns: requested before the slab is written
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int m = min(mrow0 + u * 8 + e_r8, p.M - 1);
                const float* rb = p.comb_rb + (long long)(m / p.comb_rpg) * p.comb_ldrb + e_n;
                rb0[u] = *reinterpret_cast<const f32x4*>(rb); rb1[u] = *reinterpret_cast<const f32x4*>(rb + 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) slab[(f_kg * 4 + r) * 64 + j * 16 + f_r] = acc[i][j][r];
            float v[2][8];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int row = u * 8 + e_r8;
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(slab + row * 64 + e_c8 * 8);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(slab + row * 64 + e_c8 * 8 + 4);
                const float t[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[u][e] = p.alpha * t[e] + bias8[e];
                // LayerNorm(64) as in the plain epilogue below (two-pass, eps 1e-5), then GELU: Lf
                float sm = ((v[u][0] + v[u][1]) + (v[u][2] + v[u][3])) + ((v[u][4] + v[u][5]) + (v[u][6] + v[u][7]));
                sm = group8_sum(sm);
                const float mean = sm * (1.0f / 64.0f);
                float q = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { v[u][e] -= mean; q += v[u][e] * v[u][e]; }
                q = group8_sum(q);
                const float rstd = 1.0f / sqrtf(q * (1.0f / 64.0f) + 1e-5f);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[u][e] = sq_gelu<true>(v[u][e] * rstd * lng[e] + lnb[e]);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {            // (behind BOTH read passes: the A tile lies over fp32 rows 0-7)
                const int row = u * 8 + e_r8;
                const u32x4 packed = {pack_bf16x2(v[u][0], v[u][1]), pack_bf16x2(v[u][2], v[u][3]), pack_bf16x2(v[u][4], v[u][5]), pack_bf16x2(v[u][6], v[u][7])};
                *reinterpret_cast<u32x4*>(sb + (e_c8 >> 2) * 1024 + ((row * 64 + (e_c8 & 3) * 16) ^ ((row >> 3) << 5))) = packed;
            }
            const u32x4 af0 = lds_read128(sb + f_byte), af1 = lds_read128(sb + 1024 + f_byte);
            f32x4v o[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                union { u32x4 u; bf16x8 h; } ua, ub;
                o[jj] = f32x4v{0.f, 0.f, 0.f, 0.f};
                ua.u = af0; ub.u = wcf[jj][0];
                o[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.h, ub.h, o[jj], 0, 0, 0);
                ua.u = af1; ub.u = wcf[jj][1];
                o[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.h, ub.h, o[jj], 0, 0, 0);
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int r = 0; r < 4; ++r) slab[(f_kg * 4 + r) * 64 + jj * 16 + f_r] = o[jj][r];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int row = u * 8 + e_r8;
                const int m = mrow0 + row;
                if (m >= p.M || (dbg & 1)) continue;
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(slab + row * 64 + e_c8 * 8);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(slab + row * 64 + e_c8 * 8 + 4);
                const f32x4 r0 = rb0[u], r1 = rb1[u];
                float w8[8] = {a0[0] + r0[0], a0[1] + r0[1], a0[2] + r0[2], a0[3] + r0[3], a1[0] + r1[0], a1[1] + r1[1], a1[2] + r1[2], a1[3] + r1[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) w8[e] = sq_gelu<true>(w8[e]);
                *reinterpret_cast<u32x4*>(c16p + (long long)m * p.ldc + e_n) =
                    u32x4{pack_bf16x2(w8[0], w8[1]), pack_bf16x2(w8[2], w8[3]), pack_bf16x2(w8[4], w8[5]), pack_bf16x2(w8[6], w8[7])};
            }
        };
        if constexpr ((EPI & 8) != 0) {
            slab_out_comb(std::integral_constant<int, 0>{}); slab_out_comb(std::integral_constant<int, 1>{});
            slab_out_comb(std::integral_constant<int, 2>{}); slab_out_comb(std::integral_constant<int, 3>{});
            if constexpr (MF == 8) {
                slab_out_comb(std::integral_constant<int, 4>{}); slab_out_comb(std::integral_constant<int, 5>{});
                slab_out_comb(std::integral_constant<int, 6>{}); slab_out_comb(std::integral_constant<int, 7>{});
            }
            return;
        }
        slab_out(std::integral_constant<int, 0>{}); slab_out(std::integral_constant<int, 1>{});
        slab_out(std::integral_constant<int, 2>{}); slab_out(std::integral_constant<int, 3>{});
        if constexpr (MF == 8) {
            slab_out(std::integral_constant<int, 4>{}); slab_out(std::integral_constant<int, 5>{});
            slab_out(std::integral_constant<int, 6>{}); slab_out(std::integral_constant<int, 7>{});
        }
    };

    stage_first();
    while (true) {
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Cfg::WAIT) : "memory");      // K-tile 0 has landed (persistent form: and the previous tile's stores are out)
        __builtin_amdgcn_s_barrier();
        if (grp == 1) __builtin_amdgcn_s_barrier();         // waves 4-7 run one barrier behind waves 0-3
        in_loop = true;
        for (int kt = 0; kt < nk; kt += 2) {
            tile_phases(kt, I0{});
            if (kt + 1 < nk) tile_phases(kt + 1, I1{});
        }
        in_loop = false;
        if (grp == 0) __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing empty loads have written their zeros
        __syncthreads();                                    // all fragment reads done: the buffers are free
        if constexpr (!PERSIST) {
            epilogue(m0, n0);
            break;
        } else {
            idx += G;
            const int t_next = idx < run_count(xcd) ? run_start(xcd) + idx : -1;
            int m0n = 0, n0n = 0;
            if (t_next >= 0) {                              // the next tile's first operands are requested BEFORE this tile's results are written
                tile_coords(t_next, m0n, n0n);
                set_src(m0n, n0n);
                stage_first();
            }
            epilogue(m0, n0);
            if (t_next < 0) break;
            m0 = m0n; n0 = n0n;
        }
    }
#undef P8_SYNC_MMA
}

}  // namespace

// Which shape of the eight-phase kernel takes the product, if any: 256 (256 x 256 tiles), 128 (256 x 128 tiles), 0 (none).  bf16,
// plain (no convolution view), K in whole 16-byte chunks, no split-K, a vector epilogue without row bias; 256 x 256 when that
// gives (nearly) every CU a tile, else 256 x 128 when THAT does and the product is too small for more than ~1.5 rounds of it.
int g_p8_on = -1;                  // sq_dbg_set key 14 (tests, probes): 0 = off, 1 = on, -1 = default (on)
int sq_gemm_p8_shape(const GemmArgs& a, int dtype) {
    if (dtype != SQ_BF16 || a.conv || a.splitk != 1 || a.rowbias || !a.vec_epi) return 0;
    if (a.ln64_g && (a.N % 64 || !a.ln64_b)) return 0;
    if (a.ln64_g && a.gelu_grad_of) return 0;          // no epilogue of this kernel does both (LayerNorm(64) is forward, GELU' backward)
    // from 176 tiles (tools/gemm_probe.py p8m: ahead of the kernels it replaces from 192 tiles -- 24500 x 512 x 2048: 932 vs 837 TF --,
    // level at 200 x K 1024, behind at 100); sq_dbg_set key 14 = 0 switches the kernel off (tests, probes)
    constexpr int on = 1, min_tiles = 176;
    constexpr int min_k = 512;
    // The 256 x 128 shape is reached through sq_dbg_set key 13 only (tests, probes): on the ViS training step's 6400 x 1024 x 1024
    // products it equals the 128 x 128 kernel in isolation (23.1 vs 23.0 us; its phases hold 8 MFMAs, too few to amortise two barriers)
    // and loses in the step (3.80 vs 3.55 ms): a block that owns a whole CU leaves no room for the weight-gradient products of the helper stream.
    constexpr int on128 = 0;
    if (!(g_p8_on >= 0 ? g_p8_on : on) || a.K % 8 || a.K < min_k) return 0;
    const long long rows = (a.M + BM - 1) / BM;
    // the GELU' epilogue (backward pass) reads a second [M, N] operand per tile: with fewer than two full rounds of tiles the
    // quantisation tail costs more than the main loop gains (M = 19200: 148 us against 115 for the 128 x 128 kernel)
    const long long t256 = rows * (a.N / 256) * a.batch;
    if (a.N % 256 == 0 && t256 >= (a.gelu_grad_of ? 2 * 232 : min_tiles)) return 256;
    const long long t128 = rows * (a.N / 128) * a.batch;
    if (on128 && a.N % 128 == 0 && t128 >= 150 && t128 <= 400) return 128;
    return 0;
}
bool sq_gemm_p8_eligible(const GemmArgs& a, int dtype) { return sq_gemm_p8_shape(a, dtype) != 0; }

int g_p8_group_m = -1;             // sq_dbg_set key 11: tile rows per group of the tile walk (-1 = 8)
int g_p8_sched = -1;               // sq_dbg_set key 10 (probes): 0 = one block per tile, 1 = persistent blocks; -1 = persistent
int g_p8_bn = -1;                  // sq_dbg_set key 13: forced tile width (128 / 256) when the kernel is forced (tile 88); -1 = by shape
namespace {
template <int EPI, bool PERSIST, bool DBG, int BNT>
int launch_p8(const GemmArgs& a, dim3 grid, hipStream_t stream) {
    constexpr int lds = P8Cfg<BNT>::LDS + (PERSIST ? 8 * SLAB_BYTES : 0);
    static SqDevOnce attr;       // hipFuncSetAttribute is per device
    if (attr.needed()) {
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_p8_kernel<EPI, PERSIST, DBG, BNT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr.done();
    }
    hipLaunchKernelGGL((gemm_p8_kernel<EPI, PERSIST, DBG, BNT>), grid, dim3(NT), lds, stream, a);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}
template <bool PERSIST, int BNT>
int launch_p8_pick(const GemmArgs& a, dim3 grid, hipStream_t stream) {
    if (a.comb_w) return launch_p8<13, PERSIST, false, BNT>(a, grid, stream);          // LayerNorm(64) + GELU + the ViS combiner
    if (a.dbg) return a.act == SQ_ACT_GELU ? launch_p8<1, PERSIST, true, BNT>(a, grid, stream) : launch_p8<0, PERSIST, true, BNT>(a, grid, stream);
    if (a.ln64_g) return launch_p8<5, PERSIST, false, BNT>(a, grid, stream);          // LayerNorm(64) [+ GELU when act says so]
    if (a.gelu_grad_of) return launch_p8<2, PERSIST, false, BNT>(a, grid, stream);    // GELU' multiply (backward pass)
    return a.act == SQ_ACT_GELU ? launch_p8<1, PERSIST, false, BNT>(a, grid, stream) : launch_p8<0, PERSIST, false, BNT>(a, grid, stream);
}
int p8_cus() {
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (!cus[dev & 63]) {
        hipDeviceProp_t prop;
        cus[dev & 63] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
    }
    return cus[dev & 63];
}
}  // namespace

int sq_launch_gemm_p8(const GemmArgs& a_in, hipStream_t stream) {
    GemmArgs a = a_in;
    SQ_REQUIRE(!a.ln64_g || (a.ln64_b && a.N % 64 == 0 && a.vec_epi), "gemm_p8: the LayerNorm(64) epilogue needs N %% 64 == 0 and 16-byte aligned epilogue operands");
    SQ_REQUIRE(!a.rowbias, "gemm_p8: no row-bias epilogue");
    SQ_REQUIRE(!(a.ln64_g && a.gelu_grad_of), "gemm_p8: LayerNorm(64) and GELU' epilogues cannot be combined");
    SQ_REQUIRE(a.K % 8 == 0, "gemm_p8: K=%d must be a multiple of 8 (16-byte operand chunks), also when the tile is forced", a.K);
    SQ_REQUIRE(!a.comb_w || (a.comb_rb && a.ln64_g && a.ln64_b && a.act == SQ_ACT_GELU && a.out_dtype == SQ_BF16 && a.N % 256 == 0 && a.vec_epi &&
                             !a.res && !a.rowbias && !a.Cpre && !a.C2 && !a.gelu_grad_of && a.batch == 1 && a.comb_rpg >= 1 && a.comb_ldrb % 4 == 0 &&
                             ((uintptr_t)a.comb_w & 15) == 0 && ((uintptr_t)a.comb_rb & 15) == 0),
               "gemm_p8: the combiner epilogue needs LayerNorm(64) + GELU, bf16 results, N %% 256 == 0, no residual / copies, 16-byte aligned operands");
    constexpr int env_gm = 8, env_persist = 1;
    a.tile_group_m = g_p8_group_m > 0 ? g_p8_group_m : env_gm;
    int bn = sq_gemm_p8_shape(a, SQ_BF16);
    if (g_p8_bn == 128 || g_p8_bn == 256) bn = g_p8_bn;         // probes / tests
    if (bn == 0) bn = 256;                                        // forced (tile 88) on a shape the heuristics would not pick
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + bn - 1) / bn);
    const bool persist = (g_p8_sched >= 0 ? g_p8_sched : env_persist) != 0 && a.batch == 1;
    const int cus = p8_cus() & ~7;
    if (persist && tiles > cus && cus >= 8) {
        // (a start-up skew between the blocks of an XCD, or between the XCDs, to spread the store bursts: slower by more than the delay /
        // within noise -- blocks that drift apart stop sharing operand panels in L2; removed in round 5)
        return bn == 256 ? launch_p8_pick<true, 256>(a, dim3(cus, 1, 1), stream) : launch_p8_pick<true, 128>(a, dim3(cus, 1, 1), stream);
    }
    return bn == 256 ? launch_p8_pick<false, 256>(a, dim3(tiles, 1, a.batch), stream) : launch_p8_pick<false, 128>(a, dim3(tiles, 1, a.batch), stream);
}
