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

namespace {

constexpr uint32_t OOB = 0x80000000u;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_base, uint32_t voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_base, 16, voffset, soffset, 0, 0);
}
__device__ __forceinline__ u32x4 lds_read128(const char* p) { return *reinterpret_cast<const u32x4*>(p); }

constexpr int BM = 256, BK = 32, ROWB = 64;       // 64-byte LDS rows (32 bf16)
constexpr int WTM = 2;
constexpr int A_BYTES = BM * ROWB;                // 16 KiB per plane per stage

template <int WTN> struct X3Cfg {
    static constexpr int BN = 64 * WTN;
    static constexpr int B_BYTES = BN * ROWB;
    static constexpr int STAGE_BYTES = 2 * A_BYTES + 2 * B_BYTES;     // 40 / 48 KiB
    static constexpr int NSTAGE = 3;
    static constexpr int EPI_BYTES = BM * BN * 4;                      // fp32 tile staged through the idle ring
    static constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES > EPI_BYTES ? NSTAGE * STAGE_BYTES : EPI_BYTES;
    static constexpr int LOADS = 4 + (WTN == 1 ? 1 : 2);               // LDS-DMA instructions per thread per stage
};

template <int WTN, bool CONV, bool F16>
__global__ __launch_bounds__(512) void gemm_x3_kernel(const GemmArgs p) {
    using Cfg = X3Cfg<WTN>;
    using Fmt = X3Fmt<F16>;
    constexpr int BN = Cfg::BN, B_BYTES = Cfg::B_BYTES, STAGE_BYTES = Cfg::STAGE_BYTES, NSTAGE = Cfg::NSTAGE;
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

    const bf16_t* Ah = reinterpret_cast<const bf16_t*>(p.A);
    const bf16_t* Bh = reinterpret_cast<const bf16_t*>(p.B);
    const auto rsAh = __builtin_amdgcn_make_buffer_rsrc((void*)Ah, 0, (int)p.a_bytes, 0x00020000);
    const auto rsAl = __builtin_amdgcn_make_buffer_rsrc((void*)(Ah + p.plA), 0, (int)p.a_bytes, 0x00020000);
    const auto rsBh = __builtin_amdgcn_make_buffer_rsrc((void*)Bh, 0, (int)p.b_bytes, 0x00020000);
    const auto rsBl = __builtin_amdgcn_make_buffer_rsrc((void*)(Bh + p.plB), 0, (int)p.b_bytes, 0x00020000);

    // loader geometry: a wave instruction fills 16 consecutive 64-byte LDS rows; 512 threads = one round of 128 rows
    const int r0 = tid >> 2;                            // row inside a 128-row round
    const int gc = (tid & 3) ^ ((r0 >> 2) & 3);         // 16-byte chunk of the SOURCE row this lane fetches
    uint32_t a_off[2];
    int a_ih0[2], a_iw0[2];
    uint32_t a_pix[2];
    bool a_ok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + r0 + 128 * j;
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
    // B rows: WTN == 2: one round per plane (128 rows each); WTN == 1: rows 0-63 of the round are the hi plane, 64-127 the lo plane
    uint32_t b_off;
    const bool b_lo_half = WTN == 1 && r0 >= 64;        // wave-uniform (waves 4-7)
    {
        const int n = n0 + (WTN == 1 ? (r0 & 63) : r0);
        b_off = n < p.N ? ((uint32_t)n * (uint32_t)p.ldb + (uint32_t)(gc * 8)) * 2u : OOB;
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
                glds16(rsAh, sa + j * (128 * ROWB), off, 0);
                glds16(rsAl, sa + A_BYTES + j * (128 * ROWB), off, 0);
            }
            const uint32_t ob = k_ok ? b_off + (uint32_t)(k0 * 2) : OOB;
            if constexpr (WTN == 1) {
                if (b_lo_half) glds16(rsBl, sb, ob, 0); else glds16(rsBh, sb, ob, 0);
            } else {
                glds16(rsBh, sb, ob, 0);
                glds16(rsBl, sb + B_BYTES, ob, 0);
            }
        } else {
            const int soff = k0 * 2;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint32_t off = k_ok ? a_off[j] : OOB;
                glds16(rsAh, sa + j * (128 * ROWB), off, soff);
                glds16(rsAl, sa + A_BYTES + j * (128 * ROWB), off, soff);
            }
            const uint32_t ob = k_ok ? b_off : OOB;
            if constexpr (WTN == 1) {
                if (b_lo_half) glds16(rsBl, sb, ob, soff); else glds16(rsBh, sb, ob, soff);
            } else {
                glds16(rsBh, sb, ob, soff);
                glds16(rsBl, sb + B_BYTES, ob, soff);
            }
        }
    };

    // epilogue operands that do not depend on the accumulators are requested before the K loop
    constexpr int BN8 = BN / 8;
    constexpr int RPI = 512 / BN8;            // rows per epilogue iteration (32 / 64)
    constexpr int ITER = BM / RPI;            // 8 / 4
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
    auto compute = [&](int buf) {
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
    if (nk > 1) issue_loads(1, 1);
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
        compute(cur);
        cur = cur == NSTAGE - 1 ? 0 : cur + 1;
        nxt2 = nxt2 == NSTAGE - 1 ? 0 : nxt2 + 1;
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
#pragma unroll
    for (int c0 = 0; c0 < ITER; c0 += U) {
        // residual rows of a group are requested before that group's first store (vmcnt counts stores too)
        u32x4 rh[U], rl[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            rh[u] = u32x4{0, 0, 0, 0}; rl[u] = u32x4{0, 0, 0, 0};
            const int m = m0 + e_rbase + (c0 + u) * RPI;
            if (resh && m < p.M) {
                rh[u] = *reinterpret_cast<const u32x4*>(resh + (long long)m * p.ldres + e_n);
                rl[u] = *reinterpret_cast<const u32x4*>(resh + p.plRes + (long long)m * p.ldres + e_n);
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
            for (int e = 0; e < 8; ++e) v[e] = scale8[e] * v[e] + bias8[e];
            if (resh) {
                float idn[8];
                x3_join8<F16>(rh[u], rl[u], idn);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += idn[e];
            }
            if (p.act == SQ_ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
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
}

template <int WTN, bool F16>
int launch_x3(const GemmArgs& a, hipStream_t stream) {
    using Cfg = X3Cfg<WTN>;
    static bool attr = false;
    if (!attr) {
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_x3_kernel<WTN, false, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_x3_kernel<WTN, true, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
        attr = true;
    }
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + Cfg::BN - 1) / Cfg::BN);
    const dim3 grid(tiles), block(512);
    if (a.conv) hipLaunchKernelGGL((gemm_x3_kernel<WTN, true, F16>), grid, block, Cfg::LDS_BYTES, stream, a);
    else hipLaunchKernelGGL((gemm_x3_kernel<WTN, false, F16>), grid, block, Cfg::LDS_BYTES, stream, a);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}

}  // namespace

// C = act(alpha * A.B^T + bias + res): A, B (and res, and C unless out_dtype == SQ_F32) as hi / lo bf16 planes;
// a.A / a.B / a.res / a.C point at the hi plane, the lo plane sits plA / plB / plRes / plC ELEMENTS behind it.
int sq_launch_gemm_x3(const GemmArgs& a, hipStream_t stream) {
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
    if (a.conv) SQ_REQUIRE(a.Cin % BK == 0, "conv_x3: Cin=%d must be a multiple of the K-tile (%d)", a.Cin, BK);
    else SQ_REQUIRE(a.lda % 8 == 0, "gemm_x3: lda=%d must be a multiple of 8", a.lda);
    int prof = -1;
    if (sq_prof_on()) {
        const double flops = 2.0 * a.M * (double)a.N * a.K;      // algorithmic (fp32-equivalent) work; the kernel issues 3x that in bf16 MFMAs
        const double a_elems = a.conv ? (double)a.M / (a.OH * a.OW) * a.H * a.W * a.Cin : (double)a.M * a.K;
        const double bytes = (a_elems + (double)a.N * a.K) * 4.0 + (double)a.M * a.N * (4.0 + (a.res ? 4.0 : 0.0));
        char name[96];
        snprintf(name, sizeof(name), "%s_%s_M%d_N%d_K%d", a.conv ? "conv" : "gemm", a.x3_f16 ? "f16x3" : "bf16x3", a.M, a.N, a.K);
        prof = sq_prof_begin(name, flops, bytes, stream);
    }
    int rc;
    if (a.x3_f16) rc = a.N % 128 == 0 ? launch_x3<2, true>(a, stream) : launch_x3<1, true>(a, stream);
    else rc = a.N % 128 == 0 ? launch_x3<2, false>(a, stream) : launch_x3<1, false>(a, stream);
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
