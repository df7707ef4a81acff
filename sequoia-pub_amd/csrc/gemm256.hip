// 256 x 256 block tile, 8 waves (2 x 4), 128 x 64 per wave -- the NT engine's variant for LARGE products.
//
// With the 4-wave 128 x 128 tile every wave multiplies a 64 x 64 patch: 2 A + 2 B fragments (4 KiB of LDS reads)
// feed 4 MFMAs, i.e. 128 B/clk per CU at full MFMA rate -- the LDS peak.  A 128 x 64 wave patch reads 4 + 2
// fragments for 8 MFMAs (0.75 KiB per MFMA, 96 B/clk).  Same loader (buffer_load ... lds, 128-byte rows, source-side
// XOR swizzle), same two-buffer / one-barrier K loop; the epilogue stages 32-row slabs per wave (the block tile
// would not fit LDS) and goes through the shared epi_apply.  bf16 only, plain or implicit-GEMM (conv) A, no split-K.
//
// Measured (tools/gemm_probe.py big): bare products +20..27 % over the 128 x 128 tile (8192^3: 877 -> 1050 TF;
// 102400 x 1024 x 1024: 631 -> 762; 98000 x 512 x 1024: 622 -> 790).  In the applications it does not pay yet: the
// spatial workload gains 4 %, the patch pipeline loses 4 % -- one 128 KiB block per CU cannot share the CU with a
// block of the other stream's chain, and the un-prefetched epilogue is exposed (no second block to hide it).  So it
// is OPT-IN (SQ_GEMM256=1, or tile 44 through sq_dbg_set) until its epilogue is pipelined.
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

template <int EPI, bool CONV>
__global__ __launch_bounds__(512) void gemm_nt256_kernel(const GemmArgs p) {
    constexpr int BM = 256, BN = 256, BK = 64;
    constexpr int WTM = 4, WTN = 2;                 // 32 x 32 MFMA tiles per wave
    constexpr int RA = BM / 64, RB = BN / 64;       // LDS-DMA instructions per thread per tile
    constexpr int TILE_BYTES = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

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
        char* sa = smem + buf * TILE_BYTES + wave * (8 * 128);
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
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) issue_loads(kt + 1, cur ^ 1);
        compute(cur);
        __syncthreads();
    }

    // epilogue: per wave, slabs of 32 rows x 64 columns through a private 8 KiB LDS region
    float* stage = reinterpret_cast<float*>(smem) + wave * (32 * 64);
    const bool vec = p.vec_epi != 0;
#pragma unroll
    for (int i = 0; i < WTM; ++i) {              // unrolled: a run-time index would push the accumulators to scratch
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                stage[row * 64 + j * 32 + l31] = acc[i][j][r];
            }
        __syncthreads();
#pragma unroll 1
        for (int u = 0; u < 4; ++u) {
            const int row = (lane >> 3) + 8 * u, c8 = lane & 7;
            const int m = m0 + wm * 128 + i * 32 + row, n = n0 + wn * 64 + c8 * 8;
            if (m < p.M && n < p.N) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(stage + row * 64 + c8 * 8);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(stage + row * 64 + c8 * 8 + 4);
                float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                epi_apply<EPI, true>(p, z, m, n, v, min(8, p.N - n), vec);
            }
        }
        __syncthreads();
    }
}

}  // namespace

// true when the 256-tile variant takes the product (large, bf16, enough tiles to keep 256 CUs at one block each busy)
bool sq_gemm256_eligible(const GemmArgs& a, int dtype) {
    if (dtype != SQ_BF16 || a.splitk != 1) return false;
    const long long tiles = (long long)((a.M + 255) / 256) * ((a.N + 255) / 256) * a.batch;
    static int min_tiles = -1;
    if (min_tiles < 0) {
        const char* e = getenv("SQ_GEMM256_MIN_TILES");      // experiment knob; one block per CU -> needs >= 2 full rounds
        min_tiles = e ? atoi(e) : 512;
    }
    return a.K >= 256 && a.N >= 256 && tiles >= min_tiles;
}

namespace {
template <int EPI>
int launch256(const GemmArgs& a, dim3 grid, size_t lds, hipStream_t stream) {
    static SqDevOnce attr;       // hipFuncSetAttribute is per device
    if (attr.needed()) {
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt256_kernel<EPI, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_nt256_kernel<EPI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr.done();
    }
    if (a.conv) hipLaunchKernelGGL((gemm_nt256_kernel<EPI, true>), grid, dim3(512), lds, stream, a);
    else hipLaunchKernelGGL((gemm_nt256_kernel<EPI, false>), grid, dim3(512), lds, stream, a);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}
}  // namespace

int sq_launch_gemm256(const GemmArgs& a, hipStream_t stream) {
    const int tiles = ((a.M + 255) / 256) * ((a.N + 255) / 256);
    const size_t lds = 2 * (256 + 256) * 128;
    const dim3 grid(tiles, 1, a.batch);
    if (a.gelu_grad_of) return launch256<2>(a, grid, lds, stream);
    if (a.act == SQ_ACT_GELU) return launch256<1>(a, grid, lds, stream);
    return launch256<0>(a, grid, lds, stream);
}
