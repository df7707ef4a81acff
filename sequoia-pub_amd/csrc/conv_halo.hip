// 3x3 / stride 1 / pad 1 convolution with the INPUT TILE RESIDENT in LDS (bf16): the implicit-GEMM form (gemm.hip,
// gemm_ring.hip) re-gathers the activation rows of a tile once per tap -- nine trips of the same bytes from L2 to LDS,
// and every variant of the engine saturates at ~10 TB/s of that traffic (PMC: 182 MB of HBM-side traffic for a 127 MB
// problem).  Here the tile is 256 CONSECUTIVE flat pixels of the NHWC activation, so the rows all nine taps need are the
// contiguous range [p0 - W - 1, p0 + 256 + W + 1): per 64-channel block they are copied to LDS once (one linear
// LDS-DMA stream) and the MFMA A fragments of tap (dy, dx) are read at row offset dy*W + dx; image borders are a 9-bit
// validity mask per output pixel (an invalid tap reads as zeros).  Only the weights still stream per K-tile:
//   per 256 x 128 output tile and 64-channel block:   A 40 KB (once) + B 9 x 16 KB     instead of     9 x (32 + 16) KB.
// 8 waves (4 x 2, 64 x 64 per wave).  LDS: two input-block buffers (the next block's rows arrive while the current one
// is multiplied) + a four-stage ring of weight tiles (three tiles in flight, counted vmcnt + raw s_barrier as in
// gemm_ring.hip) = 144 KiB, one block per CU.  K is walked channel-block-major (all nine taps of a block, then the next
// block): the fp32 accumulation order differs from the tap-major implicit GEMM, results agree to fp32 rounding.
#include "gemm.h"

#include <cstdlib>
#include <type_traits>

namespace {

constexpr uint32_t OOB = 0x80000000u;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_base, uint32_t voffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_base, 16, voffset, 0, 0, 0);
}
__device__ __forceinline__ u32x4 lds128(const char* p) { return *reinterpret_cast<const u32x4*>(p); }

constexpr int BM = 256, BN = 128;
constexpr int HALO_BYTES = 40960;                 // <= 320 rows of 128 B: 256 + 2 W + 2 rows, W <= 31
constexpr int BT_BYTES = BN * 128;                // one weight tile [128 n][64 k]
constexpr int NB = 4;                             // weight ring stages
constexpr int LDS_BYTES = 2 * HALO_BYTES + NB * BT_BYTES;      // 147456
constexpr int HL = HALO_BYTES / 16 / 512;         // 5 LDS-DMA instructions per thread and input block

template <int TAP> struct TapWait {               // loads issued after weight tile g's own loads when tile g is waited for
    // two newer weight tiles (2 loads each) + the next input block (HL loads) if it was issued in one of the last two
    // iterations (it is issued at tap 4, in front of that iteration's weight tile)
    static constexpr int value = 4 + ((TAP == 5 || TAP == 6) ? HL : 0);
};

__global__ __launch_bounds__(512) void conv_halo_kernel(const GemmArgs p) {
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
    const int ncb = C / 64;
    const int halo0 = p0 - W - 1;
    const int halo_slots = (BM + 2 * W + 2) * 8;

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)p.b_bytes, 0x00020000);

    auto issue_halo = [&](int cb, int buf) {      // rows [halo0, halo0 + HR) x channels [64 cb, 64 cb + 64)
#pragma unroll
        for (int u = 0; u < HL; ++u) {
            const int q = u * 512 + tid;
            const int row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
            const int px = halo0 + row;
            const bool ok = q < halo_slots && px >= 0 && px < p.M;
            glds16(rsA, HB + buf * HALO_BYTES + u * 8192 + wave * 1024, ok ? ((uint32_t)px * (uint32_t)C + (uint32_t)(cb * 64 + c * 8)) * 2u : OOB);
        }
    };
    const int br0 = tid >> 3, bc = (tid & 7) ^ ((br0 >> 1) & 7);
    auto issue_wtile = [&](int cb, int tap, int stage) {     // weights [n0 + n][tap*C + cb*64 + 0..63]
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + br0 + 64 * j;
            glds16(rsB, WR + stage * BT_BYTES + j * 8192 + wave * 1024,
                   ((uint32_t)n * (uint32_t)p.ldb + (uint32_t)(tap * C + cb * 64 + bc * 8)) * 2u);
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
    int fb_off[2][4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = wn * 64 + j * 32 + l31;
            fb_off[j][s] = row * 128 + (((2 * s + lh) ^ ((row >> 1) & 7)) << 4);
        }

    // epilogue operands (bias) before the loop
    constexpr int BN8 = BN / 8, RPI = 512 / BN8, ITER = BM / RPI;
    const int e_c8 = tid % BN8, e_rbase = tid / BN8;
    const int e_n = n0 + e_c8 * 8;
    float bias8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
    if (p.bias) {
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(p.bias + e_n), t1 = *reinterpret_cast<const f32x4*>(p.bias + e_n + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { bias8[e] = t0[e]; bias8[4 + e] = t1[e]; }
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int ntile = ncb * 9;                    // weight tiles of this output tile, g = cb * 9 + tap
    // prologue: input block 0, weight tiles 0, 1, 2
    issue_halo(0, 0);
    issue_wtile(0, 0, 0);
    issue_wtile(0, 1, 1);
    issue_wtile(0, 2, 2);

    auto step = [&](int cb, auto tapc) {
        constexpr int TAP = decltype(tapc)::value;
        const int g = cb * 9 + TAP;
        // wait for weight tile g (and everything older: the input block this tap reads)
        if (g + 2 < ntile) {
            if constexpr (TapWait<TAP>::value == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        } else if (g + 1 < ntile) {
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if constexpr (TAP == 4) {
            if (cb + 1 < ncb) issue_halo(cb + 1, (cb + 1) & 1);       // in FRONT of this iteration's weight tile
            else {                                                     // keep the per-iteration load count uniform
#pragma unroll
                for (int u = 0; u < HL; ++u) glds16(rsA, HB + ((cb + 1) & 1) * HALO_BYTES + u * 8192 + wave * 1024, OOB);
            }
        }
        {
            const int g3 = g + 3;
            if (g3 < ntile) issue_wtile(g3 / 9, g3 % 9, g3 & 3);
        }
        const char* hb = HB + (cb & 1) * HALO_BYTES;
        const char* wt = WR + (g & 3) * BT_BYTES;
        const int off = (TAP / 3 - 1) * W + (TAP % 3 - 1);
        const char* arow[2];
        int asw[2];
        bool aok[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = jc[i] + off;
            arow[i] = hb + j * 128;
            asw[i] = (j >> 1) & 7;
            aok[i] = (mask[i] >> TAP) & 1u;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x4 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = lds128(arow[i] + (((2 * s + lh) ^ asw[i]) << 4));
                if (!aok[i]) fa[i] = u32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = lds128(wt + fb_off[j][s]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    union { u32x4 u; bf16x8 h; } ua, ub;
                    ua.u = fa[i]; ub.u = fb[j];
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.h, ub.h, acc[i][j], 0, 0, 0);
                }
        }
    };
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

    // epilogue: fp32 tile through LDS, bias (+ ReLU), bf16 or fp32 rows out
    float* stage = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int col = wn * 64 + j * 32 + l31;
                stage[row * BN + col] = acc[i][j][r];
            }
    __syncthreads();
    float* c32 = p.out_dtype == SQ_F32 ? reinterpret_cast<float*>(p.C) : nullptr;
    bf16_t* c16 = p.out_dtype == SQ_BF16 ? reinterpret_cast<bf16_t*>(p.C) : nullptr;
#pragma unroll
    for (int u = 0; u < ITER; ++u) {
        const int row = e_rbase + u * RPI;
        const int m = p0 + row;
        if (m >= p.M) continue;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(stage + row * BN + e_c8 * 8);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(stage + row * BN + e_c8 * 8 + 4);
        float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[e] = p.alpha * v[e] + bias8[e];
            if (p.act == SQ_ACT_RELU) v[e] = fmaxf(v[e], 0.f);
        }
        if (c32) {
            float* d = c32 + (long long)m * p.ldc + e_n;
            *reinterpret_cast<f32x4*>(d) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(d + 4) = f32x4{v[4], v[5], v[6], v[7]};
        }
        if (c16)
            *reinterpret_cast<u32x4*>(c16 + (long long)m * p.ldc + e_n) =
                u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
    }
}

}  // namespace

// 3x3 / stride 1 / pad 1 implicit-GEMM argument sets this kernel takes over
bool sq_conv_halo_eligible(const GemmArgs& a, int dtype) {
    if (dtype != SQ_BF16 || !a.conv || a.splitk != 1 || a.batch != 1) return false;
    static int on = -1, min_tiles = 0;
    if (on < 0) {
        const char* e = getenv("SQ_CONV_HALO");
        on = (e && e[0] == '0') ? 0 : 1;
        const char* mt = getenv("SQ_CONV_HALO_MIN_TILES");
        min_tiles = mt ? atoi(mt) : 448;
    }
    if (!on) return false;
    if (a.KW != 3 || a.stride != 1 || a.pad != 1 || a.H != a.OH || a.W != a.OW || a.K != 9 * a.Cin) return false;
    if (a.Cin % 64 || a.N % BN || a.W > 31 || a.W < 3) return false;
    if (a.res || a.rowbias || a.Cpre || a.C2 || a.gelu_grad_of || a.ln64_g || a.act == SQ_ACT_GELU || !a.vec_epi) return false;
    const long long tiles = (long long)((a.M + BM - 1) / BM) * (a.N / BN);
    return tiles >= min_tiles;
}

int sq_launch_conv_halo(const GemmArgs& a, hipStream_t stream) {
    static SqDevOnce attr;       // hipFuncSetAttribute is per device
    if (attr.needed()) {
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)conv_halo_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr.done();
    }
    const int tiles = ((a.M + BM - 1) / BM) * (a.N / BN);
    hipLaunchKernelGGL(conv_halo_kernel, dim3(tiles), dim3(512), LDS_BYTES, stream, a);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}
