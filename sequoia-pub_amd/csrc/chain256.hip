// Fused 1x1 chain of ResNet-50's 14 x 14 stage (bf16), the 256-plane counterpart of chain.hip:
//   y = relu(t2 . w3^T + b3 + identity)   (Bottleneck.conv3 / bn3 / += identity / relu, src/resnet.py:83-91)   and the NEXT
//   block's   t1' = relu(y . w1'^T + b1')   (conv1 / bn1 / relu, :75-77)   in one launch.
// Unfused, the expand 1x1 (K = 256, N = 1024) moves 2 x (256 + 1024 + 1024) bytes per pixel at 3.3 TB/s and is the
// largest single kernel of the embedder; the reduce reads y back.  Fused, y is written once and never read.
//
// Construction (differs from chain.hip because a [128][256] t2 tile + [128][1024] y do not fit two blocks per CU):
//   * tile = 128 pixels, 8 waves = 4 pixel groups x 2 channel halves; wave (pg, h) owns pixels [32 pg, 32 pg + 32);
//     products are transposed (lane = pixel), so a lane's accumulators are 4 consecutive channels of its pixel;
//   * the wave's t2 rows never touch LDS: its 16 K-step fragments are loaded once into 64 VGPRs;
//   * y is produced in slices of 64 channels.  Wave (pg, h) computes channels [32 h, 32 h + 32) of the slice
//     (16 MFMAs), adds bias + identity (LDS-DMA'd one slice AHEAD into the wave's private [32][64 B] region of a
//     two-slice buffer -- HBM latency is a whole slice away -- and replaced in place by y);
//     after one barrier the pair's two regions are the K = 64 operand of the second product, of which the wave
//     accumulates t1' channels [128 h, 128 h + 128) (16 MFMAs), and are stored to HBM as 128-byte runs per pixel;
//   * weights stream through a three-deep ring of 32 KiB chunks (w3 rows of the slice / w1' columns of the slice),
//     two chunks in flight, counted s_waitcnt + raw s_barrier (global stores only make the counted wait stricter);
//     133 KiB of LDS, one block of 8 waves per CU.
// Bit-identical to the two GEMM launches it replaces (same bf16 rounding of y, same ascending-K accumulation).
#include "gemm.h"

#include <cstdio>

namespace {

constexpr uint32_t OOB = 0x80000000u;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_base, uint32_t voffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_base, 16, voffset, 0, 0, 0);
}
__device__ __forceinline__ u32x4 lds128(const char* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ u32x2 lds64(const char* p) { return *reinterpret_cast<const u32x2*>(p); }
__device__ __forceinline__ void st_lds64(char* p, u32x2 v) { *reinterpret_cast<u32x2*>(p) = v; }
__device__ __forceinline__ f32x16 mma(const u32x4& w, const u32x4& x, f32x16 acc) {
    union { u32x4 u; bf16x8 h; } a, b;
    a.u = w; b.u = x;
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, acc, 0, 0, 0);
}

struct Chain256Args {
    const bf16_t* t2;      // [P, 256]
    const bf16_t* res;     // [P, 1024]
    bf16_t* y;             // [P, 1024]
    bf16_t* t1n;           // [P, 256]
    const bf16_t* w3;      // [1024, 256]
    const bf16_t* w1n;     // [256, 1024]
    const float* b3; const float* b1n;
    int P, tiles;
    uint32_t w3_bytes, w1n_bytes;
};

constexpr int C = 256, C4 = 1024, CN = 256;
constexpr int XY1_BYTES = 128 * 128;         // one slice: [pixel group 4][half 2][32 rows][64 B]  (chunk ^= (row >> 2) & 3)
constexpr int XY_BYTES = 2 * XY1_BYTES;      // two slices: the identity of slice s+1 is fetched while slice s is worked on
constexpr int WB_BYTES = 32768;
constexpr int NBUF = 3;
constexpr int BIAS_BYTES = (C4 + CN) * 4;    // b3 | b1' as fp32: read with ds_read (lgkmcnt), outside the vmcnt bookkeeping
constexpr int LDS_BYTES = XY_BYTES + NBUF * WB_BYTES + BIAS_BYTES;
constexpr int NCHUNK = 2 * (C4 / 64);        // per slice s: chunk 2s = w3 rows [64 s, +64) as [64 n][512 B] (chunk ^= n & 15),
                                             //              chunk 2s+1 = w1' columns [64 s, +64) as [256 n][128 B] (chunk ^= (n >> 1) & 7)

__global__ __launch_bounds__(512, 1) void btl_chain256_kernel(const Chain256Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const XY = smem;
    char* const WB = smem + XY_BYTES;
    const float* const BIAS = reinterpret_cast<const float*>(smem + XY_BYTES + NBUF * WB_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wave >> 1, h = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    int t;
    {
        const int b = blockIdx.x, q = p.tiles >> 3, r = p.tiles & 7, xcd = b & 7, idx = b >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int p0 = t * 128;
    const auto rsRes = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, p.P * C4 * 2, 0x00020000);
    const auto rsY = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, p.P * C4 * 2, 0x00020000);      // rows past P: stores dropped
    const auto rsT1n = __builtin_amdgcn_make_buffer_rsrc((void*)p.t1n, 0, p.P * CN * 2, 0x00020000);
    const auto rsW3 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3, 0, (int)p.w3_bytes, 0x00020000);
    const auto rsW1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1n, 0, (int)p.w1n_bytes, 0x00020000);

    auto issue_chunk = [&](int id) {
        char* dst = WB + (id % NBUF) * WB_BYTES + wave * 1024;
        const int s = id >> 1;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = u * 512 + tid;
            if ((id & 1) == 0) {
                const int n = q >> 5, c = (q & 31) ^ (n & 15);
                glds16(rsW3, dst + u * 8192, (uint32_t)((s * 64 + n) * C + c * 8) * 2u);
            } else {
                const int n = q >> 3, c = (q & 7) ^ ((n >> 1) & 7);
                glds16(rsW1, dst + u * 8192, (uint32_t)(n * C4 + s * 64 + c * 8) * 2u);
            }
        }
    };

    // this wave's 32 pixels of t2 as the 16 K-step operands of the first product (lane = pixel l31, K half lh)
    u32x4 xr[16];
    {
        const int pr = p0 + pg * 32 + l31;
        const bf16_t* row = p.t2 + (size_t)(pr < p.P ? pr : 0) * C + lh * 8;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            xr[ks] = *reinterpret_cast<const u32x4*>(row + ks * 16);
            if (pr >= p.P) xr[ks] = u32x4{0u, 0u, 0u, 0u};
        }
    }
    {
        float* bw = reinterpret_cast<float*>(smem + XY_BYTES + NBUF * WB_BYTES);
        bw[tid] = p.b3[tid];
        bw[512 + tid] = p.b3[512 + tid];
        if (tid < CN) bw[C4 + tid] = p.b1n[tid];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // visible to every wave after the first barrier below
    }
    char* const myXY = XY + (pg * 2 + h) * 2048;
    auto issue_identity = [&](int s) {                      // own half of slice s -> own region of XY[s & 1]
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = u * 16 + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
            const int pr = p0 + pg * 32 + row;
            glds16(rsRes, myXY + (s & 1) * XY1_BYTES + u * 1024, pr < p.P ? (uint32_t)(pr * C4 + s * 64 + h * 32 + c * 8) * 2u : OOB);
        }
    };
    issue_identity(0);
    issue_chunk(0);
    issue_chunk(1);

    f32x16 acc1n[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc1n[i][e] = 0.f;
    const int swp = (l31 >> 2) & 3;

    // per-wave order of the loads:  ... identity s | chunk 2s | [y stores s-1] chunk 2s+1 | identity s+1 | chunk 2s+2 | ...
#pragma unroll
    for (int s = 0; s < C4 / 64; ++s) {
        const int j = 2 * s;
        char* const xy = XY + (s & 1) * XY1_BYTES;
        // ---- y slice (this wave's 32 channels) = t2 . w3[64 s + 32 h ..]^T ----------------------------------------
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");    // chunk j and identity s landed (this wave's part); chunk j+1 may fly
        __builtin_amdgcn_s_barrier();                       // everybody's part; XY[(s+1)&1] and chunk j-1's buffer are free
        if (s + 1 < C4 / 64) issue_identity(s + 1);
        if (j + 2 < NCHUNK) issue_chunk(j + 2);
        f32x16 accy;
#pragma unroll
        for (int e = 0; e < 16; ++e) accy[e] = 0.f;
        {
            const int n = 32 * h + l31;
            const char* wrow = WB + (j % NBUF) * WB_BYTES + n * 512;
            const int sw = n & 15;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) accy = mma(lds128(wrow + (((2 * ks + lh) ^ sw) << 4)), xr[ks], accy);
        }
        {
            char* row = xy + (pg * 2 + h) * 2048 + l31 * 64;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(BIAS + s * 64 + 32 * h + 8 * g + 4 * lh);
                char* a = row + ((g ^ swp) << 4) + 8 * lh;
                const u32x2 xi = lds64(a);
                float v[4];
                v[0] = accy[4 * g + 0] + b[0] + __uint_as_float(xi[0] << 16);
                v[1] = accy[4 * g + 1] + b[1] + __uint_as_float(xi[0] & 0xffff0000u);
                v[2] = accy[4 * g + 2] + b[2] + __uint_as_float(xi[1] << 16);
                v[3] = accy[4 * g + 3] + b[3] + __uint_as_float(xi[1] & 0xffff0000u);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                st_lds64(a, u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])});
            }
        }
        // ---- t1' (this wave's 128 channels) += y slice . w1'[128 h .., 64 s ..]^T -------------------------------
        if (j + 2 < NCHUNK) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");   // chunk j+1 landed; y written
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // both halves of y; everybody's chunk j+1; chunk j consumed
#pragma unroll
        for (int u = 0; u < 2; ++u) {                       // y slice -> global, 128-byte runs per pixel
            const int row = 16 * h + u * 8 + (lane >> 3), c = lane & 7;
            const int pr = p0 + pg * 32 + row;
            const u32x4 v = lds128(xy + (pg * 2 + (c >> 2)) * 2048 + row * 64 + (((c & 3) ^ ((row >> 2) & 3)) << 4));
            __builtin_amdgcn_raw_buffer_store_b128(v, rsY, (uint32_t)(pr * C4 + s * 64 + c * 8) * 2u, 0, 0);   // no branch per piece
        }
        if (j + 3 < NCHUNK) issue_chunk(j + 3);
        {
            const char* wb = WB + ((j + 1) % NBUF) * WB_BYTES;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int kc = 2 * ks + lh;
                const u32x4 x = lds128(xy + (pg * 2 + (kc >> 2)) * 2048 + l31 * 64 + (((kc & 3) ^ swp) << 4));
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int n = 128 * h + 32 * nt + l31;
                    acc1n[nt] = mma(lds128(wb + n * 128 + ((kc ^ ((n >> 1) & 7)) << 4)), x, acc1n[nt]);
                }
            }
        }
    }

    // t1' = relu(acc + b1') -> global through this wave's 8 KiB of the (dead) weight ring: [32 rows][256 B], chunk ^= row & 15
    __syncthreads();
    {
        char* const mine = WB + wave * 8192;
        char* row = mine + l31 * 256;
        const int sw = l31 & 15;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(BIAS + C4 + 128 * h + nt * 32 + 8 * g + 4 * lh);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc1n[nt][4 * g + e] + b[e], 0.f);
                st_lds64(row + (((nt * 4 + g) ^ sw) << 4) + 8 * lh, u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])});
            }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = u * 4 + (lane >> 4), c = (lane & 15) ^ (r & 15);
            const int pr = p0 + pg * 32 + r;
            const u32x4 v = lds128(mine + u * 1024 + lane * 16);
            __builtin_amdgcn_raw_buffer_store_b128(v, rsT1n, (uint32_t)(pr * CN + 128 * h + c * 8) * 2u, 0, 0);
        }
    }
}

}  // namespace

// t2 [P, 256], res / y [P, 1024], t1n [P, 256], bf16.  w*_bytes: extents to the end of the weight buffer.
int sq_launch_bottleneck_chain_c256(const bf16_t* t2, const bf16_t* res, bf16_t* y, bf16_t* t1n, const bf16_t* w3,
                                    const bf16_t* w1n, size_t w3_bytes, size_t w1n_bytes, const float* b3, const float* b1n,
                                    long long P, hipStream_t stream) {
    SQ_REQUIRE(P > 0 && P * C4 * 2 < (1ll << 31), "bottleneck chain: %lld pixels exceed the 2 GiB descriptor limit", P);
    SQ_REQUIRE(w3_bytes >= (size_t)C4 * C * 2 && w1n_bytes >= (size_t)CN * C4 * 2, "bottleneck chain: weight extents");
    Chain256Args a;
    a.t2 = t2; a.res = res; a.y = y; a.t1n = t1n; a.w3 = w3; a.w1n = w1n; a.b3 = b3; a.b1n = b1n;
    a.P = (int)P; a.tiles = (int)((P + 127) / 128);
    auto clamp = [](size_t b) { return (uint32_t)(b < 0x7fffffffu ? b : 0x7fffffffu); };
    a.w3_bytes = clamp(w3_bytes); a.w1n_bytes = clamp(w1n_bytes);
    static SqDevOnce attr;       // hipFuncSetAttribute is per device
    if (attr.needed()) {
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)btl_chain256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr.done();
    }
    int prof = -1;
    if (sq_prof_on()) {
        char name[96];
        snprintf(name, sizeof(name), "btl_chain_c256_cn256_P%lld", P);
        prof = sq_prof_begin(name, 2.0 * P * (256.0 * 1024 + 1024.0 * 256), (double)P * 2.0 * (256 + 1024 + 1024 + 256) + 2.0 * (1024 * 256 + 256 * 1024), stream);
    }
    hipLaunchKernelGGL(btl_chain256_kernel, dim3(a.tiles), dim3(512), LDS_BYTES, stream, a);
    SQ_LAUNCH_CHECK();
    if (prof >= 0) sq_prof_end(prof, stream);
    return SQ_OK;
}
