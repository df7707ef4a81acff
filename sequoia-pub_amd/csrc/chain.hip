// Fused 1x1 chain of ResNet-50's 28 x 28 stage (bf16):   y = relu(t2 . w3^T + b3 + identity)   (Bottleneck.conv3 / bn3 /
// += identity / relu, src/resnet.py:83-91)   followed by the NEXT block's   t1' = relu(y . w1'^T + b1')   (conv1 / bn1 /
// relu, :75-77) in one launch: y is written to HBM once and never read back -- per pixel 2 x (C + 4C + 4C + CN) bytes
// instead of 2 x (C + 4C + 4C) + 2 x (4C + CN).  Unfused, the expand 1x1 runs at 4.2 TB/s and the reduce at 3.9 TB/s
// of fabric traffic: both are bound by bytes, not by the matrix pipes.
//
// Same construction as the fused tail of the 56 x 56 stage (bottleneck.hip) without its 3x3 stage:
//   * tile = 32 NW pixels, NW = 4 or 8 waves, wave w = rows [32w, 32w + 32) x all channels; products transposed (lane =
//     pixel);
//   * the t2 tile [128][C = 128] stays in LDS; y is produced in slices of 64 channels: slice = t2 . w3[slice]^T
//     (16 MFMAs per wave), + bias + identity slice (LDS-DMA'd into the wave's own rows, replaced in place by y),
//     stored with 16-byte row-major accesses, then immediately contracted with w1'[:, slice] into the t1' accumulators;
//   * weights stream through two 16 KiB buffers, one barrier per chunk; NW = 4: 80 KiB per block, two blocks per CU
//     (used for a 128-wide next layer); NW = 8: 128 KiB, one block per CU, half the weight re-streaming (256-wide);
//   * outputs leave through buffer descriptors: rows past P are dropped by the range check, no branch per store.
// Bit-identical to the two GEMM launches it replaces (same bf16 rounding of y, same ascending-K accumulation).
#include "gemm.h"

#include <cstdio>
#include <cstdlib>

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

struct ChainArgs {
    const bf16_t* t2;      // [P, 128]
    const bf16_t* res;     // [P, 512]
    bf16_t* y;             // [P, 512]
    bf16_t* t1n;           // [P, CN]
    const bf16_t* w3;      // [512, 128]
    const bf16_t* w1n;     // [CN, 512]
    const float* b3; const float* b1n;
    int P, tiles;
    uint32_t w3_bytes, w1n_bytes;
};

constexpr int C = 128, C4 = 512;
constexpr int WB_BYTES = 16384;
constexpr int lds_bytes(int nw) { return nw * 32 * 256 + nw * 32 * 128 + 2 * WB_BYTES; }

// NW waves = 32 NW pixels per tile.  NW = 4: 80 KiB, two blocks per CU; NW = 8: 128 KiB, one block per CU -- the same
// eight waves per CU, but every tile streams all the weights, so the larger tile halves the bytes re-loaded from L2.
template <int CN, int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void btl_chain_kernel(const ChainArgs p) {
    constexpr int TP = 32 * NW;
    constexpr int A1_BYTES = TP * 256;           // t2 tile, 256-byte rows (chunk ^= row & 15)
    constexpr int XY_BYTES = TP * 128;           // identity -> y slice of 64 channels, 128-byte rows (chunk ^= (row >> 1) & 7)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const A1 = smem;
    char* const XY = smem + A1_BYTES;
    char* const WB = smem + A1_BYTES + XY_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    int t;
    {
        const int b = blockIdx.x, q = p.tiles >> 3, r = p.tiles & 7, xcd = b & 7, idx = b >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int p0 = t * TP;
    const auto rsT2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.t2, 0, p.P * C * 2, 0x00020000);
    const auto rsRes = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, p.P * C4 * 2, 0x00020000);
    const auto rsW3 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3, 0, (int)p.w3_bytes, 0x00020000);
    const auto rsW1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1n, 0, (int)p.w1n_bytes, 0x00020000);
    // stores go through descriptors as well: rows past P are dropped by the range check, no branch per piece
    const auto rsY = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, p.P * C4 * 2, 0x00020000);
    const auto rsT1n = __builtin_amdgcn_make_buffer_rsrc((void*)p.t1n, 0, p.P * CN * 2, 0x00020000);

    // chunk stream: per 64-channel slice s of y:  B_s = w3 rows [64 s, 64 s + 64) as [64 n][256 B];
    //                                             C_s,h = w1' rows [128 h, 128 h + 128), columns [64 s, 64 s + 64) as [128 n][128 B]
    constexpr int NH = CN / 128;                 // halves of the t1' channels
    constexpr int CPS = 1 + NH;
    constexpr int NCHUNK = (C4 / 64) * CPS;
    auto issue_chunk = [&](int id, int buf) {
        char* dst = WB + buf * WB_BYTES + wave * 1024;
        const int s = id / CPS, which = id % CPS;
#pragma unroll
        for (int u = 0; u < 16 / NW; ++u) {
            const int q = u * 64 * NW + tid;
            if (which == 0) {
                const int n = q >> 4, c = (q & 15) ^ (n & 15);
                glds16(rsW3, dst + u * 1024 * NW, (uint32_t)((s * 64 + n) * C + c * 8) * 2u);
            } else {
                const int n = q >> 3, c = (q & 7) ^ ((n >> 1) & 7);
                glds16(rsW1, dst + u * 1024 * NW, (uint32_t)(((which - 1) * 128 + n) * C4 + s * 64 + c * 8) * 2u);
            }
        }
    };

    // prologue: this wave's 32 rows of t2, first weight chunk
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int q = u * 64 + lane;
        const int row = wave * 32 + (q >> 4), c = (q & 15) ^ (row & 15);
        const int pr = p0 + row;
        glds16(rsT2, A1 + (wave * 32 + u * 4) * 256, pr < p.P ? (uint32_t)(pr * C + c * 8) * 2u : OOB);
    }
    issue_chunk(0, 0);

    const int m = wave * 32 + l31;
    f32x16 acc1n[CN / 32];
#pragma unroll
    for (int i = 0; i < CN / 32; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc1n[i][e] = 0.f;

#pragma unroll
    for (int s = 0; s < C4 / 64; ++s) {
        const int idB = s * CPS;
        // ---- y slice = t2 . w3[64 s ..]^T --------------------------------------------------------------------
        __syncthreads();                             // chunk B landed (and, s == 0, the t2 tile); previous chunk consumed
        issue_chunk(idB + 1, (idB + 1) & 1);
#pragma unroll
        for (int u = 0; u < 4; ++u) {                // identity slice -> this wave's rows of XY
            const int q = u * 64 + lane;
            const int row = wave * 32 + (q >> 3), c = (q & 7) ^ ((row >> 1) & 7);
            const int pr = p0 + row;
            glds16(rsRes, XY + (wave * 32 + u * 8) * 128, pr < p.P ? (uint32_t)(pr * C4 + s * 64 + c * 8) * 2u : OOB);
        }
        f32x16 accy[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) accy[i][e] = 0.f;
        {
            const char* wb = WB + (idB & 1) * WB_BYTES;
            const char* arow = A1 + m * 256;
            const int sw = m & 15;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const u32x4 x = lds128(arow + (((2 * ks + lh) ^ sw) << 4));
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int n = nt * 32 + l31;
                    accy[nt] = mma(lds128(wb + n * 256 + (((2 * ks + lh) ^ (n & 15)) << 4)), x, accy[nt]);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's identity rows are in LDS
        {
            char* row = XY + m * 128;
            const int sw = (m >> 1) & 7;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n0 = nt * 32 + 8 * g + 4 * lh;
                    const f32x4 b = *reinterpret_cast<const f32x4*>(p.b3 + s * 64 + n0);
                    char* a = row + (((nt * 4 + g) ^ sw) << 4) + 8 * lh;
                    const u32x2 xr = lds64(a);
                    float v[4];
                    v[0] = accy[nt][4 * g + 0] + b[0] + __uint_as_float(xr[0] << 16);
                    v[1] = accy[nt][4 * g + 1] + b[1] + __uint_as_float(xr[0] & 0xffff0000u);
                    v[2] = accy[nt][4 * g + 2] + b[2] + __uint_as_float(xr[1] << 16);
                    v[3] = accy[nt][4 * g + 3] + b[3] + __uint_as_float(xr[1] & 0xffff0000u);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    st_lds64(a, u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])});
                }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {                // y slice -> global, 128-byte runs per pixel
            const int q = u * 64 + lane;
            const int row = wave * 32 + (q >> 3), c = (q & 7) ^ ((row >> 1) & 7);
            const int pr = p0 + row;
            const u32x4 v = lds128(XY + (wave * 32 + u * 8) * 128 + lane * 16);
            __builtin_amdgcn_raw_buffer_store_b128(v, rsY, (uint32_t)(pr * C4 + s * 64 + c * 8) * 2u, 0, 0);
        }
        // ---- t1' += y slice . w1'[:, 64 s ..]^T ----------------------------------------------------------------
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const int idC = idB + 1 + h;
            __syncthreads();
            if (idC + 1 < NCHUNK) issue_chunk(idC + 1, (idC + 1) & 1);
            const char* wb = WB + (idC & 1) * WB_BYTES;
            const char* arow = XY + m * 128;
            const int sw = (m >> 1) & 7;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const u32x4 x = lds128(arow + (((2 * ks + lh) ^ sw) << 4));
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int n = nt * 32 + l31;
                    acc1n[h * 4 + nt] = mma(lds128(wb + n * 128 + (((2 * ks + lh) ^ ((n >> 1) & 7)) << 4)), x, acc1n[h * 4 + nt]);
                }
            }
        }
    }

    // t1' = relu(acc + b1') -> global, 128 channels at a time through this wave's rows of the (dead) t2 tile
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        char* row = A1 + m * 256;
        const int sw = m & 15;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n0 = nt * 32 + 8 * g + 4 * lh;
                const f32x4 b = *reinterpret_cast<const f32x4*>(p.b1n + h * 128 + n0);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc1n[h * 4 + nt][4 * g + e] + b[e], 0.f);
                st_lds64(row + (((nt * 4 + g) ^ sw) << 4) + 8 * lh, u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])});
            }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int q = u * 64 + lane;
            const int r = wave * 32 + (q >> 4), c = (q & 15) ^ (r & 15);
            const int pr = p0 + r;
            const u32x4 v = lds128(A1 + (wave * 32 + u * 4) * 256 + lane * 16);
            __builtin_amdgcn_raw_buffer_store_b128(v, rsT1n, (uint32_t)(pr * CN + h * 128 + c * 8) * 2u, 0, 0);
        }
    }
}

}  // namespace

// t2 [P, 128], res / y [P, 512], t1n [P, cn] (cn = 128 or 256), bf16.  w*_bytes: extents to the end of the weight buffer.
int sq_launch_bottleneck_chain_c128(const bf16_t* t2, const bf16_t* res, bf16_t* y, bf16_t* t1n, int cn, const bf16_t* w3,
                                    const bf16_t* w1n, size_t w3_bytes, size_t w1n_bytes, const float* b3, const float* b1n,
                                    long long P, hipStream_t stream) {
    SQ_REQUIRE(cn == 128 || cn == 256, "bottleneck chain: next width %d (128 or 256)", cn);
    SQ_REQUIRE(P > 0 && P * C4 * 2 < (1ll << 31), "bottleneck chain: %lld pixels exceed the 2 GiB descriptor limit", P);
    SQ_REQUIRE(w3_bytes >= (size_t)C4 * C * 2 && w1n_bytes >= (size_t)cn * C4 * 2, "bottleneck chain: weight extents");
    ChainArgs a;
    a.t2 = t2; a.res = res; a.y = y; a.t1n = t1n; a.w3 = w3; a.w1n = w1n; a.b3 = b3; a.b1n = b1n;
    const int nw = cn == 256 ? 8 : 4;               // measured against the other count: 304 vs 321 us / 249 vs 257 us
    const int tp = 32 * nw;
    a.P = (int)P; a.tiles = (int)((P + tp - 1) / tp);
    auto clamp = [](size_t b) { return (uint32_t)(b < 0x7fffffffu ? b : 0x7fffffffu); };
    a.w3_bytes = clamp(w3_bytes); a.w1n_bytes = clamp(w1n_bytes);
    static SqDevOnce attr;       // hipFuncSetAttribute is per device
    if (attr.needed()) {
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)btl_chain_kernel<128, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(4)));
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)btl_chain_kernel<256, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(4)));
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)btl_chain_kernel<128, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(8)));
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)btl_chain_kernel<256, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(8)));
        attr.done();
    }
    int prof = -1;
    if (sq_prof_on()) {
        char name[96];
        snprintf(name, sizeof(name), "btl_chain_c128_cn%d_P%lld", cn, P);
        prof = sq_prof_begin(name, 2.0 * P * (128.0 * 512 + 512.0 * cn), (double)P * 2.0 * (128 + 512 + 512 + cn) + 2.0 * (512 * 128 + cn * 512), stream);
    }
    if (nw == 4) {
        if (cn == 128) hipLaunchKernelGGL((btl_chain_kernel<128, 4>), dim3(a.tiles), dim3(256), lds_bytes(4), stream, a);
        else hipLaunchKernelGGL((btl_chain_kernel<256, 4>), dim3(a.tiles), dim3(256), lds_bytes(4), stream, a);
    } else {
        if (cn == 128) hipLaunchKernelGGL((btl_chain_kernel<128, 8>), dim3(a.tiles), dim3(512), lds_bytes(8), stream, a);
        else hipLaunchKernelGGL((btl_chain_kernel<256, 8>), dim3(a.tiles), dim3(512), lds_bytes(8), stream, a);
    }
    SQ_LAUNCH_CHECK();
    if (prof >= 0) sq_prof_end(prof, stream);
    return SQ_OK;
}
