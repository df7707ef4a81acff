// Fused "bottleneck tail" of ResNet-50 layer 1 (56 x 56 maps, 64 -> 64 -> 256 channels), bf16 mode.
//
// One launch does, for a tile of 128 consecutive pixels of the flattened [n*H*W] activation:
//     t2  = relu(conv3x3(t1) + b2)                  src/resnet.py:79-81   (Bottleneck.conv2 / bn2 / relu)
//     y   = relu(t2 . w3^T + b3 + residual)         src/resnet.py:83-91   (conv3 / bn3, += identity, relu)
//     t1' = relu(y . w1'^T + b1')                   src/resnet.py:75-77 of the NEXT block (conv1 / bn1 / relu)
// with eval-mode BatchNorm folded into the weights / biases (resnet.hip).
//
// Why: on the 56 x 56 stage the 1x1 convolutions are bound by the fabric, not by the matrix pipes (0.8 GB per
// 256-channel tensor and sub-batch of 500 patches, ~5 TB/s reached, and chunking the batch so that the tensors fit the
// 256 MiB Infinity Cache buys nothing).  Unfused, a bottleneck moves 16 C bytes per pixel (C = 64 channels x 2 B):
// reduce reads 4C writes C, 3x3 reads C writes C, expand reads C + 4C (identity) and writes 4C.  Here the 3x3 output
// and the 256-channel result feed the next product from LDS: per pixel t1 (C) and the identity (4C) are read, y (4C)
// and t1' (C) written -- 10 C, and three launches become one.
//
// Geometry.  The tile is a run of 128 flat pixels; the 3x3 taps of pixel p are the flat pixels p + dy*W + dx, so the
// input halo of a tile is the CONTIGUOUS row range [p0 - W - 1, p0 + 128 + W + 1) of t1 -- one linear LDS-DMA copy --
// and image borders are a 9-bit validity mask per pixel (an invalid tap contributes a zero fragment).
// Every product is computed transposed (MFMA A operand = weight rows, B operand = pixels): a lane then owns ONE pixel
// and 4 consecutive channels per accumulator quad, so epilogues are 8-byte LDS accesses on the lane's own row, and
// every activation row a wave touches after the 3x3 stage is its own (no barriers between the stages).
// 4 waves, wave w = pixels [32w, 32w+32) x all channels.  Weights stream through two 16 KiB LDS buffers in nine
// (CN = 64) or eleven (CN = 128) chunks per tile, each worth 16 MFMAs per wave, one barrier per chunk; two blocks
// share a CU (80 KiB each), which is what hides the chunk / halo / identity load latencies.
// LDS images are lane-linear (buffer_load ... lds), the XOR swizzles that keep ds_read_b128 conflict-free are applied
// on the source addresses: 128-byte rows chunk ^= (row >> 1) & 7, 256-byte rows chunk ^= row & 15.
//
// Results are bit-identical to the three separate launches of the GEMM engine (same bf16 roundings of t2 / y, same
// ascending-K fp32 accumulation): tests/test_gpu_resnet.py::test_fused_bottleneck_tail_is_bit_identical.
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

struct BtlArgs {
    const bf16_t* t1;      // [P, 64]   input of the 3x3
    const bf16_t* res;     // [P, 256]  identity (the block's input, or its downsample branch)
    bf16_t* y;             // [P, 256]  block output
    bf16_t* t1n;           // [P, CN]   next block's conv1 output
    const bf16_t* w2;      // [64, 576]   k = (kh*3 + kw)*64 + cin
    const bf16_t* w3;      // [256, 64]
    const bf16_t* w1n;     // [CN, 256]
    const float* b2; const float* b3; const float* b1n;
    int P, W, HW, tiles;
    uint32_t w2_bytes, w3_bytes, w1n_bytes;     // descriptor extents (to the end of the packed weight buffer)
    // downsample form (first block of the stage): the identity is  xin . wd^T + bd  (src/resnet.py:87-88), computed
    // here per 128-channel slice from the block's 64-channel input instead of being read as a 256-channel tensor
    const bf16_t* xin;     // [P, 64]
    const bf16_t* wd;      // [256, 64]
    const float* bd;
    uint32_t wd_bytes;
};

constexpr int R0_BYTES = 32768;     // t1 halo rows (128 B each)  /  identity -> y chunk [128 px][128 ch] (256-B rows)
constexpr int R1_BYTES = 16384;     // t2 [128 px][64 ch] (128-B rows); later the t1' staging when CN == 64
constexpr int WB_BYTES = 16384;     // one weight chunk
constexpr int LDS_BYTES = R0_BYTES + R1_BYTES + 2 * WB_BYTES;

template <int CN, bool DS>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void btl_tail_kernel(const BtlArgs p) {   // LDS allows two blocks per CU: do not trade registers for a third
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const R0 = smem;
    char* const R1 = smem + R0_BYTES;
    char* const WB = smem + R0_BYTES + R1_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;

    int t;
    {   // each XCD (block id % 8) walks a contiguous run of tiles: neighbouring tiles share halo rows in its L2
        const int b = blockIdx.x, q = p.tiles >> 3, r = p.tiles & 7, xcd = b & 7, idx = b >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int p0 = t * 128;
    const int W = p.W;
    const int halo0 = p0 - W - 1;                  // flat pixel of LDS row 0 of the halo tile
    const int halo_slots = (128 + 2 * W + 2) * 8;  // 16-byte slots

    const auto rsT1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.t1, 0, p.P * 128, 0x00020000);
    const auto rsRes = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, p.P * 512, 0x00020000);
    const auto rsW2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, (int)p.w2_bytes, 0x00020000);
    const auto rsW3 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3, 0, (int)p.w3_bytes, 0x00020000);
    const auto rsW1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1n, 0, (int)p.w1n_bytes, 0x00020000);
    const auto rsWd = __builtin_amdgcn_make_buffer_rsrc((void*)(DS ? p.wd : p.w3), 0, (int)(DS ? p.wd_bytes : p.w3_bytes), 0x00020000);
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void*)(DS ? p.xin : p.t1), 0, p.P * 128, 0x00020000);
    // stores go through descriptors as well: rows past P are dropped by the range check, no branch per piece (-4...5 %
    // for three of the four instantiations; hipcc schedules the branch-free <64, false> with 168 registers and no
    // fragment read-ahead, +14 %, so that one keeps the guarded stores)
    constexpr bool BRANCHY = CN == 64 && !DS;
    const auto rsY = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, p.P * 512, 0x00020000);
    const auto rsT1n = __builtin_amdgcn_make_buffer_rsrc((void*)p.t1n, 0, p.P * CN * 2, 0x00020000);

    // ---- weight chunk stream -------------------------------------------------------------------------------
    // chunk ids: 0..4 = 3x3 taps (2i, 2i+1), rows [64 n][256 B];  then per 128-channel slice nc of y:
    //   B_nc = w3 rows nc*128.. [128 n][128 B];  C_nc = w1' columns nc*128.. : CN = 64 one chunk [64 n][256 B],
    //   CN = 128 two chunks [128 n][128 B] (k halves)
    //   downsample form: one more chunk in front of B_nc, D_nc = wd rows nc*128.. [128 n][128 B]
    constexpr int NDS = DS ? 1 : 0;
    constexpr int CPS = (CN == 64 ? 2 : 3) + NDS;   // chunks per y slice
    constexpr int NCHUNK = 5 + 2 * CPS;
    auto issue_chunk = [&](int id, int buf) {
        char* dst = WB + buf * WB_BYTES + wave * 1024;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = u * 256 + tid;
            uint32_t off;
            if (id < 5) {                            // [64][256 B]: k = id*128 + chunk*8
                const int n = q >> 4, c = (q & 15) ^ (n & 15);
                off = (uint32_t)(n * 576 + id * 128 + c * 8) * 2u;
                glds16(rsW2, dst + u * 4096, off);
            } else {
                const int s = id - 5, nc = s / CPS, which = s % CPS - NDS;
                if (which < 0) {                     // [128][128 B]: wd row nc*128 + n
                    const int n = q >> 3, c = (q & 7) ^ ((n >> 1) & 7);
                    off = (uint32_t)((nc * 128 + n) * 64 + c * 8) * 2u;
                    glds16(rsWd, dst + u * 4096, off);
                } else if (which == 0) {             // [128][128 B]: w3 row nc*128 + n
                    const int n = q >> 3, c = (q & 7) ^ ((n >> 1) & 7);
                    off = (uint32_t)((nc * 128 + n) * 64 + c * 8) * 2u;
                    glds16(rsW3, dst + u * 4096, off);
                } else if (CN == 64) {               // [64][256 B]: w1' row n, k = nc*128 + chunk*8
                    const int n = q >> 4, c = (q & 15) ^ (n & 15);
                    off = (uint32_t)(n * 256 + nc * 128 + c * 8) * 2u;
                    glds16(rsW1, dst + u * 4096, off);
                } else {                             // [128][128 B]: w1' row n, k = nc*128 + (which-1)*64 + chunk*8
                    const int n = q >> 3, c = (q & 7) ^ ((n >> 1) & 7);
                    off = (uint32_t)(n * 256 + nc * 128 + (which - 1) * 64 + c * 8) * 2u;
                    glds16(rsW1, dst + u * 4096, off);
                }
            }
        }
    };

    // ---- prologue: halo rows of t1 + first weight chunk ------------------------------------------------------
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int q = u * 256 + tid;
        const int row = q >> 3, c = (q & 7) ^ ((row >> 1) & 7);
        const int px = halo0 + row;
        const bool ok = q < halo_slots && px >= 0 && px < p.P;
        glds16(rsT1, R0 + u * 4096 + wave * 1024, ok ? (uint32_t)(px * 64 + c * 8) * 2u : OOB);
    }
    issue_chunk(0, 0);

    // this lane's pixel: tap validity and halo row
    const int m = wave * 32 + l31;                   // row of the tile
    const int px = p0 + m;
    uint32_t tapmask = 0;
    {
        const int rem = px % p.HW;
        const int r = rem / W, c = rem - r * W, H = p.HW / W;
        if (px < p.P) {
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int rr = r + tp / 3 - 1, cc = c + tp % 3 - 1;
                if (rr >= 0 && rr < H && cc >= 0 && cc < W) tapmask |= 1u << tp;
            }
        }
    }
    const int jc = m + W + 1;                        // halo row of the centre tap

    f32x16 acc1n[CN / 32];
#pragma unroll
    for (int i = 0; i < CN / 32; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc1n[i][e] = 0.f;

    // ======================================== stage A: 3 x 3 =================================================
    f32x16 acc2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[i][e] = 0.f;

#pragma unroll
    for (int id = 0; id < 5; ++id) {
        __syncthreads();                             // chunk id (and the halo) landed; chunk id-1 is consumed
        issue_chunk(id + 1, (id + 1) & 1);
        const char* wb = WB + (id & 1) * WB_BYTES;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int tp = 2 * id + h;
            if (tp < 9) {
                const int j = jc + (tp / 3 - 1) * W + (tp % 3 - 1);
                const bool ok = (tapmask >> tp) & 1u;
                const char* arow = R0 + j * 128;
                const int sw = (j >> 1) & 7;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    u32x4 x = lds128(arow + (((2 * ks + lh) ^ sw) << 4));
                    if (!ok) x = u32x4{0, 0, 0, 0};
                    const int c16 = h * 8 + 2 * ks + lh;
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const int n = nt * 32 + l31;
                        acc2[nt] = mma(lds128(wb + n * 256 + ((c16 ^ (n & 15)) << 4)), x, acc2[nt]);
                    }
                }
            }
        }
    }
    // t2 = relu(acc2 + b2) -> R1 row m (this wave's own rows)
    {
        char* row = R1 + m * 128;
        const int sw = (m >> 1) & 7;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n0 = nt * 32 + 8 * g + 4 * lh;
                const f32x4 b = *reinterpret_cast<const f32x4*>(p.b2 + n0);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc2[nt][4 * g + e] + b[e], 0.f);
                st_lds64(row + (((nt * 4 + g) ^ sw) << 4) + 8 * lh, u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])});
            }
    }

    // ============================ stages B / C per 128-channel slice of y =====================================
#pragma unroll
    for (int nc = 0; nc < 2; ++nc) {
        const int idD = 5 + nc * CPS, idB = idD + NDS;
        if constexpr (DS) {
            // ---- D: identity slice = xin . wd[nc*128 ..]^T + bd, rounded to bf16 like the separate launch would ----
            __syncthreads();                         // chunk D landed; everyone is done with R0 (halo / previous y slice)
            issue_chunk(idD + 1, (idD + 1) & 1);
            char* xreg = R0 + wave * 8192;           // this wave's 32 x-rows (128 B each) at the head of its own y rows
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = u * 64 + lane;
                const int ml = q >> 3, c = (q & 7) ^ ((ml >> 1) & 7);
                const int pr = p0 + wave * 32 + ml;
                glds16(rsX, xreg + u * 1024, pr < p.P ? (uint32_t)(pr * 64 + c * 8) * 2u : OOB);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            f32x16 accd[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) accd[i][e] = 0.f;
            {
                const char* wb = WB + (idD & 1) * WB_BYTES;
                const char* arow = xreg + l31 * 128;
                const int sw = (l31 >> 1) & 7;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const u32x4 x = lds128(arow + (((2 * ks + lh) ^ sw) << 4));
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const int n = nt * 32 + l31;
                        accd[nt] = mma(lds128(wb + n * 128 + (((2 * ks + lh) ^ ((n >> 1) & 7)) << 4)), x, accd[nt]);
                    }
                }
            }
            {
                char* row = R0 + m * 256;
                const int sw = m & 15;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n0 = nt * 32 + 8 * g + 4 * lh;
                        const f32x4 b = *reinterpret_cast<const f32x4*>(p.bd + nc * 128 + n0);
                        st_lds64(row + (((nt * 4 + g) ^ sw) << 4) + 8 * lh,
                                 u32x2{pack_bf16x2(accd[nt][4 * g + 0] + b[0], accd[nt][4 * g + 1] + b[1]),
                                       pack_bf16x2(accd[nt][4 * g + 2] + b[2], accd[nt][4 * g + 3] + b[3])});
                    }
            }
        }
        // ---- B: y slice = t2 . w3[nc*128 ..]^T -----------------------------------------------------------
        __syncthreads();                             // chunk B landed; (plain form) everyone is done with R0
        issue_chunk(idB + 1, (idB + 1) & 1);
        if constexpr (!DS) {   // identity slice -> R0, this wave's 32 rows only (so its arrival needs no barrier)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = u * 64 + lane;
                const int row = wave * 32 + (q >> 4), c = (q & 15) ^ (row & 15);
                const int pr = p0 + row;
                glds16(rsRes, R0 + (wave * 32 + u * 4) * 256, pr < p.P ? (uint32_t)(pr * 256 + nc * 128 + c * 8) * 2u : OOB);
            }
        }
        f32x16 accy[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) accy[i][e] = 0.f;
        {
            const char* wb = WB + (idB & 1) * WB_BYTES;
            const char* arow = R1 + m * 128;
            const int sw = (m >> 1) & 7;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const u32x4 x = lds128(arow + (((2 * ks + lh) ^ sw) << 4));
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int n = nt * 32 + l31;
                    accy[nt] = mma(lds128(wb + n * 128 + (((2 * ks + lh) ^ ((n >> 1) & 7)) << 4)), x, accy[nt]);
                }
            }
        }
        // epilogue: y = relu(acc + b3 + identity), in place over the identity slice (lane-own bytes)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's identity rows are in LDS
        {
            char* row = R0 + m * 256;
            const int sw = m & 15;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n0 = nt * 32 + 8 * g + 4 * lh;
                    const f32x4 b = *reinterpret_cast<const f32x4*>(p.b3 + nc * 128 + n0);
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
        // y slice -> global, 16 bytes per lane, 256-byte runs per pixel (this wave's rows)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int q = u * 64 + lane;
            const int row = wave * 32 + (q >> 4), c = (q & 15) ^ (row & 15);
            const int pr = p0 + row;
            const u32x4 v = lds128(R0 + (wave * 32 + u * 4) * 256 + lane * 16);
            if constexpr (BRANCHY) { if (pr < p.P) *reinterpret_cast<u32x4*>(p.y + (size_t)pr * 256 + nc * 128 + c * 8) = v; }
            else __builtin_amdgcn_raw_buffer_store_b128(v, rsY, (uint32_t)(pr * 256 + nc * 128 + c * 8) * 2u, 0, 0);
        }
        // ---- C: t1' += y slice . w1'[:, nc*128 ..]^T -------------------------------------------------------
#pragma unroll
        for (int hc = 0; hc < CPS - 1 - NDS; ++hc) {
            const int idC = idB + 1 + hc;
            __syncthreads();                         // chunk C landed; chunk B is consumed
            if (idC + 1 < NCHUNK) issue_chunk(idC + 1, (idC + 1) & 1);
            const char* wb = WB + (idC & 1) * WB_BYTES;
            const char* arow = R0 + m * 256;
            const int sw = m & 15;
            if constexpr (CN == 64) {
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const u32x4 x = lds128(arow + (((2 * ks + lh) ^ sw) << 4));
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const int n = nt * 32 + l31;
                        acc1n[nt] = mma(lds128(wb + n * 256 + (((2 * ks + lh) ^ (n & 15)) << 4)), x, acc1n[nt]);
                    }
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const u32x4 x = lds128(arow + (((hc * 8 + 2 * ks + lh) ^ sw) << 4));
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const int n = nt * 32 + l31;
                        acc1n[nt] = mma(lds128(wb + n * 128 + (((2 * ks + lh) ^ ((n >> 1) & 7)) << 4)), x, acc1n[nt]);
                    }
                }
            }
        }
    }

    // ================================ t1' = relu(acc + b1') -> global ==========================================
    if constexpr (CN == 64) {
        char* row = R1 + m * 128;                    // t2 is dead; this wave's own rows again
        const int sw = (m >> 1) & 7;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n0 = nt * 32 + 8 * g + 4 * lh;
                const f32x4 b = *reinterpret_cast<const f32x4*>(p.b1n + n0);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc1n[nt][4 * g + e] + b[e], 0.f);
                st_lds64(row + (((nt * 4 + g) ^ sw) << 4) + 8 * lh, u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])});
            }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = u * 64 + lane;
            const int r = wave * 32 + (q >> 3), c = (q & 7) ^ ((r >> 1) & 7);
            const int pr = p0 + r;
            const u32x4 v = lds128(R1 + (wave * 32 + u * 8) * 128 + lane * 16);
            if constexpr (BRANCHY) { if (pr < p.P) *reinterpret_cast<u32x4*>(p.t1n + (size_t)pr * 64 + c * 8) = v; }
            else __builtin_amdgcn_raw_buffer_store_b128(v, rsT1n, (uint32_t)(pr * 64 + c * 8) * 2u, 0, 0);
        }
    } else {
        char* row = R0 + m * 256;                    // the y slice is dead (this wave's rows)
        const int sw = m & 15;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n0 = nt * 32 + 8 * g + 4 * lh;
                const f32x4 b = *reinterpret_cast<const f32x4*>(p.b1n + n0);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc1n[nt][4 * g + e] + b[e], 0.f);
                st_lds64(row + (((nt * 4 + g) ^ sw) << 4) + 8 * lh, u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])});
            }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int q = u * 64 + lane;
            const int r = wave * 32 + (q >> 4), c = (q & 15) ^ (r & 15);
            const int pr = p0 + r;
            const u32x4 v = lds128(R0 + (wave * 32 + u * 4) * 256 + lane * 16);
            __builtin_amdgcn_raw_buffer_store_b128(v, rsT1n, (uint32_t)(pr * 128 + c * 8) * 2u, 0, 0);
        }
    }
}

}  // namespace

// t1 [P, 64], res / y [P, 256], t1n [P, cn] (cn = 64 or 128), all bf16; P = n * H * W pixels of W-wide square maps.
// Downsample form: res == nullptr, the identity is xin [P, 64] . wd^T + bd.
// w*_bytes: bytes from each weight pointer to the end of its allocation (descriptor extent; the tail chunk of the
// 3x3 weights over-reads 128 bytes past row 63, which must stay inside the packed weight buffer).
int sq_launch_bottleneck_tail_c64(const bf16_t* t1, const bf16_t* res, bf16_t* y, bf16_t* t1n, int cn,
                                  const bf16_t* w2, const bf16_t* w3, const bf16_t* w1n, size_t w2_bytes, size_t w3_bytes, size_t w1n_bytes,
                                  const float* b2, const float* b3, const float* b1n,
                                  const bf16_t* xin, const bf16_t* wd, size_t wd_bytes, const float* bd,
                                  int n_img, int H, int W, hipStream_t stream) {
    SQ_REQUIRE(cn == 64 || cn == 128, "bottleneck tail: next width %d (64 or 128)", cn);
    SQ_REQUIRE(H == W && W >= 3 && (128 + 2 * W + 2) * 128 <= R0_BYTES, "bottleneck tail: map %d x %d does not fit the halo buffer", H, W);
    const long long P = (long long)n_img * H * W;
    SQ_REQUIRE(P > 0 && P * 512 < (1ll << 31), "bottleneck tail: %lld pixels exceed the 2 GiB descriptor limit", P);
    SQ_REQUIRE(w2_bytes >= 64 * 576 * 2 + 128 && w3_bytes >= 256 * 64 * 2 && w1n_bytes >= (size_t)cn * 256 * 2, "bottleneck tail: weight extents");
    const bool ds = res == nullptr;
    SQ_REQUIRE(!ds || (xin && wd && bd && wd_bytes >= 256 * 64 * 2), "bottleneck tail: neither an identity tensor nor a downsample branch");
    BtlArgs a;
    a.t1 = t1; a.res = res; a.y = y; a.t1n = t1n; a.w2 = w2; a.w3 = w3; a.w1n = w1n; a.b2 = b2; a.b3 = b3; a.b1n = b1n;
    a.xin = xin; a.wd = wd; a.bd = bd;
    a.P = (int)P; a.W = W; a.HW = H * W; a.tiles = (int)((P + 127) / 128);
    auto clamp = [](size_t b) { return (uint32_t)(b < 0x7fffffffu ? b : 0x7fffffffu); };
    a.w2_bytes = clamp(w2_bytes); a.w3_bytes = clamp(w3_bytes); a.w1n_bytes = clamp(w1n_bytes); a.wd_bytes = clamp(wd_bytes);
    static SqDevOnce attr;       // hipFuncSetAttribute is per device
    if (attr.needed()) {
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)btl_tail_kernel<64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)btl_tail_kernel<128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)btl_tail_kernel<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)btl_tail_kernel<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr.done();
    }
    int prof = -1;
    if (sq_prof_on()) {
        char name[96];
        snprintf(name, sizeof(name), "btl_tail_c64_cn%d%s_P%lld", cn, ds ? "_ds" : "", P);
        const double flops = 2.0 * P * (576.0 * 64 + 64.0 * 256 + 256.0 * cn + (ds ? 64.0 * 256 : 0.0));
        const double bytes = (double)P * 2.0 * (64 + (ds ? 64 : 256) + 256 + cn) + 2.0 * (64 * 576 + 256 * 64 * (ds ? 2 : 1) + cn * 256);
        prof = sq_prof_begin(name, flops, bytes, stream);
    }
    const dim3 grid(a.tiles), block(256);
    if (cn == 64 && !ds) hipLaunchKernelGGL((btl_tail_kernel<64, false>), grid, block, LDS_BYTES, stream, a);
    else if (cn == 128 && !ds) hipLaunchKernelGGL((btl_tail_kernel<128, false>), grid, block, LDS_BYTES, stream, a);
    else if (cn == 64) hipLaunchKernelGGL((btl_tail_kernel<64, true>), grid, block, LDS_BYTES, stream, a);
    else hipLaunchKernelGGL((btl_tail_kernel<128, true>), grid, block, LDS_BYTES, stream, a);
    SQ_LAUNCH_CHECK();
    if (prof >= 0) sq_prof_end(prof, stream);
    return SQ_OK;
}
