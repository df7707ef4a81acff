// ResNet-50 stem in one kernel (bf16 mode):
//   uint8 HWC patch -> /255 -> ImageNet normalise -> conv1 7x7 stride 2 pad 3 (3 -> 64, BN folded) -> ReLU
//   -> MaxPool2d(3, stride 2, padding 1)  ==  src/resnet.py:157-160 after compute_features_hdf5.py:119-120.
//
// The im2col route writes (and the GEMM re-reads) a [n*112*112, 152] matrix: 3.8 GB per 500 patches against
// 75 MB of pixels in and 100 MB of pooled activations out.  Here a block owns a 17 x 17 tile of conv outputs
// (what an 8 x 8 tile of pooled outputs needs) and never leaves the chip in between:
//   * the 39 x 40 input window is normalised once into LDS as [row][col][4 ch] bf16 (channel 3 = 0), so the
//     two pixels x 4 channels an MFMA lane needs for 8 consecutive k are one aligned ds_read_b128;
//   * K is laid out ky-major, 8 kx (kx = 7 has zero weight) x 4 ch = 32 per ky: K = 224 instead of 147,
//     traded for aligned 16-byte fragment reads (the stem is not MFMA-bound);
//   * the weights live in registers for the whole (persistent) block: 14 k-steps x 4 VGPRs per wave;
//   * waves are 2 (M: 5 tiles of 32 conv pixels) x 2 (N: 32 channels); the conv tile goes to LDS as bf16 after
//     bias + ReLU and the 3 x 3 max is taken from there (rounding is monotonic, so pooling bf16 values equals
//     rounding after pooling).
#include "gemm.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

constexpr int TP = 8;                    // pooled tile edge
constexpr int TC = 2 * TP + 1;           // conv tile edge (17)
constexpr int RW = 40;                   // staged input row pitch in pixels (39 needed + kx = 7 slot)
constexpr int RH = 2 * TC + 5;           // staged input rows (39)
constexpr int IN_BYTES = RH * RW * 8;    // 12480
constexpr int MT = 10;                   // 32-row MFMA tiles covering the 289 conv pixels
constexpr int CROW = 144;                // bytes per conv pixel in LDS (64 ch bf16 + 16 pad: conflict-free b16 writes)
constexpr int COUT_BYTES = MT * 32 * CROW;
constexpr int LUT_BYTES = 3 * 256 * 2;
constexpr int W_LD = 152;                // packed conv1 weight row: k = (ky*7 + kx)*3 + c, zero padded (resnet.hip)

struct Conv1Args {
    const uint8_t* u8;          // [n, S, S, 3] or null
    const float* f32;           // [n, 3, S, S] normalised, or null
    const bf16_t* w;            // [64, 152]
    const float* bias;          // [64]
    bf16_t* out;                // [n, S/4, S/4, 64]
    int n, S, tiles_per_side, tiles;
};

__global__ __launch_bounds__(256, 2) void conv1_pool_kernel(const Conv1Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_in = smem;
    char* s_out = smem + IN_BYTES;
    bf16_t* s_lut = reinterpret_cast<bf16_t*>(smem + IN_BYTES + COUT_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, g = lane >> 5;

    for (int i = tid; i < 3 * 256; i += 256) {     // the reference's fp32 transform, then the bf16 operand rounding
        const int c = i >> 8, v = i & 255;
        const float mean = c == 0 ? 0.485f : (c == 1 ? 0.456f : 0.406f), sd = c == 0 ? 0.229f : (c == 1 ? 0.224f : 0.225f);
        s_lut[i] = f32_to_bf16(((float)v / 255.0f - mean) / sd);
    }

    // weight fragments: k-step t = (ky, h); lane (n = l31, g) holds k = ky*32 + h*16 + g*8 + e  ->  kx = 4h + 2g + (e>>2), c = e&3
    bf16x8 wf[14];
    {
        const bf16_t* wr = p.w + (size_t)(wn * 32 + l31) * W_LD;
#pragma unroll
        for (int t = 0; t < 14; ++t) {
            const int ky = t >> 1, h = t & 1;
            union { bf16x8 v; uint16_t u[8]; } f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int kx = 4 * h + 2 * g + (e >> 2), c = e & 3;
                f.u[e] = (kx < 7 && c < 3) ? wr[(ky * 7 + kx) * 3 + c] : (uint16_t)0;
            }
            wf[t] = f.v;
        }
    }
    const float bias = p.bias[wn * 32 + l31];
    __syncthreads();                               // look-up table ready

    // A-fragment base addresses: conv pixel m = wm*160 + i*32 + l31 -> (oy, ox) in the 17 x 17 tile
    int a_off[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        int m = wm * 160 + i * 32 + l31;
        if (m >= TC * TC) m = 0;                   // rows past the tile: computed, never read
        const int oy = m / TC, ox = m - oy * TC;
        a_off[i] = ((2 * oy) * RW + 2 * ox + 2 * g) * 8;
    }

    const int S = p.S, PH = S / 4;
    const int tps2 = p.tiles_per_side * p.tiles_per_side;
    constexpr int NPX = (RH * RW + 255) / 256;     // staged pixels per thread (7)

    // uint8 source: the NEXT tile's pixels are fetched into registers while this tile is on the MFMA
    // Raw load results only: one 16-bit (channels 0, 1) and one 8-bit (channel 2) load per pixel.  They are split
    // in commit(), i.e. after the loop's back edge -- any arithmetic on them here would make the wave wait for the
    // loads before its MFMA phase instead of under it.
    uint32_t pre01[NPX], pre2[NPX];
    uint32_t pre_ok = 0;
    auto prefetch = [&](int tile) {
        const int img = tile / tps2, tt = tile - img * tps2;
        const int ty = tt / p.tiles_per_side, tx = tt - ty * p.tiles_per_side;
        const int iy0 = 4 * TP * ty - 5, ix0 = 4 * TP * tx - 5;
        const uint8_t* base = p.u8 + (size_t)img * S * S * 3;
        pre_ok = 0;
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            const int idx = tid + j * 256;
            const int r = idx / RW, q = idx - r * RW;
            const int iy = iy0 + r, ix = ix0 + q;
            const bool ok = idx < RH * RW && (unsigned)iy < (unsigned)S && (unsigned)ix < (unsigned)S;
            const uint8_t* px = base + ((uint32_t)(ok ? iy : 0) * S + (ok ? ix : 0)) * 3;
            uint16_t v01;
            __builtin_memcpy(&v01, px, 2);
            pre01[j] = v01; pre2[j] = px[2];
            pre_ok |= ok ? (1u << j) : 0u;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            const int idx = tid + j * 256;
            if (idx < RH * RW) {
                const bool ok = (pre_ok >> j) & 1;
                const uint32_t c0 = pre01[j] & 0xffu, c1 = (pre01[j] >> 8) & 0xffu, c2 = pre2[j] & 0xffu;
                const uint32_t lo = ok ? ((uint32_t)s_lut[c0] | ((uint32_t)s_lut[256 + c1] << 16)) : 0u;
                const uint32_t hi = ok ? (uint32_t)s_lut[512 + c2] : 0u;
                *reinterpret_cast<u32x2*>(s_in + idx * 8) = u32x2{lo, hi};
            }
        }
    };
    if (p.u8 && (int)blockIdx.x < p.tiles) prefetch(blockIdx.x);

    for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
        const int img = tile / tps2;
        const int tt = tile - img * tps2;
        const int ty = tt / p.tiles_per_side, tx = tt - ty * p.tiles_per_side;

        // ---- stage the normalised input window
        if (p.u8) {
            commit();
        } else {
            const int iy0 = 4 * TP * ty - 5, ix0 = 4 * TP * tx - 5;     // input pixel of staged (0, 0)
            for (int idx = tid; idx < RH * RW; idx += 256) {
                const int r = idx / RW, q = idx - r * RW;
                const int iy = iy0 + r, ix = ix0 + q;
                uint32_t lo = 0, hi = 0;
                if ((unsigned)iy < (unsigned)S && (unsigned)ix < (unsigned)S) {
                    const float* px = p.f32 + ((size_t)img * 3 * S + iy) * S + ix;
                    lo = pack_bf16x2(px[0], px[(size_t)S * S]);
                    hi = (uint32_t)f32_to_bf16(px[2 * (size_t)S * S]);
                }
                *reinterpret_cast<u32x2*>(s_in + idx * 8) = u32x2{lo, hi};
            }
        }
        __syncthreads();
        if (p.u8 && tile + (int)gridDim.x < p.tiles) prefetch(tile + gridDim.x);

        // ---- 17 x 17 x 64 conv tile on the MFMA (accumulators start at the folded-BN bias)
        f32x16 acc[5];
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = bias;
#pragma unroll
        for (int t = 0; t < 14; ++t) {
            const int ky = t >> 1, h = t & 1;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const u32x4 a = *reinterpret_cast<const u32x4*>(s_in + a_off[i] + (ky * RW + 4 * h) * 8);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), wf[t], acc[i], 0, 0, 0);
            }
        }
        // bf16 (hardware RNE pack) -> ReLU -> conv tile in LDS.  This phase is VALU-bound (80 values per lane), so it
        // works on PAIRS: v_cvt_pk_bf16_f32, and ReLU as a packed signed-16-bit max with 0 (a negative bf16 is a
        // negative int16).  C/D layout: col = l31 (channel), row = (r&3) + 8*(r>>2) + 4*g.
        {
            char* dst = s_out + (wm * 160 + 4 * g) * CROW + (wn * 32 + l31) * 2;
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 v = {acc[i][r], acc[i][r + 1]};
                    const i16x2 relu = __builtin_elementwise_max(__builtin_bit_cast(i16x2, __builtin_convertvector(v, bf16x2)), i16x2{0, 0});
                    const int m0 = i * 32 + (r & 3) + 8 * (r >> 2);               // rows m0 and m0 + 1
                    *reinterpret_cast<int16_t*>(dst + m0 * CROW) = relu[0];
                    *reinterpret_cast<int16_t*>(dst + (m0 + 1) * CROW) = relu[1];
                }
        }
        __syncthreads();

        // ---- 3 x 3 stride-2 max over the conv tile: thread = (pooled pixel, 16 channels).  All values are >= 0, so
        // bf16 order == unsigned 16-bit order (packed integer max) and 0 stands for the pool's -inf padding; taps
        // outside the image (first conv row / column of the first tile row / column) are redirected to their valid
        // neighbour, which cannot change a maximum.
        {
            const int pp = tid >> 2, cg = tid & 3;
            const int py = pp >> 3, px = pp & 7;
            u16x8 b0 = {0, 0, 0, 0, 0, 0, 0, 0}, b1 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    int cy = 2 * py + dy, cx = 2 * px + dx;                        // conv pixel inside the tile
                    if (ty == 0 && cy == 0) cy = 1;
                    if (tx == 0 && cx == 0) cx = 1;
                    const char* src = s_out + (cy * TC + cx) * CROW + cg * 32;
                    b0 = __builtin_elementwise_max(b0, *reinterpret_cast<const u16x8*>(src));
                    b1 = __builtin_elementwise_max(b1, *reinterpret_cast<const u16x8*>(src + 16));
                }
            bf16_t* dst = p.out + (((size_t)img * PH + TP * ty + py) * PH + TP * tx + px) * 64 + cg * 16;
            *reinterpret_cast<u16x8*>(dst) = b0;
            *reinterpret_cast<u16x8*>(dst + 8) = b1;
        }
        // the next tile's staging only touches s_in (all MFMA reads are behind the barrier above); its conv
        // tile is written after the next barrier, by which time every thread has finished pooling this one
    }
}

}  // namespace

// out [n, S/4, S/4, 64] bf16 = maxpool(relu(conv1(normalise(patches)) + bias)); S a multiple of 32
int sq_launch_conv1_pool_bf16(const uint8_t* u8, const float* f32_nchw, const bf16_t* w152, const float* bias, bf16_t* out,
                              int n, int S, hipStream_t stream) {
    SQ_REQUIRE(S % (4 * TP) == 0 && n >= 1, "conv1_pool: patch size %d must be a multiple of %d", S, 4 * TP);
    Conv1Args a;
    a.u8 = u8; a.f32 = f32_nchw; a.w = w152; a.bias = bias; a.out = out; a.n = n; a.S = S;
    a.tiles_per_side = S / (4 * TP);
    const long long tiles = (long long)n * a.tiles_per_side * a.tiles_per_side;
    SQ_REQUIRE(tiles < (1ll << 31), "conv1_pool: too many tiles");
    a.tiles = (int)tiles;
    const size_t lds = IN_BYTES + COUT_BYTES + LUT_BYTES;
    static SqDevOnce attr;       // hipFuncSetAttribute is per device
    if (attr.needed()) {
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)conv1_pool_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr.done();
    }
    const int grid = (int)(tiles < 512 ? tiles : 512);      // persistent: 2 blocks per CU keep their weights in registers
    int prof = -1;
    if (sq_prof_on()) {
        const double px_out = (double)n * (S / 2) * (S / 2);
        prof = sq_prof_begin("conv1_pool_bf16", 2.0 * px_out * 64 * 147, (double)n * S * S * 3 + (double)n * (S / 4) * (S / 4) * 64 * 2, stream);
    }
    hipLaunchKernelGGL(conv1_pool_kernel, dim3(grid), dim3(256), lds, stream, a);
    SQ_LAUNCH_CHECK();
    if (prof >= 0) sq_prof_end(prof, stream);
    return SQ_OK;
}
