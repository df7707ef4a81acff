// ResNet-50 stem in one kernel, split ("x3") modes -- the parity-grade twin of conv1.hip:
//   uint8 HWC patch -> /255 -> ImageNet normalise -> conv1 7x7 stride 2 pad 3 (3 -> 64, BN folded) -> ReLU
//   -> MaxPool2d(3, stride 2, padding 1)  ==  src/resnet.py:157-160 after compute_features_hdf5.py:119-120,
// with every value as a hi + lo pair of 16-bit planes (x3_fmt.h) and three MFMAs per product.
//
// The im2col route of the split modes wrote a [n*112*112, 152] matrix in two planes, read it back in the GEMM, wrote the
// 112 x 112 x 64 conv output in two planes and read that again in the pooling kernel: ~11 GB per 500 patches.  Here a
// block owns a 17 x 17 tile of conv outputs (what an 8 x 8 tile of pooled outputs needs); 75 MB of pixels come in and
// 0.4 GB of pooled hi / lo planes go out per 500 patches.
//   * the 39 x 40 input window is normalised once (the reference's fp32 operations, through a 3 x 256 table), split and
//     kept in LDS as TWO [row][col][4 ch] planes, so an MFMA lane's 8 consecutive k are one aligned ds_read_b128 per plane;
//   * K is laid out ky-major, 8 kx (kx = 7 has zero weight) x 4 ch = 32 per ky: K = 224 instead of 147;
//   * both weight planes live in registers for the whole (persistent) block: 2 x 14 k-steps x 4 VGPRs per wave;
//   * 8 waves = 4 (M) x 2 (N: 32 channels); the ten 32-pixel M tiles are dealt 3 / 3 / 2 / 2 so that the two waves
//     sharing a SIMD (w and w + 4) carry five tiles between them;
//   * the conv tile goes to LDS as fp32 after colscale / bias / ReLU, the 3 x 3 maximum is taken on fp32 values and
//     split once (splitting is monotonic: the maximum of split values equals the split maximum, so the result equals the
//     unfused GEMM -> split -> join -> max -> split chain up to the fp32 summation order inside the MFMAs).
#include "gemm.h"
#include "x3_fmt.h"

namespace {

constexpr int TP = 8;                    // pooled tile edge
constexpr int TC = 2 * TP + 1;           // conv tile edge (17)
constexpr int RW = 40;                   // staged input row pitch in pixels (39 needed + kx = 7 slot)
constexpr int RH = 2 * TC + 5;           // staged input rows (39)
constexpr int IN_BYTES = RH * RW * 8;    // 12480 per plane
constexpr int MT = 10;                   // 32-row MFMA tiles covering the 289 conv pixels
constexpr int CROW = 272;                // bytes per conv pixel in LDS (64 ch fp32 + 16 pad)
constexpr int COUT_BYTES = MT * 32 * CROW;
constexpr int LUT_BYTES = 2 * 3 * 256 * 2;
constexpr int W_LD = 152;                // packed conv1 weight row: k = (ky*7 + kx)*3 + c, zero padded (resnet.hip)
constexpr int NT = 512;
constexpr int RW_BYTES = 16384;           // fragment-ordered weights of the fused reduce
constexpr int LDS_BYTES = 2 * IN_BYTES + COUT_BYTES + LUT_BYTES + RW_BYTES;

struct Conv1X3Args {
    const uint8_t* u8;          // [n, S, S, 3] or null
    const float* f32;           // [n, 3, S, S] normalised, or null
    const uint16_t* w;          // hi plane [64, 152]; lo plane w_plane elements behind
    long long w_plane;
    const float* bias;          // [64]
    const float* colscale;      // [64] or null
    uint16_t* out;              // hi plane [n, S/4, S/4, 64]; lo plane out_plane elements behind
    long long out_plane;
    int n, S, tiles_per_side, tiles;
    // optional: the first bottleneck's reduce 1x1 (64 -> 64, src/resnet.py:75-77) on the pooled tile before it leaves the CU:
    // t1 = relu(x . w1^T * cs1 + b1) as planes beside x -- the launch that would read x back (0.8 GB per 1000 patches) is gone
    const uint16_t* w1;         // hi plane [64, 64] (row-major, or K-tile-major when w1_tiled); lo plane w_plane elements behind
    int w1_tiled;
    const float* b1; const float* cs1;
    uint16_t* t1;               // hi plane [n, S/4, S/4, 64]; lo plane out_plane elements behind; null: not fused
};

template <bool F16>
__global__ __launch_bounds__(NT) void conv1_pool_x3_kernel(const Conv1X3Args p) {
    using F = X3Fmt<F16>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_hi = smem;
    char* s_lo = smem + IN_BYTES;
    char* s_out = smem + 2 * IN_BYTES;
    uint16_t* s_lut = reinterpret_cast<uint16_t*>(smem + 2 * IN_BYTES + COUT_BYTES);      // [plane][3][256]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, g = lane >> 5;
    const int tile0 = wm < 2 ? wm * 3 : 6 + (wm - 2) * 2;     // first M tile of this wave
    const int ntile = wm < 2 ? 3 : 2;

    for (int i = tid; i < 3 * 256; i += NT) {      // the reference's fp32 transform, then the hi / lo split
        const int c = i >> 8, v = i & 255;
        const float mean = c == 0 ? 0.485f : (c == 1 ? 0.456f : 0.406f), sd = c == 0 ? 0.229f : (c == 1 ? 0.224f : 0.225f);
        const float x = ((float)v / 255.0f - mean) / sd;
        const uint32_t h = F::pack2(x, 0.f);
        const uint32_t l = F::pack2(x - F::lo_f(h), 0.f);
        s_lut[i] = (uint16_t)h;
        s_lut[768 + i] = (uint16_t)l;
    }

    // weight fragments: k-step t = (ky, h); lane (n = l31, g) holds k = ky*32 + h*16 + g*8 + e  ->  kx = 4h + 2g + (e>>2), c = e&3
    u32x4 wh[14], wl[14];
    {
        const uint16_t* wr = p.w + (size_t)(wn * 32 + l31) * W_LD;
#pragma unroll
        for (int t = 0; t < 14; ++t) {
            const int ky = t >> 1, h = t & 1;
            union { u32x4 v; uint16_t u[8]; } fh, fl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int kx = 4 * h + 2 * g + (e >> 2), c = e & 3;
                const bool ok = kx < 7 && c < 3;
                const int k = ok ? (ky * 7 + kx) * 3 + c : 0;
                fh.u[e] = ok ? wr[k] : (uint16_t)0;
                fl.u[e] = ok ? wr[p.w_plane + k] : (uint16_t)0;
            }
            wh[t] = fh.v; wl[t] = fl.v;
        }
    }
    const float bias = p.bias[wn * 32 + l31];
    const float cscale = p.colscale ? p.colscale[wn * 32 + l31] : 1.0f;
    // fused reduce: waves 0-3 = (pixel half ri, channel half rj) of the 64 x 64 result; B fragments (both planes, 4 k-steps) in registers
    const bool fuse = p.t1 != nullptr;
    const int ri = (wave >> 1) & 1, rj = wave & 1;
    // B fragments of the reduce in fragment order in LDS, written once per (persistent) block: [channel half rj][k-step][plane][lane][16 B]
    // = 16 KiB (in registers -- 32 more beside the 112 of the conv weights -- the kernel spills)
    char* const s_rw = smem + 2 * IN_BYTES + COUT_BYTES + LUT_BYTES;
    if (fuse) {
        for (int sl = tid; sl < 1024; sl += NT) {
            const int ln = sl & 63, pl = (sl >> 6) & 1, ks = (sl >> 7) & 3, j = sl >> 9;
            const int n = j * 32 + (ln & 31), k = ks * 16 + (ln >> 5) * 8;
            const size_t o = p.w1_tiled ? ((size_t)(k >> 5) * 64 + n) * 32 + (k & 31) : (size_t)n * 64 + k;
            *reinterpret_cast<u32x4*>(s_rw + sl * 16) = *reinterpret_cast<const u32x4*>(p.w1 + (pl ? p.w_plane : 0) + o);
        }
    }
    __syncthreads();                               // look-up table ready

    // A-fragment base addresses: conv pixel m = tile*32 + l31 -> (oy, ox) in the 17 x 17 tile
    int a_off[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        int m = (tile0 + i) * 32 + l31;
        if (m >= TC * TC) m = 0;                   // rows past the tile: computed, never read
        const int oy = m / TC, ox = m - oy * TC;
        a_off[i] = ((2 * oy) * RW + 2 * ox + 2 * g) * 8;
    }

    const int S = p.S, PH = S / 4;
    const int tps2 = p.tiles_per_side * p.tiles_per_side;
    constexpr int NPX = (RH * RW + NT - 1) / NT;   // staged pixels per thread (4)

    // uint8 source: the NEXT tile's pixels are fetched into registers while this tile is on the MFMA.  Raw load results
    // only (one 16-bit and one 8-bit load per pixel); they are looked up in commit(), after the loop's back edge.
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
            const int idx = tid + j * NT;
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
            const int idx = tid + j * NT;
            if (idx < RH * RW) {
                const bool ok = (pre_ok >> j) & 1;
                const uint32_t c0 = pre01[j] & 0xffu, c1 = (pre01[j] >> 8) & 0xffu, c2 = pre2[j] & 0xffu;
                const uint32_t h01 = ok ? ((uint32_t)s_lut[c0] | ((uint32_t)s_lut[256 + c1] << 16)) : 0u;
                const uint32_t h2 = ok ? (uint32_t)s_lut[512 + c2] : 0u;
                const uint32_t l01 = ok ? ((uint32_t)s_lut[768 + c0] | ((uint32_t)s_lut[768 + 256 + c1] << 16)) : 0u;
                const uint32_t l2 = ok ? (uint32_t)s_lut[768 + 512 + c2] : 0u;
                *reinterpret_cast<u32x2*>(s_hi + idx * 8) = u32x2{h01, h2};
                *reinterpret_cast<u32x2*>(s_lo + idx * 8) = u32x2{l01, l2};
            }
        }
    };
    if (p.u8 && (int)blockIdx.x < p.tiles) prefetch(blockIdx.x);

    for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
        const int img = tile / tps2;
        const int tt = tile - img * tps2;
        const int ty = tt / p.tiles_per_side, tx = tt - ty * p.tiles_per_side;

        // ---- stage the normalised, split input window
        if (p.u8) {
            commit();
        } else {
            const int iy0 = 4 * TP * ty - 5, ix0 = 4 * TP * tx - 5;     // input pixel of staged (0, 0)
            for (int idx = tid; idx < RH * RW; idx += NT) {
                const int r = idx / RW, q = idx - r * RW;
                const int iy = iy0 + r, ix = ix0 + q;
                u32x2 hv = {0, 0}, lv = {0, 0};
                if ((unsigned)iy < (unsigned)S && (unsigned)ix < (unsigned)S) {
                    const float* px = p.f32 + ((size_t)img * 3 * S + iy) * S + ix;
                    const float x0 = px[0], x1 = px[(size_t)S * S], x2 = px[2 * (size_t)S * S];
                    hv[0] = F::pack2(x0, x1); hv[1] = F::pack2(x2, 0.f);
                    lv[0] = F::rest2(x0, x1, hv[0]); lv[1] = F::pack2(x2 - F::lo_f(hv[1]), 0.f);
                }
                *reinterpret_cast<u32x2*>(s_hi + idx * 8) = hv;
                *reinterpret_cast<u32x2*>(s_lo + idx * 8) = lv;
            }
        }
        __syncthreads();
        if (p.u8 && tile + (int)gridDim.x < p.tiles) prefetch(tile + gridDim.x);

        // ---- 17 x 17 x 64 conv tile on the MFMA: the two correction terms first, the leading term last
        f32x16 acc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int t = 0; t < 14; ++t) {
            const int ky = t >> 1, h = t & 1;
            u32x4 ah[3], al[3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (i < ntile) {
                    ah[i] = *reinterpret_cast<const u32x4*>(s_hi + a_off[i] + (ky * RW + 4 * h) * 8);
                    al[i] = *reinterpret_cast<const u32x4*>(s_lo + a_off[i] + (ky * RW + 4 * h) * 8);
                }
#pragma unroll
            for (int i = 0; i < 3; ++i) if (i < ntile) F::mma(al[i], wh[t], acc[i]);
#pragma unroll
            for (int i = 0; i < 3; ++i) if (i < ntile) F::mma(ah[i], wl[t], acc[i]);
#pragma unroll
            for (int i = 0; i < 3; ++i) if (i < ntile) F::mma(ah[i], wh[t], acc[i]);
        }
        // colscale / bias / ReLU -> fp32 conv tile in LDS.  C/D layout: col = l31 (channel), row = (r&3) + 8*(r>>2) + 4*g.
        {
            char* dst = s_out + (tile0 * 32 + 4 * g) * CROW + (wn * 32 + l31) * 4;
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (i < ntile) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m0 = i * 32 + (r & 3) + 8 * (r >> 2);
                        *reinterpret_cast<float*>(dst + m0 * CROW) = fmaxf(cscale * acc[i][r] + bias, 0.f);
                    }
                }
        }
        __syncthreads();

        // ---- 3 x 3 stride-2 max over the conv tile: thread = (pooled pixel, 8 channels).  All values are >= 0, so 0 stands
        // for the pool's -inf padding; taps outside the image (first conv row / column of the first tile row / column) are
        // redirected to their valid neighbour, which cannot change a maximum.
        {
            const int pp = tid >> 3, cg = tid & 7;
            const int py = pp >> 3, px = pp & 7;
            float best[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) best[e] = 0.f;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    int cy = 2 * py + dy, cx = 2 * px + dx;                        // conv pixel inside the tile
                    if (ty == 0 && cy == 0) cy = 1;
                    if (tx == 0 && cx == 0) cx = 1;
                    const char* src = s_out + (cy * TC + cx) * CROW + cg * 32;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { best[e] = fmaxf(best[e], v0[e]); best[4 + e] = fmaxf(best[4 + e], v1[e]); }
                }
            u32x4 hi, lo;
            x3_split8<F16>(best, hi, lo);
            uint16_t* dst = p.out + (((size_t)img * PH + TP * ty + py) * PH + TP * tx + px) * 64 + cg * 8;
            *reinterpret_cast<u32x4*>(dst) = hi;
            *reinterpret_cast<u32x4*>(dst + p.out_plane) = lo;
            if (fuse) {     // the pooled tile as the A image of the reduce: [64 px][128 B] hi | lo over the (dead) input planes, chunk ^= (px >> 1) & 7
                char* a = smem + pp * 128 + ((cg ^ ((pp >> 1) & 7)) << 4);
                *reinterpret_cast<u32x4*>(a) = hi;
                *reinterpret_cast<u32x4*>(a + 8192) = lo;
            }
        }
        if (fuse) {
            // ---- t1 = relu(x . w1^T * cs1 + b1): same K order, MFMA order and epilogue arithmetic as the gemm_x3.hip launch it replaces
            __syncthreads();                       // the A image is complete; every thread has finished pooling (s_out is free)
            if (wave < 4) {
                f32x16 racc;
#pragma unroll
                for (int r = 0; r < 16; ++r) racc[r] = 0.f;
                const int row = ri * 32 + l31;
                const char* arow = smem + row * 128;
                const int sw = (row >> 1) & 7;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const u32x4 ah = *reinterpret_cast<const u32x4*>(arow + (((2 * ks + g) ^ sw) << 4));
                    const u32x4 al = *reinterpret_cast<const u32x4*>(arow + 8192 + (((2 * ks + g) ^ sw) << 4));
                    const u32x4 bh = *reinterpret_cast<const u32x4*>(s_rw + (((rj * 4 + ks) * 2) * 64 + lane) * 16);
                    const u32x4 bl = *reinterpret_cast<const u32x4*>(s_rw + (((rj * 4 + ks) * 2 + 1) * 64 + lane) * 16);
                    F::mma(al, bh, racc);
                    F::mma(ah, bl, racc);
                    F::mma(ah, bh, racc);
                }
                char* dst = s_out + (ri * 32 + 4 * g) * CROW + (rj * 32 + l31) * 4;
#pragma unroll
                for (int r = 0; r < 16; ++r) *reinterpret_cast<float*>(dst + ((r & 3) + 8 * (r >> 2)) * CROW) = racc[r];
            }
            __syncthreads();
            {
                const int pp = tid >> 3, cg = tid & 7;
                const int py = pp >> 3, px = pp & 7;
                const char* src = s_out + pp * CROW + cg * 32;
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(src), a1 = *reinterpret_cast<const f32x4*>(src + 16);
                float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.b1 + cg * 8), b1v = *reinterpret_cast<const f32x4*>(p.b1 + cg * 8 + 4);
                float sc[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
                if (p.cs1) {
                    const f32x4 s0 = *reinterpret_cast<const f32x4*>(p.cs1 + cg * 8), s1 = *reinterpret_cast<const f32x4*>(p.cs1 + cg * 8 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { sc[e] = s0[e]; sc[4 + e] = s1[e]; }
                }
                const float bb[8] = {b0[0], b0[1], b0[2], b0[3], b1v[0], b1v[1], b1v[2], b1v[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = x3_relu(sc[e] * v[e] + bb[e]);
                u32x4 hi, lo;
                x3_split8<F16>(v, hi, lo);
                uint16_t* dst = p.t1 + (((size_t)img * PH + TP * ty + py) * PH + TP * tx + px) * 64 + cg * 8;
                *reinterpret_cast<u32x4*>(dst) = hi;
                *reinterpret_cast<u32x4*>(dst + p.out_plane) = lo;
            }
            // (the next tile's staging overwrites the A image: its readers are behind the barrier above; the next conv tile is written
            // behind the next staging barrier, by which time every thread has read its t1 chunk of s_out)
        }
        // the next tile's staging only touches the input planes (all MFMA reads are behind the barrier above); its conv
        // tile is written after the next barrier, by which time every thread has finished pooling this one
    }
}

}  // namespace

// out planes [n, S/4, S/4, 64] = split(maxpool(relu(colscale * conv1(normalise(patches)) + bias))); S a multiple of 32
int sq_launch_conv1_pool_x3(int f16, const uint8_t* u8, const float* f32_nchw, const uint16_t* w152_hi, long long w_plane, const float* bias,
                            const float* colscale, uint16_t* out_hi, long long out_plane, int n, int S, hipStream_t stream,
                            const uint16_t* w1_hi, int w1_tiled, const float* b1, const float* cs1, uint16_t* t1_hi) {
    SQ_REQUIRE(S % (4 * TP) == 0 && n >= 1, "conv1_pool_x3: patch size %d must be a multiple of %d", S, 4 * TP);
    Conv1X3Args a;
    a.u8 = u8; a.f32 = f32_nchw; a.w = w152_hi; a.w_plane = w_plane; a.bias = bias; a.colscale = colscale;
    a.out = out_hi; a.out_plane = out_plane; a.n = n; a.S = S;
    SQ_REQUIRE(!t1_hi || (w1_hi && b1), "conv1_pool_x3: the fused reduce needs its weights and bias");
    a.w1 = w1_hi; a.w1_tiled = w1_tiled; a.b1 = b1; a.cs1 = cs1; a.t1 = t1_hi;
    a.tiles_per_side = S / (4 * TP);
    const long long tiles = (long long)n * a.tiles_per_side * a.tiles_per_side;
    SQ_REQUIRE(tiles < (1ll << 31), "conv1_pool_x3: too many tiles");
    a.tiles = (int)tiles;
    static SqDevOnce attr;       // hipFuncSetAttribute is per device
    if (attr.needed()) {
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)conv1_pool_x3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)conv1_pool_x3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr.done();
    }
    const int grid = (int)(tiles < 256 ? tiles : 256);      // persistent: one block per CU keeps its weight planes in registers
    int prof = -1;
    if (sq_prof_on()) {
        const double px_out = (double)n * (S / 2) * (S / 2);
        prof = sq_prof_begin(f16 ? (t1_hi ? "conv1_pool_reduce_f16x3" : "conv1_pool_f16x3") : (t1_hi ? "conv1_pool_reduce_bf16x3" : "conv1_pool_bf16x3"),
                             2.0 * px_out * 64 * 147 + (t1_hi ? 2.0 * n * (S / 4) * (S / 4) * 64.0 * 64.0 : 0.0),
                             (double)n * S * S * 3 + (double)n * (S / 4) * (S / 4) * 64 * 4 * (t1_hi ? 2 : 1), stream);
    }
    if (f16) hipLaunchKernelGGL(conv1_pool_x3_kernel<true>, dim3(grid), dim3(NT), LDS_BYTES, stream, a);
    else hipLaunchKernelGGL(conv1_pool_x3_kernel<false>, dim3(grid), dim3(NT), LDS_BYTES, stream, a);
    SQ_LAUNCH_CHECK();
    if (prof >= 0) sq_prof_end(prof, stream);
    return SQ_OK;
}
