// ResNet-50 patch embedding (forward_extract) on the MFMA GEMM engine.
//
// Stands behind src/resnet.py:155-170 (forward_extract), :73-93 (Bottleneck.forward, stride on the 3x3),
// :98-136 (topology [3,4,6,3]) in eval mode, and the patch transform of
// pre_processing/compute_features_hdf5.py:49-51,119-120 (uint8 HWC -> /255 -> ImageNet normalise).
//
// Layout: activations NHWC (channels contiguous = the GEMM K axis), weights [Cout][kh][kw][Cin] with
// eval-mode BatchNorm folded in (w' = w * g/sqrt(var+eps), b' = beta - mean * g/sqrt(var+eps); folded on
// the host in fp64).  Every convolution is one launch of the NT GEMM engine:
//   1x1 stride 1        plain GEMM on [n*H*W, Cin]
//   3x3 / 1x1 stride 2  implicit GEMM (the A-tile loader gathers the taps, zero padding via buffer OOB)
//   conv1 7x7 (Cin = 3) fp32: explicit im2col (K = 147 padded to 152) fused with the uint8 -> normalised cast,
//                       then the max-pool kernel;  bf16: one fused stem kernel (conv1.hip)
// with bias + residual + ReLU in the GEMM epilogue.  The final 7x7 average is a small memory-bound kernel.
#include "../../include/sequoia_hip.h"
#include "gemm.h"
#include "x3_fmt.h"

int sq_launch_bottleneck_tail_c64(const bf16_t* t1, const bf16_t* res, bf16_t* y, bf16_t* t1n, int cn,
                                  const bf16_t* w2, const bf16_t* w3, const bf16_t* w1n, size_t w2_bytes, size_t w3_bytes, size_t w1n_bytes,
                                  const float* b2, const float* b3, const float* b1n,
                                  const bf16_t* xin, const bf16_t* wd, size_t wd_bytes, const float* bd,
                                  int n_img, int H, int W, hipStream_t stream);
int sq_launch_bottleneck_chain_c256(const bf16_t* t2, const bf16_t* res, bf16_t* y, bf16_t* t1n, const bf16_t* w3,
                                    const bf16_t* w1n, size_t w3_bytes, size_t w1n_bytes, const float* b3, const float* b1n,
                                    long long P, hipStream_t stream);
int sq_launch_bottleneck_chain_c128(const bf16_t* t2, const bf16_t* res, bf16_t* y, bf16_t* t1n, int cn, const bf16_t* w3,
                                    const bf16_t* w1n, size_t w3_bytes, size_t w1n_bytes, const float* b3, const float* b1n,
                                    long long P, hipStream_t stream);
int sq_launch_conv1_pool_bf16(const uint8_t* u8, const float* f32_nchw, const bf16_t* w152, const float* bias, bf16_t* out,
                              int n, int S, hipStream_t stream);
int sq_launch_chain_x3_c64(int f16, const uint16_t* t2, long long plT2, const uint16_t* res, long long plRes, uint16_t* y, long long plY,
                           uint16_t* t1n, long long plT1n, int n2, const uint16_t* w3, const uint16_t* w1n, long long plW, size_t w3_bytes,
                           const float* b3, const float* cs3, const float* b1n, const float* cs1n,
                           const uint16_t* xin, long long plX, const uint16_t* wd, size_t wd_bytes, const float* bd, const float* csd,
                           const uint16_t* t1, long long plT1, const uint16_t* w2, size_t w2_bytes, const float* b2, const float* cs2, int W, int HW,
                           long long P, uint16_t* frag, int w_tiled, hipStream_t stream);
size_t sq_chain_x3_frag_bytes();
bool sq_chain_x3w_eligible(int c, int n2);
int sq_launch_chain_x3w(int f16, int c, const uint16_t* t2, long long plT2, const uint16_t* res, long long plRes, uint16_t* y, long long plY,
                        uint16_t* t1n, long long plT1n, int n2, const uint16_t* w3, const uint16_t* w1n, long long plW, size_t w3_bytes, size_t w1n_bytes,
                        const float* b3, const float* cs3, const float* b1n, const float* cs1n, long long P, int w_tiled, hipStream_t stream);
int sq_launch_conv1_pool_x3(int f16, const uint8_t* u8, const float* f32_nchw, const uint16_t* w152_hi, long long w_plane, const float* bias,
                            const float* colscale, uint16_t* out_hi, long long out_plane, int n, int S, hipStream_t stream,
                            const uint16_t* w1_hi, int w1_tiled, const float* b1, const float* cs1, uint16_t* t1_hi);

namespace {

constexpr int CONV1_K = 147, CONV1_KP = 152;
constexpr float BN_MEAN[3] = {0.485f, 0.456f, 0.406f};
constexpr float BN_STD[3] = {0.229f, 0.224f, 0.225f};

// A0[m, k] for conv1: m = (img, oh, ow), k = (kh*7 + kw)*3 + c; zero for padding taps and k >= 147.
// src_u8: uint8 NHWC (transform fused) or src_f32: fp32 NCHW (already normalised, reference tensor layout).
// The uint8 transform goes through a 3 x 256 look-up table built per block with exactly the reference's
// fp32 operations (ConvertImageDtype: u8 / 255; Normalize: (x - mean) / std), so results are bit-identical
// to evaluating the formula per element; tap geometry comes from a 152-entry table.
template <typename T>
__global__ __launch_bounds__(256) void im2col_conv1_kernel(const uint8_t* __restrict__ src_u8, const float* __restrict__ src_f32,
                                                           T* __restrict__ out, int n, int S, int OH) {
    __shared__ float lut[3][256];
    __shared__ int8_t t_kh[CONV1_KP], t_kw[CONV1_KP], t_c[CONV1_KP];
    for (int i = threadIdx.x; i < 3 * 256; i += 256) {
        const int c = i >> 8, v = i & 255;
        const float p = (float)v / 255.0f;
        lut[c][v] = (p - BN_MEAN[c]) / BN_STD[c];
    }
    for (int k = threadIdx.x; k < CONV1_KP; k += 256) {
        const int tap = k / 3;
        t_c[k] = k < CONV1_K ? (int8_t)(k - tap * 3) : (int8_t)-1;
        t_kh[k] = (int8_t)(tap / 7);
        t_kw[k] = (int8_t)(tap % 7);
    }
    __syncthreads();
    constexpr int CH = CONV1_KP / 8;     // 19 chunks of 8
    // 32-bit index arithmetic (64-bit div/mod costs hundreds of instructions per element; total < 2^31 checked by the caller)
    const uint32_t total = (uint32_t)n * OH * OH * CH;
    for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const uint32_t ch = idx % CH;
        const uint32_t m = idx / CH;
        const uint32_t t1 = m / OH;
        const int ow = (int)(m - t1 * OH), img = (int)(t1 / OH), oh = (int)(t1 - (uint32_t)img * OH);
        const int ih0 = oh * 2 - 3, iw0 = ow * 2 - 3;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = ch * 8 + e;
            const int c = t_c[k];
            const int ih = ih0 + t_kh[k], iw = iw0 + t_kw[k];
            float x = 0.f;
            if (c >= 0 && (unsigned)ih < (unsigned)S && (unsigned)iw < (unsigned)S) {
                if (src_u8) x = lut[c][src_u8[(((size_t)img * S + ih) * S + iw) * 3 + c]];
                else x = src_f32[(((size_t)img * 3 + c) * S + ih) * S + iw];
            }
            v[e] = x;
        }
        if constexpr (sizeof(T) == 2) {
            u32x4 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
            *reinterpret_cast<u32x4*>(out + (size_t)m * CONV1_KP + ch * 8) = o;
        } else {
            f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
            *reinterpret_cast<f32x4*>(out + (size_t)m * CONV1_KP + ch * 8) = a;
            *reinterpret_cast<f32x4*>(out + (size_t)m * CONV1_KP + ch * 8 + 4) = b;
        }
    }
}

__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const bf16_t* p) { return bf16_to_f32(*p); }
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(bf16_t* p, float v) { *p = f32_to_bf16(v); }

// MaxPool2d(3, stride 2, padding 1) on NHWC (resnet.py:105); thread = (pixel, 16-byte channel group)
template <typename T>
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const T* __restrict__ in, T* __restrict__ out, int n, int H, int OH, int C) {
    constexpr int V = 16 / (int)sizeof(T);     // channels per 16-byte access
    const int CG = C / V;
    const uint32_t total = (uint32_t)n * OH * OH * CG;
    for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const uint32_t cg = idx % CG;
        const uint32_t px = idx / CG;
        const uint32_t t1 = px / OH;
        const int ow = (int)(px - t1 * OH), img = (int)(t1 / OH), oh = (int)(t1 - (uint32_t)img * OH);
        float best[V];
#pragma unroll
        for (int e = 0; e < V; ++e) best[e] = -INFINITY;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ih = oh * 2 - 1 + kh, iw = ow * 2 - 1 + kw;
                if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)H) {
                    const u32x4 t = *reinterpret_cast<const u32x4*>(in + (((size_t)img * H + ih) * H + iw) * C + cg * V);
                    if constexpr (sizeof(T) == 2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            best[2 * e] = fmaxf(best[2 * e], __uint_as_float(t[e] << 16));
                            best[2 * e + 1] = fmaxf(best[2 * e + 1], __uint_as_float(t[e] & 0xffff0000u));
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) best[e] = fmaxf(best[e], __uint_as_float(t[e]));
                    }
                }
            }
        u32x4 o;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__float_as_uint(best[2 * e]) >> 16) | (__float_as_uint(best[2 * e + 1]) & 0xffff0000u);   // exact: values are bf16
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = __float_as_uint(best[e]);
        }
        *reinterpret_cast<u32x4*>(out + (size_t)px * C + cg * V) = o;
    }
}

// AvgPool2d(7) (resnet.py:110,166): mean of the top-left 7x7 window of the final map, fp32 out
template <typename T>
__global__ void avgpool7_kernel(const T* __restrict__ in, float* __restrict__ out, int n, int H, int C) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * C) return;
    const int c = idx % C, img = idx / C;
    float acc = 0.f;
    for (int h = 0; h < 7; ++h)
        for (int w = 0; w < 7; ++w) acc += ldf(in + (((size_t)img * H + h) * H + w) * C + c);
    out[idx] = acc / 49.0f;
}

// ---- split (SQ_BF16X3 / SQ_F16X3) variants: every tensor is a hi plane and a lo plane of bf16 / fp16 (x3_fmt.h, gemm_x3.hip) ----

// AvgPool2d(7) over hi / lo planes, fp32 out (same summation order as avgpool7_kernel)
// A non-finite pooled value (an fp16 plane overflowed somewhere upstream: x3_fmt.h x3_relu) ORs 1 into *nonfinite.
template <bool F16>
__global__ void avgpool7_x3_kernel(const bf16_t* __restrict__ in, long long plane, float* __restrict__ out, int n, int H, int C,
                                   unsigned* __restrict__ nonfinite) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * C) return;
    const int c = idx % C, img = idx / C;
    float acc = 0.f;
    for (int h = 0; h < 7; ++h)
        for (int w = 0; w < 7; ++w) {
            const size_t o = (((size_t)img * H + h) * H + w) * C + c;
            acc += X3Fmt<F16>::one(in[o]) + X3Fmt<F16>::one(in[plane + o]);
        }
    out[idx] = acc / 49.0f;
    if (nonfinite && !(fabsf(acc) <= 3.0e38f)) atomicOr(nonfinite, 1u);
}

struct ConvSpec { int cin, cout, k, stride, pad; };

void build_specs(ConvSpec* specs) {
    int i = 0;
    specs[i++] = {3, 64, 7, 2, 3};
    const int planes[4] = {64, 128, 256, 512}, blocks[4] = {3, 4, 6, 3};
    int inplanes = 64;
    for (int li = 0; li < 4; ++li)
        for (int b = 0; b < blocks[li]; ++b) {
            const int s = (b == 0 && li > 0) ? 2 : 1;
            specs[i++] = {inplanes, planes[li], 1, 1, 0};
            specs[i++] = {planes[li], planes[li], 3, s, 1};
            specs[i++] = {planes[li], planes[li] * 4, 1, 1, 0};
            if (b == 0) specs[i++] = {inplanes, planes[li] * 4, 1, s, 0};
            inplanes = planes[li] * 4;
        }
}

struct RnBufs { void* col; void* act[5]; void* frag; size_t bytes; };

void rn_bufs(int dtype, int n, int S, char* base, RnBufs* o) {
    size_t off = 0;
    auto take = [&](size_t bytes) { off = sq_align_up(off, 256); char* p = base ? base + off : nullptr; off += bytes; return (void*)p; };
    const size_t es = sq_dtype_size(dtype);
    const size_t OH = S / 2;
    o->col = dtype == SQ_F32 ? take((size_t)n * OH * OH * CONV1_KP * es) : nullptr;    // bf16 and the split modes: fused stem, no im2col matrix
    const size_t act = (size_t)n * OH * OH * 64 * es;          // largest activation: conv1 out == layer1 out
    for (int i = 0; i < 5; ++i) o->act[i] = take(act);
    o->frag = (dtype == SQ_BF16X3 || dtype == SQ_F16X3) ? take(sq_chain_x3_frag_bytes()) : nullptr;     // chain_x3.hip's fragment-ordered weights
    o->bytes = sq_align_up(off, 256);
}

}  // namespace

extern "C" int sq_resnet50_layout_init(sq_resnet50_layout* out) {
    SQ_REQUIRE(out != nullptr, "resnet50_layout: null");
    ConvSpec specs[SQ_RESNET50_CONVS];
    build_specs(specs);
    int64_t w = 0, b = 0;
    for (int i = 0; i < SQ_RESNET50_CONVS; ++i) {
        sq_conv_desc& d = out->conv[i];
        d.cin = specs[i].cin; d.cout = specs[i].cout; d.k = specs[i].k; d.stride = specs[i].stride; d.pad = specs[i].pad;
        d.k_padded = i == 0 ? CONV1_KP : specs[i].k * specs[i].k * specs[i].cin;
        d.w_off = w; d.b_off = b;
        w += (int64_t)d.cout * d.k_padded;
        b += d.cout;
        w = (w + 7) / 8 * 8;
        b = (b + 7) / 8 * 8;
    }
    out->w_total = w;
    out->b_total = b;
    return SQ_OK;
}

extern "C" size_t sq_resnet50_workspace_bytes(int dtype, int n_patches, int patch_size) {
    if (n_patches < 1 || patch_size < 32 || patch_size % 32) return 0;
    RnBufs b;
    rn_bufs(dtype, n_patches, patch_size, nullptr, &b);
    return b.bytes;
}

extern "C" int sq_resnet50_extract(int dtype, const void* weights, const float* bias, const uint8_t* patches_u8,
                                   const float* patches_f32_nchw, int n, int S, float* features, void* workspace,
                                   size_t workspace_bytes, sq_stream_t stream_) {
    return sq_resnet50_extract_checked(dtype, weights, bias, patches_u8, patches_f32_nchw, n, S, features, workspace, workspace_bytes,
                                       nullptr, stream_);
}

extern "C" int sq_resnet50_extract_checked(int dtype, const void* weights, const float* bias, const uint8_t* patches_u8,
                                           const float* patches_f32_nchw, int n, int S, float* features, void* workspace,
                                           size_t workspace_bytes, uint32_t* nonfinite_flag, sq_stream_t stream_) {
    hipStream_t st = (hipStream_t)stream_;
    SQ_REQUIRE(dtype == SQ_F32 || dtype == SQ_BF16 || dtype == SQ_BF16X3 || dtype == SQ_F16X3, "resnet50: dtype %d", dtype);
    SQ_REQUIRE(weights && bias && features && workspace, "resnet50: null pointer");
    SQ_REQUIRE((patches_u8 != nullptr) != (patches_f32_nchw != nullptr), "resnet50: give exactly one of patches_u8 / patches_f32_nchw");
    SQ_REQUIRE(n >= 1 && S >= 224 && S % 32 == 0, "resnet50: n=%d patch_size=%d (need a multiple of 32, >= 224)", n, S);
    sq_resnet50_layout lay;
    sq_resnet50_layout_init(&lay);
    RnBufs b;
    rn_bufs(dtype, n, S, (char*)workspace, &b);
    if (b.bytes > workspace_bytes) {
        sq_set_error("resnet50: workspace %zu < required %zu", workspace_bytes, b.bytes);
        return SQ_ERR_WORKSPACE;
    }
    const bool x3 = dtype == SQ_BF16X3 || dtype == SQ_F16X3;
    const bool f16 = dtype == SQ_F16X3;           // fp16 planes; `bias` then carries [b_total] biases followed by [b_total] per-channel scales
    const float* colscale = x3 ? bias + lay.b_total : nullptr;      // weights: hi plane [w_total] then lo plane [w_total]; activations: hi / lo planes
    const size_t es = x3 ? 2 : sq_dtype_size(dtype);   // bytes per element of ONE plane
    const bool lp = dtype == SQ_BF16;
    auto W = [&](const sq_conv_desc& d) { return (const void*)((const char*)weights + (size_t)d.w_off * es); };
    const size_t w_bytes_total = (size_t)lay.w_total * es;
    const size_t act_cap = (size_t)n * (S / 2) * (S / 2) * 64 * es;
    const long long act_plane = (long long)n * (S / 2) * (S / 2) * 64;            // x3: elements between an activation's hi and lo plane
    SQ_REQUIRE(act_cap < (1ull << 31) && (lp || x3 || (size_t)n * (S / 2) * (S / 2) * CONV1_KP * es < (1ull << 31)),
               "resnet50: sub-batch of %d patches exceeds the 2 GiB buffer-descriptor limit (bf16 and the split modes: <= 1300 patches of 224, fp32: <= 280)", n);

    // conv as a GEMM launch.  in: NHWC [n, H, H, cin];  out: [n, OH, OH, cout]
    // dual (split modes): the block's downsample branch dd(din) -- din = the block's input [n, dH, dH, dd.cin] -- rides in the
    // expand launch as a second product instead of being written and read back as `res` (gemm.h A2 / B2)
    auto conv = [&](const sq_conv_desc& d, const void* in, int H, void* out, int OH, const void* res, int act,
                    const sq_conv_desc* dd = nullptr, const void* din = nullptr, int dH = 0) -> int {
        GemmArgs g;
        g.M = n * OH * OH; g.N = d.cout; g.K = d.k_padded;
        g.A = in;
        if (d.k == 1 && d.stride == 1) {
            g.lda = d.cin; g.a_bytes = (size_t)g.M * d.cin * es;
        } else {
            g.conv = 1; g.H = H; g.W = H; g.Cin = d.cin; g.OH = OH; g.OW = OH; g.KW = d.k; g.stride = d.stride; g.pad = d.pad;
            g.a_bytes = (size_t)n * H * H * d.cin * es;
        }
        g.B = W(d); g.ldb = d.k_padded; g.b_bytes = w_bytes_total - (size_t)d.w_off * es;
        g.bias = bias + d.b_off;
        g.res = res; g.ldres = d.cout; g.res_dtype = dtype;
        g.act = act;
        g.C = out; g.ldc = d.cout; g.out_dtype = dtype;
        if (x3) {
            g.plA = act_plane; g.plC = act_plane; g.plRes = act_plane; g.plB = lay.w_total;
            g.x3_f16 = f16; g.colscale = colscale + d.b_off;
            g.b_tiled = 1;             // every convolution behind the stem: K-tile-major planes (resnet.py split_planes)
            if (dd) {
                g.A2 = din; g.plA2 = act_plane; g.lda2 = dd->cin; g.a2_bytes = (size_t)n * dH * dH * dd->cin * es;
                g.B2 = W(*dd); g.ldb2 = dd->k_padded; g.b2_bytes = w_bytes_total - (size_t)dd->w_off * es; g.K2 = dd->k_padded;
                g.bias2 = bias + dd->b_off; g.colscale2 = colscale + dd->b_off;
                g.dH = dH; g.dW = dH; g.dOH = OH; g.dOW = OH; g.dstride = dd->stride;
            }
            return sq_launch_gemm_x3(g, st);
        }
        return sq_launch_gemm(g, dtype, st);
    };
#define RUN(expr) do { if (int _e = (expr)) return _e; } while (0)

    const int OH1 = S / 2;
    int H = OH1 / 2;    // after the max-pool
    bool stem_t1 = false;
    if (lp) {           // conv1 + bn1 + relu + maxpool in one kernel (conv1.hip)
        const sq_conv_desc& d = lay.conv[0];
        RUN(sq_launch_conv1_pool_bf16(patches_u8, patches_f32_nchw, (const bf16_t*)W(d), bias + d.b_off, (bf16_t*)b.act[1], n, S, st));
    } else if (x3) {    // the same fusion on hi / lo planes (conv1_x3.hip)
        const sq_conv_desc& d = lay.conv[0];
        // the first bottleneck's reduce 1x1 (64 -> 64) rides in the stem launch (SQ_RESNET_NO_STEM_REDUCE=1: its own launch): its
        // output lands in act[2], which the block loop below then finds as the first block's t1
        const sq_conv_desc& r1 = lay.conv[1];
        stem_t1 = r1.k == 1 && r1.stride == 1 && r1.cin == 64 && r1.cout == 64 && !sq_env_flag("SQ_RESNET_NO_STEM_REDUCE");
        RUN(sq_launch_conv1_pool_x3(f16, patches_u8, patches_f32_nchw, (const uint16_t*)W(d), lay.w_total, bias + d.b_off, colscale + d.b_off,
                                    (uint16_t*)b.act[1], act_plane, n, S, st,
                                    stem_t1 ? (const uint16_t*)W(r1) : nullptr, 1, stem_t1 ? bias + r1.b_off : nullptr,
                                    stem_t1 ? colscale + r1.b_off : nullptr, stem_t1 ? (uint16_t*)b.act[2] : nullptr));
    } else {
        {   // conv1 + bn1 + relu
            const size_t work = (size_t)n * OH1 * OH1 * (CONV1_KP / 8);
            size_t nb = (work + 255) / 256; if (nb > 65535) nb = 65535;
            hipLaunchKernelGGL(im2col_conv1_kernel<float>, dim3((int)nb), dim3(256), 0, st, patches_u8, patches_f32_nchw, (float*)b.col, n, S, OH1);
            SQ_LAUNCH_CHECK();
            GemmArgs g;
            const sq_conv_desc& d = lay.conv[0];
            g.M = n * OH1 * OH1; g.N = 64; g.K = CONV1_KP;
            g.A = b.col; g.lda = CONV1_KP; g.a_bytes = (size_t)g.M * CONV1_KP * es;
            g.B = W(d); g.ldb = CONV1_KP; g.b_bytes = w_bytes_total;
            g.bias = bias + d.b_off; g.act = SQ_ACT_RELU;
            g.C = b.act[0]; g.ldc = 64; g.out_dtype = dtype;
            RUN(sq_launch_gemm(g, dtype, st));
        }
        {
            const size_t work = (size_t)n * H * H * 64 / (16 / es);
            size_t nb = (work + 255) / 256; if (nb > 65535) nb = 65535;
            hipLaunchKernelGGL(maxpool3x3s2_kernel<float>, dim3((int)nb), dim3(256), 0, st, (const float*)b.act[0], (float*)b.act[1], n, OH1, H, 64);
            SQ_LAUNCH_CHECK();
        }
    }
    // bottleneck stack: x lives in act[xi]; t1, t2, ds, y are taken from the other four buffers.
    // bf16, 56 x 56 stage (layer 1): the 3x3, the expand 1x1 (+ identity, ReLU) and the NEXT block's reduce 1x1 are one
    // launch (bottleneck.hip) -- that block's conv1 output then already sits in act[t1i] when its turn comes.
    const bool fuse56 = lp && !sq_env_flag("SQ_RESNET_NO_FUSE") && (128 + 2 * H + 2) * 128 <= 32768;
    const bool fuse_chain = lp && !sq_env_flag("SQ_RESNET_NO_FUSE") && !sq_env_flag("SQ_RESNET_NO_CHAIN");
    const bool fuse_chain256 = fuse_chain;
    int xi = 1, ci = 1, t1i = stem_t1 ? 2 : -1;
    const int blocks[4] = {3, 4, 6, 3};
    for (int li = 0; li < 4; ++li)
        for (int bk = 0; bk < blocks[li]; ++bk) {
            int free_[4], nf = 0;
            for (int i = 0; i < 5; ++i) if (i != xi && i != t1i) free_[nf++] = i;
            void* x = b.act[xi];
            const int t1_idx = t1i >= 0 ? t1i : free_[--nf];
            void* t1 = b.act[t1_idx];
            const sq_conv_desc& c1 = lay.conv[ci]; const sq_conv_desc& c2 = lay.conv[ci + 1]; const sq_conv_desc& c3 = lay.conv[ci + 2];
            const bool has_ds = bk == 0;
            const int OH = H / c2.stride;
            if (t1i < 0) RUN(conv(c1, x, H, t1, H, nullptr, SQ_ACT_RELU));
            const int cnext = ci + (has_ds ? 4 : 3);
            if (fuse56 && li == 0) {
                void* y = b.act[free_[1]]; void* t1n = b.act[free_[2]];
                const sq_conv_desc& n1 = lay.conv[cnext];
                const sq_conv_desc& dsd = lay.conv[ci + 3];           // only read when has_ds
                auto rest = [&](const sq_conv_desc& d) { return w_bytes_total - (size_t)d.w_off * es; };
                // first block: the downsample branch (64 -> 256, stride 1 here) is computed inside the launch from x
                RUN(sq_launch_bottleneck_tail_c64((const bf16_t*)t1, has_ds ? nullptr : (const bf16_t*)x, (bf16_t*)y, (bf16_t*)t1n, n1.cout,
                                                  (const bf16_t*)W(c2), (const bf16_t*)W(c3), (const bf16_t*)W(n1), rest(c2), rest(c3), rest(n1),
                                                  bias + c2.b_off, bias + c3.b_off, bias + n1.b_off,
                                                  has_ds ? (const bf16_t*)x : nullptr, has_ds ? (const bf16_t*)W(dsd) : nullptr,
                                                  has_ds ? rest(dsd) : 0, has_ds ? bias + dsd.b_off : nullptr, n, H, H, st));
                xi = free_[1];
                t1i = free_[2];
                ci = cnext;
                continue;
            }
            void* t2 = b.act[free_[0]]; void* ds = b.act[free_[1]]; void* y = b.act[free_[2]];
            // split modes, 64-plane stage (56 x 56): 3x3 + expand 1x1 + identity + ReLU and the NEXT block's reduce 1x1 in one
            // launch (chain_x3.hip, tail form; SQ_RESNET_NO_TAIL=1: the 3x3 as its own launch, SQ_RESNET_NO_CHAIN=1: nothing
            // fused): t2 never leaves the CU, y is written once and not read back by the reduce; t1' lands in the buffer of
            // the other narrow tensor.  First block: the downsample branch (64 -> 256, stride 1 here) is computed inside too.
            const bool x3_chain = x3 && !sq_env_flag("SQ_RESNET_NO_CHAIN") && c3.cin == 64 && c3.cout == 256 && c2.stride == 1 &&
                cnext < SQ_RESNET50_CONVS && lay.conv[cnext].k == 1 && lay.conv[cnext].stride == 1 && lay.conv[cnext].cin == 256 &&
                (lay.conv[cnext].cout == 64 || lay.conv[cnext].cout == 128) &&
                (!has_ds || (lay.conv[ci + 3].k == 1 && lay.conv[ci + 3].stride == 1 && lay.conv[ci + 3].cin == 64 && !sq_env_flag("SQ_RESNET_NO_CHAIN_DS")));
            const bool x3_tail = x3_chain && c2.k == 3 && c2.cin == 64 && c2.cout == 64 && H <= 70 && !sq_env_flag("SQ_RESNET_NO_TAIL");
            if (!x3_tail) RUN(conv(c2, t1, H, t2, OH, nullptr, SQ_ACT_RELU));
            if (x3_chain) {
                const sq_conv_desc& n1 = lay.conv[cnext];
                const sq_conv_desc& dsd = lay.conv[ci + 3];           // only read when has_ds
                auto rest = [&](const sq_conv_desc& d) { return w_bytes_total - (size_t)d.w_off * es; };
                // t1' goes where no input of this launch lives: conv1's buffer when the 3x3 ran separately, else the t2 buffer
                void* t1n = x3_tail ? t2 : t1;
                RUN(sq_launch_chain_x3_c64(f16, x3_tail ? nullptr : (const uint16_t*)t2, act_plane, has_ds ? nullptr : (const uint16_t*)x, act_plane,
                                           (uint16_t*)y, act_plane, (uint16_t*)t1n, act_plane, n1.cout, (const uint16_t*)W(c3), (const uint16_t*)W(n1),
                                           lay.w_total, rest(c3), bias + c3.b_off, colscale + c3.b_off, bias + n1.b_off, colscale + n1.b_off,
                                           has_ds ? (const uint16_t*)x : nullptr, act_plane, has_ds ? (const uint16_t*)W(dsd) : nullptr,
                                           has_ds ? rest(dsd) : 0, has_ds ? bias + dsd.b_off : nullptr, has_ds ? colscale + dsd.b_off : nullptr,
                                           x3_tail ? (const uint16_t*)t1 : nullptr, act_plane, x3_tail ? (const uint16_t*)W(c2) : nullptr,
                                           x3_tail ? rest(c2) : 0, x3_tail ? bias + c2.b_off : nullptr, x3_tail ? colscale + c2.b_off : nullptr, H, H * H,
                                           (long long)n * OH * OH, (uint16_t*)b.frag, 1, st));
                ci = cnext;
                xi = free_[2];
                t1i = x3_tail ? free_[0] : t1_idx;
                H = OH;
                continue;
            }
            // split modes, plain bottlenecks of the 128- and 256-plane stages (28 x 28, 14 x 14): expand 1x1 + identity + ReLU and the
            // NEXT block's reduce 1x1 in one launch (chain_x3w.hip; SQ_RESNET_NO_CHAINW=1: two launches): y is written once and not
            // read back; t1' lands in the buffer conv1's output occupied (dead once conv2 has read it)
            if (x3 && !has_ds && !sq_env_flag("SQ_RESNET_NO_CHAINW") && c3.k == 1 && c3.cout == 4 * c3.cin && cnext < SQ_RESNET50_CONVS &&
                lay.conv[cnext].k == 1 && lay.conv[cnext].stride == 1 && lay.conv[cnext].cin == c3.cout &&
                sq_chain_x3w_eligible(c3.cin, lay.conv[cnext].cout)) {
                const sq_conv_desc& n1 = lay.conv[cnext];
                auto rest = [&](const sq_conv_desc& d) { return w_bytes_total - (size_t)d.w_off * es; };
                RUN(sq_launch_chain_x3w(f16, c3.cin, (const uint16_t*)t2, act_plane, (const uint16_t*)x, act_plane, (uint16_t*)y, act_plane,
                                        (uint16_t*)t1, act_plane, n1.cout, (const uint16_t*)W(c3), (const uint16_t*)W(n1), lay.w_total, rest(c3), rest(n1),
                                        bias + c3.b_off, colscale + c3.b_off, bias + n1.b_off, colscale + n1.b_off, (long long)n * OH * OH, 1, st));
                ci = cnext;
                xi = free_[2];
                t1i = t1_idx;
                H = OH;
                continue;
            }
            const void* identity = x;
            // split modes: the downsample branch of layers 2-4 (1x1, stride 2) as a second product of the expand launch
            // (SQ_RESNET_NO_DUAL=1: its own launch, a [n, OH, OH, 4 planes] tensor written and read back)
            const bool x3_dual = x3 && has_ds && lay.conv[ci + 3].k == 1 && c3.k == 1 && c3.cout % 128 == 0 && !sq_env_flag("SQ_RESNET_NO_DUAL");
            if (has_ds && !x3_dual) {
                RUN(conv(lay.conv[ci + 3], x, H, ds, OH, nullptr, SQ_ACT_NONE));
                identity = ds;
            }
            // 128-plane stage (28 x 28): expand 1x1 + identity + ReLU and the NEXT block's reduce 1x1 in one launch
            // (chain.hip); its output lands in the buffer conv1's output occupied (dead once conv2 has read it)
            if (fuse_chain && c3.cin == 128 && c3.cout == 512 && cnext < SQ_RESNET50_CONVS && lay.conv[cnext].k == 1 &&
                lay.conv[cnext].cin == 512 && (lay.conv[cnext].cout == 128 || lay.conv[cnext].cout == 256)) {
                const sq_conv_desc& n1 = lay.conv[cnext];
                auto rest = [&](const sq_conv_desc& d) { return w_bytes_total - (size_t)d.w_off * es; };
                RUN(sq_launch_bottleneck_chain_c128((const bf16_t*)t2, (const bf16_t*)identity, (bf16_t*)y, (bf16_t*)t1, n1.cout,
                                                    (const bf16_t*)W(c3), (const bf16_t*)W(n1), rest(c3), rest(n1), bias + c3.b_off,
                                                    bias + n1.b_off, (long long)n * OH * OH, st));
                ci = cnext;
                xi = free_[2];
                t1i = t1_idx;
                H = OH;
                continue;
            }
            // 256-plane stage (14 x 14): the same chain with the t2 rows held in registers (chain256.hip)
            if (fuse_chain256 && c3.cin == 256 && c3.cout == 1024 && cnext < SQ_RESNET50_CONVS && lay.conv[cnext].k == 1 &&
                lay.conv[cnext].cin == 1024 && lay.conv[cnext].cout == 256) {
                const sq_conv_desc& n1 = lay.conv[cnext];
                auto rest = [&](const sq_conv_desc& d) { return w_bytes_total - (size_t)d.w_off * es; };
                RUN(sq_launch_bottleneck_chain_c256((const bf16_t*)t2, (const bf16_t*)identity, (bf16_t*)y, (bf16_t*)t1,
                                                    (const bf16_t*)W(c3), (const bf16_t*)W(n1), rest(c3), rest(n1), bias + c3.b_off,
                                                    bias + n1.b_off, (long long)n * OH * OH, st));
                ci = cnext;
                xi = free_[2];
                t1i = t1_idx;
                H = OH;
                continue;
            }
            if (x3_dual) RUN(conv(c3, t2, OH, y, OH, nullptr, SQ_ACT_RELU, &lay.conv[ci + 3], x, H));
            else RUN(conv(c3, t2, OH, y, OH, identity, SQ_ACT_RELU));       // relu(bn3(conv3) + identity)
            ci = cnext;
            xi = free_[2];
            t1i = -1;
            H = OH;
        }
    {
        const int total = n * 2048;
        if (f16) hipLaunchKernelGGL(avgpool7_x3_kernel<true>, dim3((total + 255) / 256), dim3(256), 0, st, (const bf16_t*)b.act[xi], act_plane, features, n, H, 2048, nonfinite_flag);
        else if (x3) hipLaunchKernelGGL(avgpool7_x3_kernel<false>, dim3((total + 255) / 256), dim3(256), 0, st, (const bf16_t*)b.act[xi], act_plane, features, n, H, 2048, nonfinite_flag);
        else if (lp) hipLaunchKernelGGL(avgpool7_kernel<bf16_t>, dim3((total + 255) / 256), dim3(256), 0, st, (const bf16_t*)b.act[xi], features, n, H, 2048);
        else hipLaunchKernelGGL(avgpool7_kernel<float>, dim3((total + 255) / 256), dim3(256), 0, st, (const float*)b.act[xi], features, n, H, 2048);
        SQ_LAUNCH_CHECK();
    }
#undef RUN
    (void)act_cap;
    return SQ_OK;
}
