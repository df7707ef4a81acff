// HE2RNA comparator (src/he2rna.py:42-106) -- the device side of forward_fixed_k after the per-tile MLP:
//   mask[b, n]   = max_c x[b, c, n] > 0                                              (he2rna.py:94-95)
//   s[b, g, n]   = scores[b, n, g] * mask[b, n]                                      (:96)
//   out_k[b, g]  = sum_{j<k} sorted_desc(s[b, g, :])[j] * mask[b, j] / sum_{j<k} mask[b, j]     (:97-98)
//   eval         = sum_k out_k / len(ks)                                             (:88-91)
// The 1x1 Conv1d layers of the MLP (:101-106) are sq_linear launches on the token-major tensor [B*N, C].
//
// A thread owns one (slide, gene) column of N <= 128 tile scores, kept in LDS as [n][thread] (conflict-free, reads
// coalesced over genes).  The descending sort is never materialised: the rank of every entry (entries that are larger,
// plus equal ones with a smaller tile index) says which position weight mask[b, rank] it meets and which of the top-k
// sums it belongs to.  O(N^2) compares per column -- 0.7 ms for 64 slides x 20 820 genes -- and the same ranks route
// the gradient in the backward kernel (d out / d s[n] = sum_{k > rank} scale * mask[b, rank] / cnt_k).
// Sums are accumulated in fp64 (the reference adds fp32 values in sorted order: equal within fp32 rounding).
#include "../../include/sequoia_hip.h"
#include "elementwise.h"

namespace {

constexpr int HE_MAX_N = 128, HE_MAX_KS = 16, HE_THREADS = 256;

struct HeKs {
    int k[HE_MAX_KS];
    int n;
    float scale;
};

__global__ __launch_bounds__(256) void he2rna_mask_kernel(const float* __restrict__ x, int rows, int C, float* __restrict__ mask) {
    // one wave per row: max over the C channels of a token-major row
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, x[(size_t)row * C + c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) mask[row] = m > 0.f ? 1.f : 0.f;
}

template <bool BWD>
__global__ __launch_bounds__(HE_THREADS) void he2rna_topk_kernel(const float* __restrict__ scores, int lds_, const float* __restrict__ mask,
                                                                 const HeKs ks, const float* __restrict__ gout, float* __restrict__ out,
                                                                 int ldo, int N, int G) {
    extern __shared__ float sval[];                  // [N][HE_THREADS]
    __shared__ float w[HE_MAX_N];                    // mask of the slide's tile positions
    __shared__ double cnt[HE_MAX_KS];                // sum_{j<k} mask[b, j]
    const int b = blockIdx.y, g = blockIdx.x * HE_THREADS + threadIdx.x;
    const bool live = g < G;
    for (int n = threadIdx.x; n < N; n += HE_THREADS) w[n] = mask[(size_t)b * N + n];
    __syncthreads();
    if (threadIdx.x < ks.n) {
        double c = 0.0;
        for (int j = 0; j < ks.k[threadIdx.x]; ++j) c += (double)w[j];
        cnt[threadIdx.x] = c;
    }
    for (int n = 0; n < N; ++n)
        sval[n * HE_THREADS + threadIdx.x] = live ? scores[((size_t)b * N + n) * lds_ + g] * w[n] : 0.f;
    __syncthreads();
    double acc[HE_MAX_KS];
#pragma unroll
    for (int i = 0; i < HE_MAX_KS; ++i) acc[i] = 0.0;
    const float go = BWD && live ? gout[(size_t)b * G + g] : 0.f;
    for (int n = 0; n < N; ++n) {
        const float v = sval[n * HE_THREADS + threadIdx.x];
        int r = 0;
        for (int m = 0; m < N; ++m) {
            const float u = sval[m * HE_THREADS + threadIdx.x];
            r += (u > v) || (u == v && m < n);
        }
        const double wr = (double)w[r];
        if constexpr (!BWD) {
#pragma unroll
            for (int i = 0; i < HE_MAX_KS; ++i)
                if (i < ks.n && r < ks.k[i]) acc[i] += (double)v * wr;
        } else {
            double d = 0.0;
#pragma unroll
            for (int i = 0; i < HE_MAX_KS; ++i)
                if (i < ks.n && r < ks.k[i]) d += (double)ks.scale * wr / cnt[i];
            if (live) out[((size_t)b * N + n) * ldo + g] = (float)(d * (double)go) * w[n];
        }
    }
    if constexpr (!BWD) {
        if (live) {
            float pred = 0.f;                        // pred += out_k / len(ks), in list order (he2rna.py:88-91)
            for (int i = 0; i < ks.n; ++i) pred += (float)(acc[i] / cnt[i]) * ks.scale;
            out[(size_t)b * ldo + g] = pred;
        }
    }
}

int fill_ks(const int32_t* ks, int n_ks, float scale, int N, HeKs* o) {
    SQ_REQUIRE(ks && n_ks >= 1 && n_ks <= HE_MAX_KS, "he2rna: %d values of k (1..%d)", n_ks, HE_MAX_KS);
    for (int i = 0; i < n_ks; ++i) {
        SQ_REQUIRE(ks[i] >= 1 && ks[i] <= N, "he2rna: k=%d out of range for %d tiles (torch.topk would raise)", ks[i], N);
        o->k[i] = ks[i];
    }
    o->n = n_ks;
    o->scale = scale;
    return SQ_OK;
}

}  // namespace

extern "C" int sq_he2rna_tile_mask(const float* x_tokens, int n_rows, int n_channels, float* mask, sq_stream_t stream_) {
    SQ_REQUIRE(x_tokens && mask && n_rows >= 1 && n_channels >= 1, "he2rna_tile_mask: bad arguments");
    hipLaunchKernelGGL(he2rna_mask_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream_, x_tokens, n_rows, n_channels, mask);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}

extern "C" int sq_he2rna_topk_mean(const float* scores, int ld_scores, const float* mask, const int32_t* ks, int n_ks, float scale,
                                   float* out, int B, int N, int G, sq_stream_t stream_) {
    SQ_REQUIRE(scores && mask && out && B >= 1 && G >= 1 && ld_scores >= G, "he2rna_topk_mean: bad arguments");
    SQ_REQUIRE(N >= 1 && N <= HE_MAX_N, "he2rna_topk_mean: %d tiles per slide (1..%d)", N, HE_MAX_N);
    HeKs k;
    if (int e = fill_ks(ks, n_ks, scale, N, &k)) return e;
    const size_t lds = (size_t)N * HE_THREADS * sizeof(float);
    static SqDevOnce attr;       // hipFuncSetAttribute is per device
    if (attr.needed()) {
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)he2rna_topk_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, HE_MAX_N * HE_THREADS * 4));
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)he2rna_topk_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, HE_MAX_N * HE_THREADS * 4));
        attr.done();
    }
    hipLaunchKernelGGL(he2rna_topk_kernel<false>, dim3((G + HE_THREADS - 1) / HE_THREADS, B), dim3(HE_THREADS), lds, (hipStream_t)stream_,
                       scores, ld_scores, mask, k, (const float*)nullptr, out, G, N, G);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}

extern "C" int sq_he2rna_topk_mean_bwd(const float* scores, int ld_scores, const float* mask, const int32_t* ks, int n_ks, float scale,
                                       const float* grad_out, float* grad_scores, int ld_grad, int B, int N, int G, sq_stream_t stream_) {
    SQ_REQUIRE(scores && mask && grad_out && grad_scores && B >= 1 && G >= 1 && ld_scores >= G && ld_grad >= G, "he2rna_topk_mean_bwd: bad arguments");
    SQ_REQUIRE(N >= 1 && N <= HE_MAX_N, "he2rna_topk_mean_bwd: %d tiles per slide (1..%d)", N, HE_MAX_N);
    HeKs k;
    if (int e = fill_ks(ks, n_ks, scale, N, &k)) return e;
    const size_t lds = (size_t)N * HE_THREADS * sizeof(float);
    static SqDevOnce attr;       // hipFuncSetAttribute is per device
    if (attr.needed()) {
        SQ_HIP_CHECK(hipFuncSetAttribute((const void*)he2rna_topk_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, HE_MAX_N * HE_THREADS * 4));
        attr.done();
    }
    hipLaunchKernelGGL(he2rna_topk_kernel<true>, dim3((G + HE_THREADS - 1) / HE_THREADS, B), dim3(HE_THREADS), lds, (hipStream_t)stream_,
                       scores, ld_scores, mask, k, grad_out, grad_scores, ld_grad, N, G);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}
