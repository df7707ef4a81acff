// Per-gene evaluation statistics over a test set -- the device side of evaluation/evaluate_model.py:67-96:
// for every gene column of real / pred / random [n, G]: the three Pearson correlations (fp64, centred two-pass
// as scipy.stats.pearsonr does), RMSE(pred), RMSE(random), mean(real), the 25 % / 75 % quantiles of real
// (numpy's default linear method) and the constant-column flag the reference tests with len(set(col)) == 1.
// p-values (Student t / Steiger) are O(G) host arithmetic on these (evalstats.py).
//
// Layout: the [n, G] tables are row-major, so the moment kernel gives one gene to each lane (coalesced across
// genes) and walks the n rows; the quantiles need each column sorted: real is transposed to [G, n] once and a
// workgroup bitonic-sorts one column in LDS.
#include "../../include/sequoia_hip.h"
#include "elementwise.h"

namespace {

constexpr int MAX_N = 8192;

__global__ __launch_bounds__(256) void gene_moments_kernel(const float* __restrict__ real, const float* __restrict__ pred,
                                                           const float* __restrict__ rnd, int n, int G, double* __restrict__ out) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    double sx = 0, sy = 0, sz = 0;
    float x0 = real[g], y0 = pred[g], z0 = rnd[g];
    bool cx = true, cy = true, cz = true;
    for (int i = 0; i < n; ++i) {
        const float x = real[(size_t)i * G + g], y = pred[(size_t)i * G + g], z = rnd[(size_t)i * G + g];
        sx += x; sy += y; sz += z;
        cx = cx && x == x0; cy = cy && y == y0; cz = cz && z == z0;
    }
    const double mx = sx / n, my = sy / n, mz = sz / n;
    double xx = 0, yy = 0, zz = 0, xy = 0, xz = 0, yz = 0, exy = 0, exz = 0;
    for (int i = 0; i < n; ++i) {
        const double xr = real[(size_t)i * G + g], yr = pred[(size_t)i * G + g], zr = rnd[(size_t)i * G + g];
        const double x = xr - mx, y = yr - my, z = zr - mz;
        xx += x * x; yy += y * y; zz += z * z; xy += x * y; xz += x * z; yz += y * z;
        exy += (xr - yr) * (xr - yr); exz += (xr - zr) * (xr - zr);
    }
    auto clip = [](double r) { return r > 1.0 ? 1.0 : (r < -1.0 ? -1.0 : r); };      // pearsonr clips to [-1, 1]
    out[0 * (size_t)G + g] = clip(xy / (sqrt(xx) * sqrt(yy)));
    out[1 * (size_t)G + g] = clip(xz / (sqrt(xx) * sqrt(zz)));
    out[2 * (size_t)G + g] = clip(yz / (sqrt(yy) * sqrt(zz)));
    out[3 * (size_t)G + g] = sqrt(exy / n);
    out[4 * (size_t)G + g] = sqrt(exz / n);
    out[5 * (size_t)G + g] = mx;
    out[8 * (size_t)G + g] = (cx || cy || cz) ? 1.0 : 0.0;
}

// one workgroup per gene: column (contiguous after the transpose) -> LDS -> bitonic sort -> numpy linear quantiles
__global__ __launch_bounds__(256) void gene_quantile_kernel(const float* __restrict__ realT, int n, int ldt, int G, int npow2,
                                                            double* __restrict__ out) {
    extern __shared__ float col[];
    const int g = blockIdx.x;
    for (int i = threadIdx.x; i < npow2; i += 256) col[i] = i < n ? realT[(size_t)g * ldt + i] : INFINITY;
    __syncthreads();
    for (int k = 2; k <= npow2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < npow2; i += 256) {
                const int l = i ^ j;
                if (l > i) {
                    const float a = col[i], b = col[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { col[i] = b; col[l] = a; }
                }
            }
            __syncthreads();
        }
    if (threadIdx.x < 2) {
        // numpy.quantile(method="linear"): virtual index q*(n-1); lerp a + (b-a)*t, taken from the b side when t >= 0.5
        const double q = threadIdx.x == 0 ? 0.25 : 0.75;
        const double pos = q * (n - 1);
        const int lo = (int)floor(pos);
        const int hi = lo + 1 < n ? lo + 1 : n - 1;
        const double t = pos - lo;
        const double a = col[lo], b = col[hi];
        const double v = t >= 0.5 ? b - (b - a) * (1.0 - t) : a + (b - a) * t;
        out[(6 + threadIdx.x) * (size_t)G + g] = v;
    }
}

}  // namespace

extern "C" size_t sq_gene_eval_workspace_bytes(int n, int num_outputs) {
    if (n < 2 || n > MAX_N || num_outputs < 1) return 0;
    return sq_align_up((size_t)num_outputs * (size_t)((n + 3) / 4 * 4) * 4, 256);
}

extern "C" int sq_gene_eval_stats(const float* real, const float* pred, const float* random_pred, int n, int G, double* out9,
                                  void* workspace, size_t workspace_bytes, sq_stream_t stream_) {
    hipStream_t st = (hipStream_t)stream_;
    SQ_REQUIRE(real && pred && random_pred && out9 && workspace, "gene_eval_stats: null pointer");
    SQ_REQUIRE(n >= 2 && n <= MAX_N && G >= 1, "gene_eval_stats: n=%d samples (2..%d), G=%d", n, MAX_N, G);
    const size_t need = sq_gene_eval_workspace_bytes(n, G);
    if (workspace_bytes < need) {
        sq_set_error("gene_eval_stats: workspace %zu < required %zu", workspace_bytes, need);
        return SQ_ERR_WORKSPACE;
    }
    hipLaunchKernelGGL(gene_moments_kernel, dim3((G + 255) / 256), dim3(256), 0, st, real, pred, random_pred, n, G, out9);
    SQ_LAUNCH_CHECK();
    const int ldt = (n + 3) / 4 * 4;
    if (int e = sq_k_transpose(real, G, workspace, ldt, n, G, 4, 1, 0, 0, st)) return e;      // [n, G] -> [G, ldt]
    int npow2 = 2;
    while (npow2 < n) npow2 <<= 1;
    hipLaunchKernelGGL(gene_quantile_kernel, dim3(G), dim3(256), (size_t)npow2 * 4, st, (const float*)workspace, n, ldt, G, npow2, out9);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}
