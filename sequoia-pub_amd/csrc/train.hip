// Training-step kernels: fused MSE loss + gradient, fused AdamW (+ bf16 shadow refresh), and the
// per-batch metrics the reference computes on the host every batch (MAE + mean per-gene Pearson).
#include "../../include/sequoia_hip.h"
#include "sq_common.h"

namespace {

constexpr int RED_BLOCKS = 1024;

__device__ __forceinline__ double block_sum_f64(double v, double* sh) {
    v = wave_sum_f64(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
    __syncthreads();
    return r;   // valid on thread 0
}

// grad = scale * (pred - target); partial[b] = sum (pred - target)^2 over the block's grid-stride slice
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ pred, const float* __restrict__ target, size_t n,
                                                  float scale, float* __restrict__ grad, double* __restrict__ partial) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float d = pred[i] - target[i];
        if (grad) grad[i] = scale * d;
        acc += (double)d * (double)d;
    }
    const double s = block_sum_f64(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void mse_final_kernel(const double* __restrict__ partial, int nblk, double inv_n, float* __restrict__ loss) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) s += partial[i];
    s = block_sum_f64(s, sh);
    if (threadIdx.x == 0) *loss = (float)(s * inv_n);
}

// torch.optim.AdamW(amsgrad=False) single-tensor update (main.py:180-183), fp32 state
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             bf16_t* __restrict__ p_lp, size_t n, float lr, float beta1, float beta2, float eps, float wd,
                             float step_size, float bc2_sqrt, float grad_scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * grad_scale;
        float pi = p[i];
        pi *= (1.0f - lr * wd);
        const float mi = m[i] + (gi - m[i]) * (1.0f - beta1);            // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = v[i] * beta2 + (1.0f - beta2) * gi * gi;        // mul_(beta2).addcmul_(g, g, 1 - beta2)
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi -= step_size * (mi / denom);
        m[i] = mi; v[i] = vi; p[i] = pi;
        if (p_lp) p_lp[i] = f32_to_bf16(pi);
    }
}

// thread = gene: fp64 Pearson r(target[:, g], pred[:, g]) as np.corrcoef; genes with a constant target are
// skipped, NaN r (constant prediction) dropped (he2rna.py:140-149); abs-error sum for sklearn MAE.
__global__ __launch_bounds__(256) void metrics_kernel(const float* __restrict__ pred, const float* __restrict__ target, int B, int G,
                                                      double* __restrict__ partial) {
    __shared__ double sh[4];
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    double r_sum = 0.0, r_cnt = 0.0, ae = 0.0;
    if (g < G) {
        double sy = 0.0, sp = 0.0;
        const float y0 = target[g];
        bool varies = false;
        for (int b = 0; b < B; ++b) {
            const float y = target[(size_t)b * G + g], p = pred[(size_t)b * G + g];
            sy += y; sp += p;
            varies |= (y != y0);
            ae += fabs((double)p - (double)y);
        }
        if (varies) {
            const double my = sy / B, mp = sp / B;
            double syy = 0.0, spp = 0.0, syp = 0.0;
            for (int b = 0; b < B; ++b) {
                const double yc = (double)target[(size_t)b * G + g] - my, pc = (double)pred[(size_t)b * G + g] - mp;
                syy += yc * yc; spp += pc * pc; syp += yc * pc;
            }
            const double r = syp / sqrt(syy * spp);
            if (r == r) { r_sum = r; r_cnt = 1.0; }
        }
    }
    double s;
    s = block_sum_f64(r_sum, sh); if (threadIdx.x == 0) partial[blockIdx.x * 3 + 0] = s;
    s = block_sum_f64(r_cnt, sh); if (threadIdx.x == 0) partial[blockIdx.x * 3 + 1] = s;
    s = block_sum_f64(ae, sh);    if (threadIdx.x == 0) partial[blockIdx.x * 3 + 2] = s;
}

__global__ __launch_bounds__(256) void metrics_final_kernel(const double* __restrict__ partial, int nblk, double inv_bg, float* __restrict__ out) {
    __shared__ double sh[4];
    double rs = 0.0, rc = 0.0, ae = 0.0;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) { rs += partial[i * 3]; rc += partial[i * 3 + 1]; ae += partial[i * 3 + 2]; }
    rs = block_sum_f64(rs, sh); rc = block_sum_f64(rc, sh); ae = block_sum_f64(ae, sh);
    if (threadIdx.x == 0) {
        out[0] = (float)(ae * inv_bg);          // MAE
        out[1] = (float)(rs / rc);              // mean Pearson over valid genes (NaN when none, as np.mean([]))
        out[2] = (float)rc;                     // number of genes that entered the mean
    }
}

}  // namespace

extern "C" size_t sq_train_scratch_bytes(int num_outputs) {
    const size_t a = (size_t)RED_BLOCKS * sizeof(double);
    const size_t b = (size_t)((num_outputs + 255) / 256) * 3 * sizeof(double);
    return (a > b ? a : b) + 256;
}

extern "C" int sq_mse_loss_grad(const float* pred, const float* target, size_t n, float grad_scale, float* grad, float* loss_out,
                                void* scratch, sq_stream_t stream_) {
    SQ_REQUIRE(pred && target && loss_out && scratch && n > 0, "mse: null pointer or n == 0");
    hipStream_t st = (hipStream_t)stream_;
    size_t nb = (n + 255) / 256;
    if (nb > RED_BLOCKS) nb = RED_BLOCKS;
    hipLaunchKernelGGL(mse_kernel, dim3((int)nb), dim3(256), 0, st, pred, target, n, grad_scale, grad, (double*)scratch);
    SQ_LAUNCH_CHECK();
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(256), 0, st, (const double*)scratch, (int)nb, 1.0 / (double)n, loss_out);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}

extern "C" int sq_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, void* params_lp, size_t n,
                             float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                             sq_stream_t stream_) {
    SQ_REQUIRE(params && grads && exp_avg && exp_avg_sq && n > 0 && step >= 1, "adamw: bad arguments (step=%d)", step);
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    size_t nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(adamw_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream_, params, grads, exp_avg, exp_avg_sq,
                       (bf16_t*)params_lp, n, lr, beta1, beta2, eps, weight_decay, step_size, bc2_sqrt, grad_scale);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}

extern "C" int sq_batch_metrics(const float* pred, const float* target, int B, int G, float* out3, void* scratch,
                                sq_stream_t stream_) {
    SQ_REQUIRE(pred && target && out3 && scratch && B > 0 && G > 0, "metrics: bad arguments");
    hipStream_t st = (hipStream_t)stream_;
    const int nb = (G + 255) / 256;
    hipLaunchKernelGGL(metrics_kernel, dim3(nb), dim3(256), 0, st, pred, target, B, G, (double*)scratch);
    SQ_LAUNCH_CHECK();
    hipLaunchKernelGGL(metrics_final_kernel, dim3(1), dim3(256), 0, st, (const double*)scratch, nb, 1.0 / ((double)B * G), out3);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}
