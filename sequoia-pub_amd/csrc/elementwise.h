// Small memory-bound kernels shared by the ViS / ResNet / training paths.
#pragma once
#include "sq_common.h"

// X[b,n,:] = x[b,n,:] + pos[n,:]   (tformer_lin.py:100); optional bf16 copy
int sq_k_add_pos(const float* x, const float* pos, float* X, bf16_t* Xh, int B, int N, int D, hipStream_t s);
// out[b,:] = mean_n X[b,n,:]       (tformer_lin.py:22 via s(mean x), :103); optional bf16 copy
int sq_k_token_mean(const float* X, float* out, bf16_t* outh, int B, int N, int D, hipStream_t s);
// y = LayerNorm_D(x) * g + b       (rows of length D <= 4096, eps 1e-5); out f32 or bf16; optional mean/rstd save
int sq_k_ln_rows(const float* x, const float* g, const float* b, void* y, int out_dtype, int R, int D,
                 float* mean_out, float* rstd_out, hipStream_t s);
// y = GELU(LayerNorm_64(x) * g + b) per 64-wide head group; x f32 [R, C] with C % 64 == 0, g/b [C]
int sq_k_ln64_gelu(const float* x, const float* g, const float* b, void* y, int out_dtype, int R, int C, hipStream_t s);
// dst = (T) src
int sq_k_f32_to_bf16(const float* src, bf16_t* dst, size_t n, hipStream_t s);
int sq_k_bf16_to_f32(const bf16_t* src, float* dst, size_t n, hipStream_t s);
// dst[c][r] = src[r][c]  (2-byte or 4-byte elements), batched
int sq_k_transpose(const void* src, void* dst, int R, int C, int elem_size, int batch, hipStream_t s);
