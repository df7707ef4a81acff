// Small memory-bound kernels shared by the ViS / ResNet / training paths.
#pragma once
#include "sq_common.h"

// X[b,n,:] = x[b,n,:] + pos[n,:]   (tformer_lin.py:100); optional bf16 copy
int sq_k_add_pos(const float* x, const float* pos, float* X, bf16_t* Xh, int B, int N, int D, hipStream_t s);
// out[b,:] = mean_n X[b,n,:]       (tformer_lin.py:22 via s(mean x), :103); optional bf16 copy
int sq_k_add_pos_gather(const float* src, const int32_t* idx, const float* pos, float* X, bf16_t* Xh, int B, int N, int D, hipStream_t s);
int sq_k_token_mean(const float* X, float* out, bf16_t* outh, int B, int N, int D, hipStream_t s);
int sq_k_token_mean_any(const void* X, int in_dtype, float* out, bf16_t* outh, int B, int N, int D, hipStream_t s);   // X fp32 or bf16
// y = LayerNorm_D(x) * g + b       (rows of length D <= 4096, eps 1e-5); out f32 or bf16; optional mean/rstd save
int sq_k_ln_rows(const float* x, const float* g, const float* b, void* y, int out_dtype, int R, int D,
                 float* mean_out, float* rstd_out, hipStream_t s);
int sq_k_ln_rows_any(const void* x, int in_dtype, const float* g, const float* b, void* y, int out_dtype, int R, int D,
                     float* mean_out, float* rstd_out, hipStream_t s);   // x fp32 or bf16
// y = GELU(LayerNorm_64(x) * g + b) per 64-wide head group; x f32 [R, C] with C % 64 == 0, g/b [C]
int sq_k_ln64_gelu(const float* x, const float* g, const float* b, void* y, int out_dtype, int R, int C, hipStream_t s);
// y[(b, n), :] = GELU(LayerNorm_64(f_tile[idx[b, n], :] + f_pos[n, :]) * g + b) per 64-wide head group; a negative index is a zero row of
// f_tile.  f_tile f32 [rows, C], f_pos f32 [N, C], idx int32 [B, N]: the first layer's local projection of a window batch taken from
// per-TILE projections (the projection is linear in tile feature + position; sq_vis_forward_tiles)
int sq_k_gather_ln64_gelu(const float* f_tile, const float* f_pos, const int32_t* idx, const float* g, const float* b, void* y, int out_dtype,
                          int B, int N, int C, hipStream_t s);
// dst = (T) src
int sq_k_f32_to_bf16(const float* src, bf16_t* dst, size_t n, hipStream_t s);
int sq_k_bf16_to_f32(const bf16_t* src, float* dst, size_t n, hipStream_t s);
// dst[c][r] = src[r][c] for r < R, 0 for R <= r < ldd  (2- or 4-byte elements; src leading dim lds, dst ldd >= R);
// batched with element strides.  The zero padding lets ragged R feed the GEMM's 16-byte K chunks.
int sq_k_transpose(const void* src, int lds_, void* dst, int ldd, int R, int C, int elem_size, int batch,
                   long long sstride, long long dstride, hipStream_t s);
// the same, many matrices in ONE launch (a backward pass transposes every weight once)
#define SQ_MAX_TRANSPOSE_JOBS 64
struct sq_transpose_job {
    const void* src; void* dst;
    long long sstride, dstride;        // element strides between the matrices of a batched job
    int lds, ldd, R, C, tile0;
};
struct sq_transpose_jobs {
    sq_transpose_job job[SQ_MAX_TRANSPOSE_JOBS];
    int n = 0, tiles = 0;
};
int sq_transpose_jobs_add(sq_transpose_jobs* jobs, const void* src, int lds_, void* dst, int ldd, int R, int C, int batch,
                          long long sstride, long long dstride);
int sq_k_transpose_multi(const sq_transpose_jobs& jobs, int elem_size, hipStream_t s);
// dst[r][0:C] = (T) src[r][0:C], dst[r][C:ldd] = 0
int sq_k_cast_pad(const float* src, int lds_, void* dst, int dst_dtype, int ldd, int R, int C, hipStream_t s);

// ---- backward helpers -------------------------------------------------------------------------
// Deferred column sums: the LayerNorm backward kernels leave <= 512 partial [dg | db] rows; instead of one small
// reduction launch each, a pass can collect them as jobs and finish a whole layer's worth with ONE launch.
#define SQ_MAX_COLSUM_JOBS 8
struct sq_colsum_job { const float* x; float* out; float* out2; int R, C, ld, split, blk0; };
struct sq_colsum_jobs { sq_colsum_job job[SQ_MAX_COLSUM_JOBS]; int n = 0, blocks = 0; };
int sq_colsum_jobs_add(sq_colsum_jobs* jobs, const float* x, int R, int C, int ld, float* out, float* out2, int split);
int sq_k_colsum_multi(const sq_colsum_jobs& jobs, hipStream_t s);
// out[c] = sum_r x[r, c]   (x f32 or bf16, leading dim ld); ws: >= colsum_ws_floats(C) floats
size_t sq_colsum_ws_floats(int C);
int sq_k_colsum(const void* x, int dtype, int R, int C, int ld, float* ws, float* out, hipStream_t s);
// out[g, c] = scale * sum_{n<N} x[g*N + n, c]     (x f32 or bf16 [G*N, C])
int sq_k_group_sum(const void* x, int dtype, int G, int N, int C, float scale, float* out, hipStream_t s);
// dst[b, n, :] = scale * src[b, :]; optional bf16 copy
int sq_k_bcast_rows(const float* src, float scale, float* dst, bf16_t* dst_lp, int B, int N, int D, hipStream_t s);
// out[n, :] = sum_b x[b, n, :]
int sq_k_batch_sum(const float* x, float* out, int B, int ND, hipStream_t s);
// LayerNorm_D backward: dx = dres + dLN(dy; x, g);  dg/db = column sums.  ws >= ln_bwd_ws_floats(D) floats
size_t sq_ln_bwd_ws_floats(int D);
int sq_k_ln_rows_bwd(const float* dy, const float* x, const float* g, const float* dres, float* dx, bf16_t* dx_lp,
                     float* dg, float* db, float* ws, int R, int D, hipStream_t s, sq_colsum_jobs* defer = nullptr);
// the same with dy / x / dres in fp32 or bf16 and an optional fp32 result (dx may be null when dx_lp is given): the lean bf16 training stream
int sq_k_ln_rows_bwd_any(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* g, const void* dres, int dres_dtype,
                         float* dx, bf16_t* dx_lp, float* dg, float* db, float* ws, int R, int D, hipStream_t s, sq_colsum_jobs* defer = nullptr);
int sq_k_ln64_gelu_bwd_any(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* g, const float* b, void* dx, int out_dtype,
                           float* dg, float* db, float* ws, int R, int C, hipStream_t s, sq_colsum_jobs* defer = nullptr);
// backward of y = GELU(LN64(x)*g + b): dx (f32 or bf16 by out_dtype), dg/db [C].  ws >= ln_bwd_ws_floats(C)
int sq_k_ln64_gelu_bwd(const float* dy, const float* x, const float* g, const float* b, void* dx, int out_dtype,
                       float* dg, float* db, float* ws, int R, int C, hipStream_t s, sq_colsum_jobs* defer = nullptr);
