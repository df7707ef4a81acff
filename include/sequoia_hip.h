/* libsequoia_hip.so -- C ABI of the MI355X-native SEQUOIA hot path.
 *
 * The reference (gevaertlab/sequoia-pub) has no FFI / plugin interface: the path is
 * reached through three Python object interfaces.  Each entry
 * point below states the reference interface it stands behind (file:line under
 * /root/reference).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions: every function returns 0 on success and a negative code on error
 * (message: sq_last_error(), thread-local).  Nothing here allocates device memory:
 * the caller (PyTorch-ROCm is only the allocator) passes raw device pointers,
 * element counts and a workspace whose size is queried with *_workspace_bytes.
 * Every call is asynchronous on the given hipStream_t.  Plain C types only.
 */
#ifndef SEQUOIA_HIP_H
#define SEQUOIA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sq_stream_t; /* hipStream_t */
typedef void* sq_event_t;  /* hipEvent_t */

#define SQ_DTYPE_F32 0  /* exact fp32 MFMA (v_mfma_f32_32x32x2_f32): parity mode */
#define SQ_DTYPE_BF16 1 /* bf16 MFMA, fp32 accumulate: perf mode              */
#define SQ_DTYPE_BF16X3 2 /* split bf16 (sq_resnet50_extract only): every fp32 value as hi + lo bf16 planes, a.b = a_hi.b_hi + a_hi.b_lo +
                             a_lo.b_hi on bf16 MFMAs with fp32 accumulation -- 2^-18 per operand, fp32's exponent range */
#define SQ_DTYPE_F16X3 3  /* the same with fp16 planes: 22 significant bits (fp32-class results), values must stay below 65504
                             (an overflow propagates to the features as NaN and raises sq_resnet50_extract_checked's flag); the fast parity mode */

#define SQ_MAX_DEPTH 16
#define SQ_HEAD_DIM 64 /* dimensions_f = dimensions_s = dimensions_c = 64 (src/main.py:147,167,202) */

const char* sq_last_error(void);
int sq_version(void);
/* 1 when a gfx950 device is visible to the HIP runtime, else 0 (never fails) */
int sq_device_ok(void);

/* HIP-event timing of the library's own launches (bench.py roofline leg; no reference
 * counterpart).  sq_prof_enable(1) makes every instrumented launch record two events on
 * its stream; sq_prof_report writes a JSON array of {name,count,total_ms,flops,bytes}
 * (flops/bytes = algorithmic work per launch) into buf and clears the records.
 * sq_prof_enable(2) additionally brackets every instrumented launch with marker launches whose grid size
 * carries the launch's class number ((id + 2) blocks of 64 threads in front, one block behind), so that a
 * dispatch-ordered counter trace (rocprofv3 --kernel-trace --pmc) can be attributed to classes without
 * knowing kernel symbols; sq_prof_marker_names writes the JSON array id -> class name. */
int sq_prof_enable(int on);
int sq_prof_report(char* buf, size_t cap);
int sq_prof_marker_names(char* buf, size_t cap);

/* ------------------------------------------------------------------------------
 * ViS aggregator  (src/tformer_lin.py:80-106 ViS; :64-77 SummaryTransformer;
 * :29-48 MultiHeadSummary; :7-26 SummaryMixing; :51-61 FeedForward)
 * ---------------------------------------------------------------------------- */
typedef struct sq_vis_config {
    int32_t input_dim;    /* D: 1024 (UNI) or 2048 (ResNet-50); multiple of 64 */
    int32_t depth;        /* main.py:36 default 6 */
    int32_t nheads;       /* main.py:37 default 16 */
    int32_t num_outputs;  /* G = 20820 genes */
    int32_t num_clusters; /* 100 tokens (tformer_lin.py:83) */
} sq_vis_config;

/* Offsets (in elements) of every reference tensor inside ONE flat parameter buffer.
 * Per-head tensors are stored head-major and contiguous, so each reference tensor
 * `transformer.layers.{l}.0.mixers.{h}.f.weight` etc. is a contiguous slice. */
typedef struct sq_vis_layer_offsets {
    int64_t f_w, f_b;       /* mixers.{h}.f.{weight,bias}              [H][64][D], [H][64] */
    int64_t s_w, s_b;       /* mixers.{h}.s.{weight,bias}              [H][64][D], [H][64] */
    int64_t lnf_g, lnf_b;   /* mixers.{h}.local_norm.{weight,bias}     [H][64]             */
    int64_t lns_g, lns_b;   /* mixers.{h}.summary_norm.{weight,bias}   [H][64]             */
    int64_t c_w, c_b;       /* mixers.{h}.c.{weight,bias}              [H][64][128], [H][64] */
    int64_t proj_w, proj_b; /* projection.{weight,bias}                [D][H*64], [D]      */
    int64_t ffln_g, ffln_b; /* net.0 LayerNorm(D)                                           */
    int64_t ff1_w, ff1_b;   /* net.1 Linear(D, D)                                           */
    int64_t ff2_w, ff2_b;   /* net.3 Linear(D, D)                                           */
} sq_vis_layer_offsets;

typedef struct sq_vis_layout {
    int64_t pos;                  /* pos_emb1D [num_clusters][D] */
    int64_t head_ln_g, head_ln_b; /* linear_head.0 */
    int64_t head_w, head_b;       /* linear_head.1 [G][D], [G] */
    int64_t total;                /* elements in the flat buffer */
    sq_vis_layer_offsets layer[SQ_MAX_DEPTH];
} sq_vis_layout;

int sq_vis_layout_init(const sq_vis_config* cfg, sq_vis_layout* out);

size_t sq_vis_workspace_bytes(const sq_vis_config* cfg, int dtype, int batch, int save_for_backward);

/* ViS.forward (tformer_lin.py:97-106): x f32 [B, num_clusters, D] -> out f32 [B, G].
 * params: flat fp32 buffer (sq_vis_layout); params_lp: bf16 copy of the same buffer
 * (dtype == SQ_DTYPE_BF16) or NULL.  save_for_backward keeps the per-layer activations
 * sq_vis_backward needs in the workspace. */
int sq_vis_forward(const sq_vis_config* cfg, int dtype, const float* params, const void* params_lp, const float* x,
                   float* out, int batch, int save_for_backward, void* workspace, size_t workspace_bytes,
                   sq_stream_t stream);

/* sq_vis_forward with two options the sliding-window path (spatial_vis/visualize.py:35-102) uses:
 *  - gather: instead of x, give gather_src f32 [gather_rows, D] (the tile-feature cache) and gather_idx int32
 *    [B, num_clusters]; token (b, t) is row gather_idx[b, t] of the cache, a negative index is a zero row (the
 *    zero padding of visualize.py:72-75) -- the [B, 100, D] window batch is never materialised;
 *  - head_in: instead of out, receive the linear head's INPUT LayerNorm(mean_tokens X) as f32 [B, D]; the head is
 *    linear, so the per-tile mean over windows (visualize.py:97-100) can be taken on these D-vectors and the head
 *    applied once per tile (sq_linear) instead of materialising [n_windows, G] predictions.
 * Exactly one of (x | gather_src + gather_idx) and exactly one of (out | head_in) must be given. */
int sq_vis_forward_ex(const sq_vis_config* cfg, int dtype, const float* params, const void* params_lp, const float* x,
                      const float* gather_src, const int32_t* gather_idx, int gather_rows, float* out, float* head_in,
                      int batch, int save_for_backward, void* workspace, size_t workspace_bytes, sq_stream_t stream);

/* The gather + head_in form of sq_vis_forward_ex for the window loop of spatial_vis/visualize.py:46-82, with the first layer's
 * local projection f (src/tformer_lin.py:20, mixers.{h}.f of layer 0, all heads) taken from per-TILE projections: f is linear in
 * x = tile feature + pos_emb1D (tformer_lin.py:99-100), so for token (b, t)  f(x) = f_tile[gather_idx[b, t]] + f_pos[t]  with
 *   f_tile f32 [gather_rows, nheads*64] = cache . Wf^T            (no bias; a negative index contributes a zero row)
 *   f_pos  f32 [num_clusters, nheads*64] = pos_emb1D . Wf^T + bf
 * computed by the caller once per slide (two sq_linear calls) instead of once per window token.  Inference only. */
int sq_vis_forward_tiles(const sq_vis_config* cfg, int dtype, const float* params, const void* params_lp, const float* gather_src,
                         const int32_t* gather_idx, int gather_rows, const float* f_tile, const float* f_pos, float* head_in,
                         int batch, void* workspace, size_t workspace_bytes, sq_stream_t stream);

/* Backward of ViS.forward -- replaces torch autograd over tformer_lin.py in the training loop
 * (src/vit.py:163-180 `loss.backward()`).  grad_out f32 [B, G]; grad_params: flat f32 buffer with
 * the parameter layout, fully overwritten; grad_x f32 [B, num_clusters, D] or NULL.
 * fwd_workspace must be the workspace of the matching sq_vis_forward(save_for_backward = 1). */
size_t sq_vis_backward_workspace_bytes(const sq_vis_config* cfg, int dtype, int batch);
int sq_vis_backward(const sq_vis_config* cfg, int dtype, const float* params, const void* params_lp,
                    const float* grad_out, float* grad_params, float* grad_x, int batch, void* fwd_workspace,
                    size_t fwd_workspace_bytes, void* bwd_workspace, size_t bwd_workspace_bytes, sq_stream_t stream);

/* Data-parallel training (src/main.py DDP-less reference; BASELINE config 4): the flat gradient splits into
 * depth + 1 contiguous buckets [lo, hi) (elements), listed in the order the backward pass finishes them (head
 * first, then layers last to first; layer 0's bucket also holds pos_emb1D).  sq_vis_backward_buckets records
 * bucket_events[i] (hipEvent_t) on `stream` as soon as bucket i is final, so the caller can start that bucket's
 * RCCL all-reduce on another stream while the rest of the backward pass still runs.  Returns the bucket count. */
int sq_vis_grad_buckets(const sq_vis_config* cfg, int64_t* lo, int64_t* hi, int cap);
int sq_vis_backward_buckets(const sq_vis_config* cfg, int dtype, const float* params, const void* params_lp,
                            const float* grad_out, float* grad_params, float* grad_x, int batch, void* fwd_workspace,
                            size_t fwd_workspace_bytes, void* bwd_workspace, size_t bwd_workspace_bytes, sq_stream_t stream,
                            const sq_event_t* bucket_events, int n_bucket_events);


/* ------------------------------------------------------------------------------
 * Softmax ViT baseline  (src/vit.py:49-115: Attention :49-74, Transformer :76-89, ViT :91-115;
 * `--model_type vit`, src/main.py:160-163: mlp_dim 2048, dim_head 64).  Same conventions as ViS:
 * one flat fp32 parameter buffer (sq_vit_layout), optional bf16 shadow, workspace from *_workspace_bytes.
 * ---------------------------------------------------------------------------- */
typedef struct sq_vit_config {
    int32_t dim, depth, heads, mlp_dim, num_outputs, num_clusters; /* dim_head = 64 */
} sq_vit_config;
typedef struct sq_vit_layer_offsets {
    int64_t ln1_g, ln1_b; /* layers.{l}.0.norm                         */
    int64_t qkv_w;        /* layers.{l}.0.to_qkv.weight [3*heads*64][dim] */
    int64_t out_w;        /* layers.{l}.0.to_out.weight [dim][heads*64]   */
    int64_t ln2_g, ln2_b; /* layers.{l}.1.net.0                         */
    int64_t ff1_w, ff1_b; /* layers.{l}.1.net.1 Linear(dim, mlp_dim)    */
    int64_t ff2_w, ff2_b; /* layers.{l}.1.net.3 Linear(mlp_dim, dim)    */
} sq_vit_layer_offsets;
typedef struct sq_vit_layout {
    int64_t pos, head_ln_g, head_ln_b, head_w, head_b, total;
    sq_vit_layer_offsets layer[SQ_MAX_DEPTH];
} sq_vit_layout;
int sq_vit_layout_init(const sq_vit_config* cfg, sq_vit_layout* out);
size_t sq_vit_workspace_bytes(const sq_vit_config* cfg, int dtype, int batch, int save_for_backward);
int sq_vit_forward(const sq_vit_config* cfg, int dtype, const float* params, const void* params_lp, const float* x,
                   float* out, int batch, int save_for_backward, void* workspace, size_t workspace_bytes,
                   sq_stream_t stream);
size_t sq_vit_backward_workspace_bytes(const sq_vit_config* cfg, int dtype, int batch);
int sq_vit_backward(const sq_vit_config* cfg, int dtype, const float* params, const void* params_lp,
                    const float* grad_out, float* grad_params, float* grad_x, int batch, void* fwd_workspace,
                    size_t fwd_workspace_bytes, void* bwd_workspace, size_t bwd_workspace_bytes, sq_stream_t stream);

/* ------------------------------------------------------------------------------
 * Training-step pieces of src/vit.py:117-243 `train`
 * ---------------------------------------------------------------------------- */
size_t sq_train_scratch_bytes(int num_outputs);
/* nn.MSELoss() (vit.py:129,166): *loss_out = mean((pred-target)^2) over n elements (device scalar);
 * grad[i] = grad_scale * (pred[i]-target[i])  (grad_scale = 2/n for the plain loss; grad may be NULL). */
int sq_mse_loss_grad(const float* pred, const float* target, size_t n, float grad_scale, float* grad, float* loss_out,
                     void* scratch, sq_stream_t stream);
/* torch.optim.AdamW(lr, amsgrad=False, weight_decay) (src/main.py:180-183) on a flat fp32 buffer;
 * step is the 1-based step count; grads are multiplied by grad_scale first (1/world for an
 * all-reduced sum); params_lp (bf16 shadow) is refreshed in the same pass when not NULL. */
int sq_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, void* params_lp, size_t n,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                  sq_stream_t stream);
/* Per-batch metrics the reference computes on the host every batch (vit.py:167-168):
 * out3[0] = sklearn mean_absolute_error, out3[1] = compute_correlations (src/he2rna.py:140-149:
 * mean per-gene Pearson r over genes with a non-constant target, NaN r dropped), out3[2] = #genes used. */
int sq_batch_metrics(const float* pred, const float* target, int batch, int num_outputs, float* out3, void* scratch,
                     sq_stream_t stream);

/* Per-gene test-set statistics of evaluation/evaluate_model.py:67-96 for real / pred / random [n, G] (row-major,
 * n = test slides <= 8192).  out9 is double [9][G]: 0 r(real, pred), 1 r(real, random), 2 r(pred, random)
 * (scipy.stats.pearsonr: centred, clipped to [-1, 1]; NaN when a column is constant), 3 RMSE(real, pred),
 * 4 RMSE(real, random), 5 mean(real), 6 / 7 the 0.25 / 0.75 quantiles of real (numpy linear method),
 * 8 = 1.0 when any of the three columns is constant (the reference's len(set(col)) == 1 branch).
 * The p-values (Student t, Steiger) are O(G) host arithmetic on these. */
size_t sq_gene_eval_workspace_bytes(int n, int num_outputs);
int sq_gene_eval_stats(const float* real, const float* pred, const float* random_pred, int n, int num_outputs, double* out9,
                       void* workspace, size_t workspace_bytes, sq_stream_t stream);

/* Per-tile vote of sliding-window predictions (spatial_vis/visualize.py:86-101): win_pred f32 [n_windows, G];
 * tile_windows int32 [n_tiles, max_votes] = the windows containing each tile in visiting order, packed, -1 padded.
 * mode 0: out[t] = mean of the listed rows (stride < 10, :97-101); mode 1: the last listed row (stride 10: later
 * windows overwrite, :90-92).  Tiles in no kept window get `fill`.  out f32 [n_tiles, G]. */
int sq_window_vote(const float* win_pred, int n_windows, int num_outputs, const int32_t* tile_windows, int n_tiles,
                   int max_votes, int mode, float fill, float* out, sq_stream_t stream);

/* ------------------------------------------------------------------------------
 * Per-slide k-Means + cluster means  (pre_processing/kmean_features.py:96-108:
 *   KMeans(n_clusters=100, random_state=0).fit(features).labels_ ; per-label np.mean(...))
 * scikit-learn semantics (k-means++ with 2+log(k) local trials, Lloyd, max_iter 300, tol 1e-4),
 * batched over n_slides independent slides of n_samples x dim fp32 features each.
 * The MT19937 draws are data-independent, so the host passes them in: first_center =
 * RandomState(0).choice(n) and uniforms f64 [n_clusters-1, n_local_trials] (device memory).
 * Outputs (device): labels i32 [S, n]; cluster_features f32 [S, k, dim] (may be NULL);
 * seed_indices i32 [S, k] (may be NULL); n_iter i32 [S] (may be NULL).
 * Synchronises the stream between Lloyd bursts (the stop decision is read on the host).
 * ---------------------------------------------------------------------------- */
size_t sq_kmeans_workspace_bytes(int n_slides, int n_samples, int dim, int n_clusters);
int sq_kmeans_fit(const float* X, int n_slides, int n_samples, int dim, int n_clusters, int first_center,
                  const double* uniforms, int n_local_trials, int max_iter, double tol, int32_t* labels,
                  float* cluster_features, int32_t* seed_indices, int32_t* n_iter, void* workspace,
                  size_t workspace_bytes, sq_stream_t stream);

/* ------------------------------------------------------------------------------
 * ResNet-50 patch embedding  (src/resnet.py:155-170 forward_extract; :73-93 Bottleneck;
 * :98-136 topology; eval-mode BN; patch transform pre_processing/compute_features_hdf5.py:49-51,119-120)
 * Weights are packed by the caller per sq_resnet50_layout: conv i at w_off (elements) as
 * [cout][kh][kw][cin] (conv 0: K = 147 zero-padded to k_padded = 152) with eval-mode BatchNorm
 * folded in, bias (fp32) at b_off.  Conv order: conv1; then per bottleneck conv1, conv2, conv3,
 * [downsample.0 for the first block of each layer].
 * dtype SQ_DTYPE_BF16X3 / SQ_DTYPE_F16X3: `weights` = the 16-bit hi plane [w_total] followed by the lo plane [w_total]
 * (hi = cvt(w'), lo = cvt(w' - hi)) of the packed fp32 weights w' = w * s[cout], and `bias` = [b_total] biases followed by
 * [b_total] per-output-channel factors 1 / s (at the same b_off): s = 1 for bf16 planes; for fp16 planes a power of two
 * that lifts each weight row to max |w'| in (128, 256] so that the lo plane stays in fp16's normal range.
 * In these two dtypes every convolution behind the stem (i >= 1) is stored K-TILE-MAJOR inside its block of either plane:
 * element (n, k) of [cout][k_padded] at w_off + ((k / 32) * cout + n) * 32 + k % 32 (k_padded % 32 == 0 for all of them), so
 * that a 32-deep K-tile of consecutive output channels is one contiguous run; conv 0 keeps [64][152] rows.
 * sq_resnet50_extract: give EITHER patches_u8 (uint8 NHWC [n, S, S, 3]: /255 and ImageNet
 * normalisation fused) OR patches_f32_nchw (fp32 [n, 3, S, S], already normalised: the tensor the
 * reference feeds forward_extract).  features: f32 [n, 2048].  S in {224, 256, ...}, multiple of 32.
 * ---------------------------------------------------------------------------- */
#define SQ_RESNET50_CONVS 53
typedef struct sq_conv_desc {
    int64_t w_off, b_off;
    int32_t cin, cout, k, stride, pad, k_padded;
} sq_conv_desc;
typedef struct sq_resnet50_layout {
    sq_conv_desc conv[SQ_RESNET50_CONVS];
    int64_t w_total, b_total;
} sq_resnet50_layout;
int sq_resnet50_layout_init(sq_resnet50_layout* out);
size_t sq_resnet50_workspace_bytes(int dtype, int n_patches, int patch_size);
int sq_resnet50_extract(int dtype, const void* weights, const float* bias, const uint8_t* patches_u8,
                        const float* patches_f32_nchw, int n_patches, int patch_size, float* features,
                        void* workspace, size_t workspace_bytes, sq_stream_t stream);
/* The same, with a guard for the reduced range of SQ_DTYPE_F16X3: nonfinite_flag (device word, caller-zeroed, may be NULL)
 * gets bit 0 OR-ed in when any pooled feature of this call is not finite.  An activation >= 65504 anywhere in the network
 * becomes inf planes, NaN in the next product, and the split modes' ReLU lets NaN through (csrc/x3_fmt.h), so every such
 * overflow reaches the features and the flag; the caller re-runs those patches in SQ_DTYPE_F32 (resnet.py does).
 * The reference computes src/resnet.py:155-170 in fp32, where this cannot happen below 3.4e38. */
int sq_resnet50_extract_checked(int dtype, const void* weights, const float* bias, const uint8_t* patches_u8,
                                const float* patches_f32_nchw, int n_patches, int patch_size, float* features,
                                void* workspace, size_t workspace_bytes, uint32_t* nonfinite_flag, sq_stream_t stream);

/* dst_bf16[i] = bf16(src[i]) -- refresh of the bf16 parameter shadow after an optimizer step */
int sq_cast_f32_to_bf16(const float* src, void* dst_bf16, size_t n, sq_stream_t stream);
/* dst[i] = fp32(src_bf16[i]) -- with the call above the pack / unpack of the bf16 gradient exchange (BASELINE config 4: "RCCL grad
 * all-reduce over xGMI, bf16"; the reference trains on one device, src/main.py:78, so there is no counterpart to cite): a gradient
 * bucket is cast to bf16, summed over the ranks in bf16 (half the ring traffic of the fp32 form), cast back into the fp32 flat
 * gradient that AdamW reads (train.py FusedTrainStep) */
int sq_cast_bf16_to_f32(const void* src_bf16, float* dst, size_t n, sq_stream_t stream);

/* ------------------------------------------------------------------------------
 * Fused linear layer  C = act(A . W^T + bias + residual)   -- the nn.Linear call sites of
 * src/tformer_lin.py:14-16,37,55,57,93 and the 1x1 convolutions of src/resnet.py:60,66.
 * A [M,K] (lda) and W [N,K] (ldw) are `dtype` (f32 or bf16), bias f32 [N] or NULL,
 * residual [M,N] (ldres; res_dtype f32 or bf16) or NULL, act: 0 none, 1 exact-erf GELU, 2 ReLU,
 * C [M,N] (ldc) in out_dtype.  K, lda, ldw multiples of 16 bytes / element size.
 * workspace (optional fp32 scratch, may be NULL): lets skinny problems run split-K.
 * ---------------------------------------------------------------------------- */
int sq_linear(int dtype, const void* A, int lda, const void* W, int ldw, const float* bias, const void* residual,
              int ldres, int res_dtype, int act, void* C, int out_dtype, int ldc, int M, int N, int K, void* workspace,
              size_t workspace_bytes, sq_stream_t stream);

/* Split-bf16 ("bf16x3") linear layer / convolution -- the arithmetic of SQ_DTYPE_BF16X3, exposed for the 1x1 / 3x3
 * convolutions of src/resnet.py:60-66 on caller-owned planes: every fp32 tensor is a bf16 hi plane (bf16(v)) and a
 * bf16 lo plane (bf16(v - hi)) of the same shape;  C = act(A . W^T + bias + residual) with
 * a.w = a_hi.w_hi + a_hi.w_lo + a_lo.w_hi on bf16 MFMAs, fp32 accumulation.  Output: hi / lo planes (C_hi, C_lo) or
 * fp32 (C_f32) -- give exactly one.  act: 0 none, 2 ReLU.  K, N, lda, ldw, ldc, ldres multiples of 8; planes 16-byte
 * aligned and a multiple of 16 bytes apart.  fmt: 0 = bf16 planes, 1 = fp16 planes.  colscale (optional, [N] f32): factor on
 * the accumulator column before the bias (undoes a power-of-two pre-scaling of the weight rows).  conv_geom: NULL for a plain [M,K] A, or
 * {n_img, H, W, Cin, OH, OW, KW, stride, pad} for an implicit-GEMM view of an NHWC activation (K = KH*KW*Cin, Cin % 32 == 0). */
int sq_linear_x3(int fmt, const void* A_hi, const void* A_lo, int lda, const void* W_hi, const void* W_lo, int ldw, const float* bias,
                 const float* colscale, const void* res_hi, const void* res_lo, int ldres, int act, void* C_hi, void* C_lo, float* C_f32, int ldc,
                 int M, int N, int K, const int* conv_geom, sq_stream_t stream);

/* Weight gradient of a linear layer: dW[N_out, N_in] (f32, lddw) = dY[T, N_out]^T . X[T, N_in], contracting
 * over the T token rows, straight from the token-major tensors (the `loss.backward()` of the nn.Linear call
 * sites above, src/vit.py:178).  dY (lddy) and X (ldx) are `dtype`; N_in % 8 == 0; lddy >= round_up(N_out, 8).
 * dbias (optional, [N_out] f32): the bias gradient sum_t dY[t, :], computed by the same launch.
 * workspace: optional fp32 scratch enabling deterministic split-K over the tokens. */
int sq_linear_weight_grad(int dtype, const void* dY, int lddy, const void* X, int ldx, float* dW, int lddw, float* dbias,
                          int n_out, int n_in, int n_tokens, void* workspace, size_t workspace_bytes, sq_stream_t stream);
/* The same for up to FOUR same-shape layers in ONE launch (host arrays of device pointers; dbias: NULL, or one pointer per member):
 * the weight gradients of a transformer layer's linear maps (src/tformer_lin.py:18-26,54-57 under `loss.backward()`, src/vit.py:178) are
 * independent products whose 128 x 128 tile grids fill a quarter of the chip each -- together they fill it without slicing the tokens.
 * bf16 members whose extents are multiples of 128 run on the four-stage ring form of the kernel (csrc/gemm_tn.hip). */
int sq_linear_weight_grad_group(int dtype, int n_members, const void* const* dY, const void* const* X, float* const* dW,
                                float* const* dbias, int lddy, int ldx, int lddw, int n_out, int n_in, int n_tokens, sq_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * UNI patch embedder: timm ``vit_large_patch16_224`` with ``init_values=1e-5, num_classes=0`` as
 * /root/reference/pre_processing/compute_features_hdf5.py:62-68 and spatial_vis/visualize.py:220-232 build it
 * (``feat_model(image)`` at :127-129 returns the normalised class token, [1, 1024]).  timm is not vendored in the
 * reference and absent here: the entry points follow timm's published VisionTransformer algorithm.
 *
 * sq_uni_layout: element offsets into ONE flat fp32 buffer; every timm tensor (patch_embed.proj, cls_token, pos_embed,
 * blocks.{i}.{norm1,attn.qkv,attn.proj,ls1,norm2,mlp.fc1,mlp.fc2,ls2}, norm) is a contiguous slice.
 * sq_uni_forward: params = that buffer (biases / LayerNorm / embeddings are read from it); params_exec = the same layout
 * in the compute dtype with the LayerScale gains folded into attn.proj / mlp.fc2 WEIGHTS; bias_exec = fp32 buffer of
 * the same layout whose attn.proj / mlp.fc2 BIASES carry the gains.  Give EITHER patches_u8 (uint8 NHWC [n, S, S, 3]:
 * ToTensor + Normalize of compute_features_hdf5.py:53-56 are fused in; S must equal img_size) OR patches_f32_nchw
 * (normalised, the reference's tensor).  out: f32 [n, dim].
 * ---------------------------------------------------------------------------------------------------------- */
#define SQ_UNI_MAX_DEPTH 32
typedef struct sq_uni_config { int32_t dim, depth, heads, mlp_dim, img_size; } sq_uni_config;
typedef struct sq_uni_layer_offsets {
    int64_t ln1_g, ln1_b, qkv_w, qkv_b, proj_w, proj_b, ls1, ln2_g, ln2_b, fc1_w, fc1_b, fc2_w, fc2_b, ls2;
} sq_uni_layer_offsets;
typedef struct sq_uni_layout {
    int64_t patch_w, patch_b, cls, pos, norm_g, norm_b, total;
    sq_uni_layer_offsets layer[SQ_UNI_MAX_DEPTH];
} sq_uni_layout;
int sq_uni_layout_init(const sq_uni_config* cfg, sq_uni_layout* out);
size_t sq_uni_workspace_bytes(const sq_uni_config* cfg, int dtype, int n_patches);
int sq_uni_forward(const sq_uni_config* cfg, int dtype, const float* params, const void* params_exec, const float* bias_exec,
                   const uint8_t* patches_u8, const float* patches_f32_nchw, int n_patches, float* out, void* workspace,
                   size_t workspace_bytes, sq_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * HE2RNA comparator (/root/reference/src/he2rna.py:42-106, built by src/pretrain_gtex.py:102-105).
 * The per-tile MLP (1x1 Conv1d layers, he2rna.py:101-106) is sq_linear on the token-major tensor [B * N, C]; these
 * entry points are forward_fixed_k's masking and top-k aggregation (:93-99) and its gradient.
 *   sq_he2rna_tile_mask : mask[row] = 1 if max_c x_tokens[row, c] > 0 else 0           (:94-95; f32 [n_rows, C])
 *   sq_he2rna_topk_mean : out[b, g] = scale * sum_i  (sum_{j<k_i} sorted_desc(s[b, :, g])[j] * mask[b, j]) /
 *                         (sum_{j<k_i} mask[b, j]),   s[b, n, g] = scores[b, n, g] * mask[b, n]   (:96-98);
 *                         scale = 1 for one k (training, :85-86), 1 / len(ks) for the eval mean over ks (:88-91).
 *                         scores f32 [B, N, ld_scores >= G] token-major, N <= 128 tiles, 1 <= k_i <= N, out f32 [B, G].
 *                         0/0 (the first k tiles of a slide all masked) is NaN, as in the reference.
 *   sq_he2rna_topk_mean_bwd : grad_scores f32 [B, N, ld_grad >= G] from grad_out f32 [B, G] (columns >= G untouched).
 * ---------------------------------------------------------------------------------------------------------- */
int sq_he2rna_tile_mask(const float* x_tokens, int n_rows, int n_channels, float* mask, sq_stream_t stream);
int sq_he2rna_topk_mean(const float* scores, int ld_scores, const float* mask, const int32_t* ks, int n_ks, float scale,
                        float* out, int batch, int n_tiles, int n_genes, sq_stream_t stream);
int sq_he2rna_topk_mean_bwd(const float* scores, int ld_scores, const float* mask, const int32_t* ks, int n_ks, float scale,
                            const float* grad_out, float* grad_scores, int ld_grad, int batch, int n_tiles, int n_genes,
                            sq_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SEQUOIA_HIP_H */
