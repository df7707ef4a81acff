// MFMA "NT" GEMM engine of libsequoia_hip:
//   C[M,N] = act(alpha * A[M,K] . B[N,K]^T + bias + rowbias + res) * GELU'(gelu_grad_of)
// Both operands are K-contiguous (activations row-major, nn.Linear / packed conv weights
// [out, in]).  A can also be an implicit-GEMM view of an NHWC activation (conv taps).
#pragma once
#include "sq_common.h"

#define SQ_ACT_NONE 0
#define SQ_ACT_GELU 1
#define SQ_ACT_RELU 2

struct GemmArgs {
    const void* A = nullptr;      // [M, K] (lda) in T, or NHWC activation when conv != 0
    const void* B = nullptr;      // [N, K] (ldb) in T
    void* C = nullptr;            // [M, N] (ldc) f32 or bf16 (out_dtype)
    bf16_t* C2 = nullptr;         // optional bf16 copy of C (ldc2)
    void* Cpre = nullptr;         // optional copy of the pre-activation value (ldpre), f32 or bf16 (pre_dtype)
    const float* bias = nullptr;      // [N]
    const float* rowbias = nullptr;   // [ceil(M / rows_per_group), N] (ldrb)
    const void* res = nullptr;        // [M, N] (ldres) f32 or bf16 (res_dtype)
    const void* gelu_grad_of = nullptr;   // [M, N] (ldgg), f32 or bf16 (gg_dtype): result *= GELU'(gelu_grad_of[m,n])
    // per-head LayerNorm(64) + affine, applied after bias / residual (and after the Cpre copy), before the activation:
    // [N] gain and bias; needs N % 64 == 0, the 16-byte epilogue path and no split-K (else the launcher refuses)
    const float* ln64_g = nullptr;
    const float* ln64_b = nullptr;
    float alpha = 1.0f;               // scales the accumulator before the epilogue terms
    float* colsum_a = nullptr;        // TN only, batch == 1: [M] sums of A's columns over the K contracted rows (bias gradient, unscaled)
    int M = 0, N = 0, K = 0;
    int lda = 0, ldb = 0, ldc = 0, ldc2 = 0, ldpre = 0, ldres = 0, ldrb = 0, ldgg = 0;
    int rows_per_group = 1;
    int act = SQ_ACT_NONE;
    int out_dtype = SQ_F32, res_dtype = SQ_F32, pre_dtype = SQ_F32, gg_dtype = SQ_F32;
    int batch = 1;
    long long sA = 0, sB = 0, sC = 0, sC2 = 0, sPre = 0, sBias = 0, sRb = 0, sRes = 0, sGg = 0;   // per-batch strides (elements)
    // implicit-GEMM convolution view of A: NHWC [n, H, W, Cin]; K = KH*KW*Cin, tap-major
    int conv = 0, H = 0, W = 0, Cin = 0, OH = 0, OW = 0, KW = 1, stride = 1, pad = 0;
    size_t a_bytes = 0, b_bytes = 0;   // extents of A and B for the buffer descriptors (< 2 GiB)
    // SQ_BF16X3 only: A / B / res / C point at the hi plane; the lo plane lies this many ELEMENTS behind it
    long long plA = 0, plB = 0, plRes = 0, plC = 0;
    int x3_f16 = 0;                    // planes are fp16 (SQ_F16X3) instead of bf16
    const float* colscale = nullptr;   // [N] per-column factor on the accumulator (undoes power-of-two pre-scaled weight rows)
    // split-K: set splitk_ws (fp32 scratch) to allow it; the launcher picks the slice count
    float* splitk_ws = nullptr;
    size_t splitk_ws_bytes = 0;
    int splitk = 1;                    // set by the launcher
    // TN products only: up to 4 same-shape problems in ONE launch (grid.z = ngroup); gA / gB / gC / gcs replace A / B / C /
    // colsum_a per member (gcs: all set or all null).  Lets independent weight gradients share a launch so that the grid
    // fills the chip without split-K (4 x 64 tiles of a 1024 x 1024 gradient = 256 tiles)
    int ngroup = 0;
    const void* gA[4] = {nullptr, nullptr, nullptr, nullptr};
    const void* gB[4] = {nullptr, nullptr, nullptr, nullptr};
    void* gC[4] = {nullptr, nullptr, nullptr, nullptr};
    float* gcs[4] = {nullptr, nullptr, nullptr, nullptr};
    int vec_epi = 0;                   // set by sq_launch_gemm: all epilogue operands allow 16-byte accesses
    int b_tiled = 0;                   // split modes: B is K-tile-major -- element (n, k) at ((k / 32) * N + n) * 32 + k % 32 of its plane (K % 32 == 0):
                                       // a K-tile of BN rows is one contiguous run, every LDS-DMA request whole cache lines
    // split modes, dual form (gemm_x3.hip): the identity of the epilogue is not read but computed in the launch as a second
    // product, identity = join(split(colscale2 * (gather(A2) . B2^T) + bias2)) -- a bottleneck's downsample branch
    // (src/resnet.py:87-88): row m = output pixel (img, oh, ow) reads the A2 row of input pixel (img, oh * s, ow * s)
    const void* A2 = nullptr; long long plA2 = 0; size_t a2_bytes = 0; int lda2 = 0;   // NHWC input [n, dH, dW, K2] planes
    const void* B2 = nullptr; size_t b2_bytes = 0; int ldb2 = 0; int K2 = 0;           // [N, K2] planes, plB apart, b_tiled as B
    const float* bias2 = nullptr; const float* colscale2 = nullptr;
    int dH = 0, dW = 0, dOH = 0, dOW = 0, dstride = 1;
    int tile_group_m = 0;              // gemm_p8.hip: tile rows per group of the tile walk (set by its launcher)
    // gemm_p8.hip only (sq_launch_gemm_p8 directly): the ViS combiner in the f projection's epilogue (src/tformer_lin.py:20-25).  With
    // ln64_g / ln64_b and act = GELU set, C receives  GELU( GELU(LN64(A.B^T + bias))_h . comb_w_h[:, 0:64]^T + comb_rb[m / comb_rpg, h] )
    // per head h = 64 columns: the per-head 64 x 64 product runs on the wave's LDS slab, Lf is never written
    const void* comb_w = nullptr;      // bf16 [N / 64][64][128]: row o of head h = the combiner weight row, its first 64 columns apply to Lf
    const float* comb_rb = nullptr;    // f32 [ceil(M / comb_rpg), N] (comb_ldrb): the summary half's product + the combiner bias (Cs)
    int comb_ldrb = 0, comb_rpg = 1;
    int dbg = 0;                       // ablation switches (tools/gemm_probe.py): 1 no stores, 2 no global loads after tile 0, 4 no MFMA
};

// member i of a group's pointer array, i wave-uniform: selects, not an indexed load (indexing a by-value kernel argument
// array with a run-time index makes hipcc copy the whole argument block to scratch: 520 B per lane, the kernel ran 1.5x slower)
template <typename P>
static __device__ __forceinline__ P sq_group_pick(P const (&a)[4], int i) { return i == 0 ? a[0] : i == 1 ? a[1] : i == 2 ? a[2] : a[3]; }

// dtype: SQ_F32 (v_mfma_f32_32x32x2_f32, exact fp32) or SQ_BF16 (v_mfma_f32_32x32x16_bf16)
int sq_launch_gemm(const GemmArgs& a, int dtype, hipStream_t stream);
bool sq_gemm_takes_p8_256(const GemmArgs& a);      // bf16: would sq_launch_gemm run gemm_p8.hip's 256 x 256 kernel (the one with the combiner epilogue)?
// split-bf16 product (gemm_x3.hip): hi/lo bf16 planes, three bf16 MFMAs per product, fp32 accumulation; bias / residual / ReLU only
int sq_launch_gemm_x3(const GemmArgs& a, hipStream_t stream);
// TN product for weight gradients:  C[M,N] = alpha * sum_k A[k, M-index] * B[k, N-index]
//   A [K, M] (lda) and B [K, N] (ldb) are row-major with the CONTRACTION index as the row, i.e. the
//   activations exactly as the forward pass stored them (no transposed copies).  a.M / a.N = extents of C,
//   a.K = number of rows contracted.  Same epilogue / split-K fields as sq_launch_gemm.
int sq_launch_gemm_tn(const GemmArgs& a, int dtype, hipStream_t stream);
int sq_launch_splitk_reduce(const GemmArgs& a, hipStream_t stream);
// experiment knobs (not part of the product ABI): key 0 = forced tile (WTM*10+WTN, 0 = auto), key 1 = dbg flags
extern "C" int sq_dbg_set(int key, int value);
