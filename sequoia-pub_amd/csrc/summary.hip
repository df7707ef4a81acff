// Summary branch of a SummaryMixing layer in ONE launch (bf16 mode), forward:
//   Sm = Xbar Ws^T + bs            (tformer_lin.py:22  s(mean_tokens x);  Xbar [B, D] from token_mean)
//   Ts = GELU(LayerNorm64(Sm))     (:22-23, per head)
//   Cs = Ts_h Wc_h[:, 64:]^T + bc  (the summary half of the combiner, :24-25; added per slide to the local half)
// These are per-SLIDE tensors ([64, 1024] at the bench size): as four launches (GEMM + its K-slice reduce, LN64+GELU,
// batched 64x64 GEMM) they cost four launch latencies on the helper stream and made the main stream wait ~13 us per
// layer.  Here one workgroup owns one head (64 output columns) x 64 slides: K loop over D with register-prefetched
// 64 x 64 tiles, LayerNorm over the staged 64 x 64 tile (one head = one tile row), then the 64 x 64 x 64 product
// straight from LDS.
#include "gemm_epi.h"
#include "vis.h"

namespace {

constexpr int LDT = 72;      // bf16 elements per staged row (64 + 8 pad: 144-byte pitch keeps b128 fragment reads spread)

__device__ __forceinline__ bf16x8 frag(const bf16_t* base, int row, int kstep, int g) {
    return *reinterpret_cast<const bf16x8*>(base + row * LDT + kstep * 16 + g * 8);
}

__global__ __launch_bounds__(256) void summary_fwd_kernel(const bf16_t* __restrict__ Xbar, const bf16_t* __restrict__ Ws,
                                                          const float* __restrict__ bs, const float* __restrict__ lng,
                                                          const float* __restrict__ lnb, const bf16_t* __restrict__ Wc,
                                                          const float* __restrict__ bc, float* __restrict__ Sm,
                                                          bf16_t* __restrict__ Ts, float* __restrict__ Cs, int B, int D, int HD) {
    __shared__ __attribute__((aligned(16))) bf16_t sA[64 * LDT];      // Xbar tile, later Ts
    __shared__ __attribute__((aligned(16))) bf16_t sB[64 * LDT];      // Ws tile, later Wc[:, 64:]
    __shared__ float sT[64][65];                                      // Sm tile for the LayerNorm
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, g = lane >> 5;
    const int h = blockIdx.x, b0 = blockIdx.y * 64;

    // loader: thread -> (row = tid >> 2, 16-element half-chunks 2 x (tid & 3)) : 2 x 16 B per operand per K-tile
    const int lr = tid >> 2, lc = (tid & 3) * 16;
    const bool a_ok = b0 + lr < B;
    const bf16_t* pa = Xbar + (size_t)(b0 + lr) * D + lc;
    const bf16_t* pb = Ws + (size_t)(h * 64 + lr) * D + lc;
    const u32x4 zero = {0, 0, 0, 0};
    u32x4 ra0, ra1, rb0, rb1;
    auto fetch = [&](int k0) {
        ra0 = a_ok ? *reinterpret_cast<const u32x4*>(pa + k0) : zero;
        ra1 = a_ok ? *reinterpret_cast<const u32x4*>(pa + k0 + 8) : zero;
        rb0 = *reinterpret_cast<const u32x4*>(pb + k0);
        rb1 = *reinterpret_cast<const u32x4*>(pb + k0 + 8);
    };
    auto stash = [&]() {
        *reinterpret_cast<u32x4*>(sA + lr * LDT + lc) = ra0;
        *reinterpret_cast<u32x4*>(sA + lr * LDT + lc + 8) = ra1;
        *reinterpret_cast<u32x4*>(sB + lr * LDT + lc) = rb0;
        *reinterpret_cast<u32x4*>(sB + lr * LDT + lc + 8) = rb1;
    };
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    fetch(0);
    for (int k0 = 0; k0 < D; k0 += 64) {
        __syncthreads();                 // previous tile fully consumed
        stash();
        __syncthreads();
        if (k0 + 64 < D) fetch(k0 + 64);     // next tile's loads fly under this tile's MFMAs
#pragma unroll
        for (int s = 0; s < 4; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(sA, wm * 32 + l31, s, g), frag(sB, wn * 32 + l31, s, g), acc, 0, 0, 0);
    }
    // Sm tile (+ bias) -> LDS.  C/D layout: col = l31, row = (r & 3) + 8 * (r >> 2) + 4 * g
    {
        const int col = wn * 32 + l31;
        const float bias = bs[h * 64 + col];
#pragma unroll
        for (int r = 0; r < 16; ++r) sT[wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * g][col] = acc[r] + bias;
    }
    // the second product's weight tile Wc_h[:, 64:128] ([64 out][64 k], row pitch 128 in memory) while the LN runs
    {
        const bf16_t* pw = Wc + (size_t)(h * 64 + lr) * 128 + 64 + lc;
        rb0 = *reinterpret_cast<const u32x4*>(pw);
        rb1 = *reinterpret_cast<const u32x4*>(pw + 8);
    }
    __syncthreads();
    // LayerNorm(64) + GELU per slide row: 4 threads per row, 16 columns each (two-pass statistics, eps 1e-5)
    {
        const int row = tid >> 2, c0 = (tid & 3) * 16;
        float v[16], sm = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) { v[e] = sT[row][c0 + e]; sm += v[e]; }
        sm = group4_sum(sm);
        const float mean = sm * (1.0f / 64.0f);
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) { const float d = v[e] - mean; q += d * d; }
        q = group4_sum(q);
        const float rstd = 1.0f / sqrtf(q * (1.0f / 64.0f) + 1e-5f);
        const bool ok = b0 + row < B;
        uint32_t packed[8];
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            const int c = h * 64 + c0 + e;
            const float y0 = sq_gelu<true>((v[e] - mean) * rstd * lng[c] + lnb[c]);
            const float y1 = sq_gelu<true>((v[e + 1] - mean) * rstd * lng[c + 1] + lnb[c + 1]);
            packed[e >> 1] = pack_bf16x2(y0, y1);
        }
        if (ok) {
            float* dsm = Sm + (size_t)(b0 + row) * HD + h * 64 + c0;
#pragma unroll
            for (int e = 0; e < 16; e += 4) *reinterpret_cast<f32x4*>(dsm + e) = f32x4{v[e], v[e + 1], v[e + 2], v[e + 3]};
            bf16_t* dts = Ts + (size_t)(b0 + row) * HD + h * 64 + c0;
            *reinterpret_cast<u32x4*>(dts) = u32x4{packed[0], packed[1], packed[2], packed[3]};
            *reinterpret_cast<u32x4*>(dts + 8) = u32x4{packed[4], packed[5], packed[6], packed[7]};
        }
        // Ts tile and the weight tile for the second product (all reads of sA / sB by the K loop are behind barriers)
        *reinterpret_cast<u32x4*>(sA + row * LDT + c0) = u32x4{packed[0], packed[1], packed[2], packed[3]};
        *reinterpret_cast<u32x4*>(sA + row * LDT + c0 + 8) = u32x4{packed[4], packed[5], packed[6], packed[7]};
        *reinterpret_cast<u32x4*>(sB + lr * LDT + lc) = rb0;
        *reinterpret_cast<u32x4*>(sB + lr * LDT + lc + 8) = rb1;
    }
    __syncthreads();
    // Cs = Ts Wc[:, 64:]^T + bc
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(sA, wm * 32 + l31, s, g), frag(sB, wn * 32 + l31, s, g), acc, 0, 0, 0);
    {
        const int col = h * 64 + wn * 32 + l31;
        const float bias = bc[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = b0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
            if (row < B) Cs[(size_t)row * HD + col] = acc[r] + bias;
        }
    }
}

}  // namespace

// bf16 only; D % 64 == 0.  Xbar / Ts bf16 [B, *]; Ws, Wc in the bf16 parameter shadow; Sm, Cs f32.
int sq_launch_summary_fwd(const void* Xbar, const void* Ws, const float* bs, const float* lng, const float* lnb, const void* Wc,
                          const float* bc, float* Sm, void* Ts, float* Cs, int B, int D, int H, hipStream_t stream) {
    SQ_REQUIRE(D % 64 == 0 && B >= 1 && H >= 1, "summary_fwd: D=%d must be a multiple of 64", D);
    int prof = -1;
    if (sq_prof_on()) prof = sq_prof_begin("summary_fwd_bf16", 2.0 * B * (double)H * 64 * (D + 64), ((double)B * D + (double)H * 64 * D) * 2.0, stream);
    hipLaunchKernelGGL(summary_fwd_kernel, dim3(H, (B + 63) / 64), dim3(256), 0, stream, (const bf16_t*)Xbar, (const bf16_t*)Ws, bs, lng, lnb,
                       (const bf16_t*)Wc, bc, Sm, (bf16_t*)Ts, Cs, B, D, H * 64);
    SQ_LAUNCH_CHECK();
    if (prof >= 0) sq_prof_end(prof, stream);
    return SQ_OK;
}
