// Backward-pass helper kernels (column reductions, LayerNorm / LN64+GELU gradients).
// All reductions are two-stage with a fixed order, so gradients are run-to-run deterministic.
#include "elementwise.h"

namespace {

constexpr float LN_EPS = 1e-5f;
constexpr int COLSUM_SPLITS = 64;
constexpr int LN_BWD_PARTIALS = 512;     // partial [dg | db] rows a backward kernel leaves: one colsum_small launch finishes them

template <typename T> __device__ __forceinline__ float ld_as_f32(const T* p, size_t i);
template <> __device__ __forceinline__ float ld_as_f32<float>(const float* p, size_t i) { return p[i]; }
template <> __device__ __forceinline__ float ld_as_f32<bf16_t>(const bf16_t* p, size_t i) { return bf16_to_f32(p[i]); }

// stage 1: block (64 columns x 4 row lanes) sums rows [r_lo, r_hi) of its column strip
template <typename T>
__global__ __launch_bounds__(256) void colsum_stage1(const T* __restrict__ x, int R, int C, int ld, int rows_per_split,
                                                     float* __restrict__ partial) {
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    const int r_lo = blockIdx.y * rows_per_split;
    const int r_hi = min(R, r_lo + rows_per_split);
    float acc = 0.f;
    if (c < C)
        for (int r = r_lo + ty; r < r_hi; r += 4) acc += ld_as_f32<T>(x, (size_t)r * ld + c);
    red[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && c < C) partial[(size_t)blockIdx.y * C + c] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}

// stage 2: block = 64 columns x 4 groups of partials (fixed order -> deterministic); columns >= split go to out2
__global__ __launch_bounds__(256) void colsum_stage2(const float* __restrict__ partial, int nsplit, int C, float* __restrict__ out,
                                                     float* __restrict__ out2, int split) {
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    float acc = 0.f;
    if (c < C)
        for (int s = ty; s < nsplit; s += 4) acc += partial[(size_t)s * C + c];
    red[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && c < C) {
        const float v = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
        if (out2 && c >= split) out2[c - split] = v;
        else out[c] = v;
    }
}

// few rows (<= 512 partial rows of a backward kernel): ONE launch; block = 16 columns x 16 row lanes, the 16 lane
// sums combined through LDS in a fixed tree (deterministic); columns >= split go to out2
__global__ __launch_bounds__(256) void colsum_small_kernel(const float* __restrict__ x, int R, int C, int ld, float* __restrict__ out,
                                                           float* __restrict__ out2, int split) {
    __shared__ float red[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + tx;
    float a0 = 0.f, a1 = 0.f;
    if (c < C) {
        int r = ty;
        for (; r + 16 < R; r += 32) { a0 += x[(size_t)r * ld + c]; a1 += x[(size_t)(r + 16) * ld + c]; }
        if (r < R) a0 += x[(size_t)r * ld + c];
    }
    red[ty][tx] = a0 + a1;
    __syncthreads();
    if (ty == 0 && c < C) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k += 4) v += (red[k][tx] + red[k + 1][tx]) + (red[k + 2][tx] + red[k + 3][tx]);
        if (out2 && c >= split) out2[c - split] = v;
        else out[c] = v;
    }
}

// the same for several matrices in one launch (block -> job by block prefix)
__global__ __launch_bounds__(256) void colsum_multi_kernel(const sq_colsum_jobs jobs) {
    __shared__ float red[16][17];
    int j = 0;
    while (j + 1 < jobs.n && (int)blockIdx.x >= jobs.job[j + 1].blk0) ++j;
    const sq_colsum_job& J = jobs.job[j];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = ((int)blockIdx.x - J.blk0) * 16 + tx;
    const float* x = J.x;
    const int R = J.R, C = J.C, ld = J.ld;
    float a0 = 0.f, a1 = 0.f;
    if (c < C) {
        int r = ty;
        for (; r + 16 < R; r += 32) { a0 += x[(size_t)r * ld + c]; a1 += x[(size_t)(r + 16) * ld + c]; }
        if (r < R) a0 += x[(size_t)r * ld + c];
    }
    red[ty][tx] = a0 + a1;
    __syncthreads();
    if (ty == 0 && c < C) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k += 4) v += (red[k][tx] + red[k + 1][tx]) + (red[k + 2][tx] + red[k + 3][tx]);
        if (J.out2 && c >= J.split) J.out2[c - J.split] = v;
        else J.out[c] = v;
    }
}

// out[g, c] = scale * sum_n x[g*N + n, c]: thread = (g, 4-column chunk, quarter of the rows); the four
// quarters are combined through LDS in a fixed order
template <typename T>
__global__ __launch_bounds__(256) void group_sum4_kernel(const T* __restrict__ x, int G, int N, int C4, float scale,
                                                         float* __restrict__ out, T* __restrict__ out_lp) {
    __shared__ float4 red[4][64];
    const int tx = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + tx;            // (g, c4) flat
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < G * C4) {
        const int g = i / C4, c4 = i - g * C4;
        const int per = (N + 3) / 4, lo = q * per, hi = min(N, lo + per);
        for (int n = lo; n < hi; ++n) {
            const size_t e = (((size_t)g * N + n) * C4 + c4) * 4;
            acc.x += ld_as_f32<T>(x, e); acc.y += ld_as_f32<T>(x, e + 1); acc.z += ld_as_f32<T>(x, e + 2); acc.w += ld_as_f32<T>(x, e + 3);
        }
    }
    red[q][tx] = acc;
    __syncthreads();
    if (q == 0 && i < G * C4) {
        float4 r;
        r.x = ((red[0][tx].x + red[1][tx].x) + (red[2][tx].x + red[3][tx].x)) * scale;
        r.y = ((red[0][tx].y + red[1][tx].y) + (red[2][tx].y + red[3][tx].y)) * scale;
        r.z = ((red[0][tx].z + red[1][tx].z) + (red[2][tx].z + red[3][tx].z)) * scale;
        r.w = ((red[0][tx].w + red[1][tx].w) + (red[2][tx].w + red[3][tx].w)) * scale;
        reinterpret_cast<float4*>(out)[i] = r;
        if (out_lp) {
            if constexpr (sizeof(T) == 2) reinterpret_cast<uint2*>(out_lp)[i] = make_uint2(pack_bf16x2(r.x, r.y), pack_bf16x2(r.z, r.w));
        }
    }
}

template <typename T>
__global__ void group_sum_kernel(const T* __restrict__ x, int G, int N, int C, float scale, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G * C) return;
    const int g = i / C, c = i - g * C;
    float acc = 0.f;
    for (int n = 0; n < N; ++n) acc += ld_as_f32<T>(x, ((size_t)g * N + n) * C + c);
    out[i] = acc * scale;
}

__global__ void bcast_rows_kernel(const float4* __restrict__ src, float scale, float4* __restrict__ dst, uint2* __restrict__ dst_lp,
                                  int N, int D4, uint32_t total4) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += gridDim.x * blockDim.x) {
        const uint32_t b = i / ((uint32_t)N * D4);
        const int d = (int)(i % (uint32_t)D4);
        float4 v = src[(size_t)b * D4 + d];
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        if (dst) dst[i] = v;
        if (dst_lp) dst_lp[i] = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
    }
}

__global__ void batch_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int ND) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ND) return;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc += x[(size_t)b * ND + i];
    out[i] = acc;
}

// four consecutive elements of row-major [.., ld] data, fp32 or bf16 (wave-uniform switch): the lean bf16 training stream keeps
// saved activations and the gradient stream in bf16 only
template <bool is_bf16>
__device__ __forceinline__ float4 ld4_any(const void* base, size_t elem_off, int c4) {
    if constexpr (is_bf16) {
        const uint2 u = reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(base) + elem_off)[c4];
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
    }
    return reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + elem_off)[c4];
}

// one wave per row (grid-strided); lane owns float4 columns i*64+lane; per-block dg/db partials go to ws
// LEAN: dy, x and dres are bf16 (the lean training stream); a compile-time switch -- a run-time one puts every load behind its own
// branch and the loads of a row no longer go out together (measured: 22.8 -> 25.4 us although the kernel read fewer bytes)
template <int MAXI, bool LEAN>
__global__ __launch_bounds__(256) void ln_rows_bwd_kernel(const void* __restrict__ dy, const void* __restrict__ x,
                                                          const float* __restrict__ g, const void* __restrict__ dres,
                                                          float* __restrict__ dx, bf16_t* __restrict__ dx_lp,
                                                          float* __restrict__ ws, int R, int D) {
    const int lane = threadIdx.x & 63;
    const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    const int D4 = D >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4 ag[MAXI], ab[MAXI], gg[MAXI];
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        ag[i] = ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int c = i * 64 + lane;
        gg[i] = c < D4 ? g4[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int row = wave_global; row < R; row += nwaves) {
        float4 xv[MAXI], dv[MAXI];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int c = i * 64 + lane;
            xv[i] = c < D4 ? ld4_any<LEAN>(x, (size_t)row * D, c) : make_float4(0.f, 0.f, 0.f, 0.f);
            dv[i] = c < D4 ? ld4_any<LEAN>(dy, (size_t)row * D, c) : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (xv[i].x + xv[i].y) + (xv[i].z + xv[i].w);
        }
        const float mean = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int c = i * 64 + lane;
            if (c < D4) {
                xv[i].x -= mean; xv[i].y -= mean; xv[i].z -= mean; xv[i].w -= mean;
                q += (xv[i].x * xv[i].x + xv[i].y * xv[i].y) + (xv[i].z * xv[i].z + xv[i].w * xv[i].w);
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + LN_EPS);
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            // xhat, dxhat = dy * gamma; accumulate parameter grads
            xv[i].x *= rstd; xv[i].y *= rstd; xv[i].z *= rstd; xv[i].w *= rstd;
            ag[i].x += dv[i].x * xv[i].x; ag[i].y += dv[i].y * xv[i].y; ag[i].z += dv[i].z * xv[i].z; ag[i].w += dv[i].w * xv[i].w;
            ab[i].x += dv[i].x; ab[i].y += dv[i].y; ab[i].z += dv[i].z; ab[i].w += dv[i].w;
            dv[i].x *= gg[i].x; dv[i].y *= gg[i].y; dv[i].z *= gg[i].z; dv[i].w *= gg[i].w;
            c1 += (dv[i].x + dv[i].y) + (dv[i].z + dv[i].w);
            c2 += (dv[i].x * xv[i].x + dv[i].y * xv[i].y) + (dv[i].z * xv[i].z + dv[i].w * xv[i].w);
        }
        c1 = wave_sum(c1) / (float)D;
        c2 = wave_sum(c2) / (float)D;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int c = i * 64 + lane;
            if (c < D4) {
                float4 o;
                o.x = rstd * (dv[i].x - c1 - xv[i].x * c2);
                o.y = rstd * (dv[i].y - c1 - xv[i].y * c2);
                o.z = rstd * (dv[i].z - c1 - xv[i].z * c2);
                o.w = rstd * (dv[i].w - c1 - xv[i].w * c2);
                if (dres) {
                    const float4 r4 = ld4_any<LEAN>(dres, (size_t)row * D, c);
                    o.x += r4.x; o.y += r4.y; o.z += r4.z; o.w += r4.w;
                }
                if (dx) reinterpret_cast<float4*>(dx + (size_t)row * D)[c] = o;
                if (dx_lp) reinterpret_cast<uint2*>(dx_lp + (size_t)row * D)[c] = make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
            }
        }
    }
    // one partial row [dg | db] per BLOCK: waves 1..3 hand their sums to wave 0 through LDS, added in wave order
    __shared__ float4 red[3][2][64];
    const int wave = threadIdx.x >> 6;
    float4* wg = reinterpret_cast<float4*>(ws + (size_t)blockIdx.x * 2 * D);
    float4* wb = reinterpret_cast<float4*>(ws + (size_t)blockIdx.x * 2 * D + D);
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        if (wave > 0) { red[wave - 1][0][lane] = ag[i]; red[wave - 1][1][lane] = ab[i]; }
        __syncthreads();
        const int c = i * 64 + lane;
        if (wave == 0 && c < D4) {
            float4 a = ag[i], b = ab[i];
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                const float4 ra = red[w][0][lane], rb = red[w][1][lane];
                a.x += ra.x; a.y += ra.y; a.z += ra.z; a.w += ra.w;
                b.x += rb.x; b.y += rb.y; b.z += rb.z; b.w += rb.w;
            }
            wg[c] = a; wb[c] = b;
        }
        __syncthreads();
    }
}

// thread owns float4 column(s) (fixed), walks rows; 16 lanes = one 64-wide group
template <int NCH, bool FAST, bool LEAN>
__global__ __launch_bounds__(256) void ln64_gelu_bwd_kernel(const void* __restrict__ dy, const void* __restrict__ x,
                                                            const float* __restrict__ g, const float* __restrict__ b,
                                                            void* __restrict__ dx, int out_bf16, float* __restrict__ ws,
                                                            int R, int C) {
    const int C16 = C >> 2;
    const int cols_per_iter = NCH > 1 ? 256 : min(C16, 256);
    const int rpi = 256 / cols_per_iter;                 // rows per block iteration
    const int rsub = threadIdx.x / cols_per_iter;
    const int col0 = threadIdx.x % cols_per_iter;
    float4 ag[NCH], ab[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) ag[k] = ab[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int row = blockIdx.x * rpi + rsub; row < R; row += gridDim.x * rpi) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c4 = k * 256 + col0;
            const float4 v = ld4_any<LEAN>(x, (size_t)row * C, c4);
            const float4 d = ld4_any<LEAN>(dy, (size_t)row * C, c4);
            const float4 gg = reinterpret_cast<const float4*>(g)[c4], bb = reinterpret_cast<const float4*>(b)[c4];
            float s = (v.x + v.y) + (v.z + v.w);
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            const float mean = s * (1.0f / 64.0f);
            float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
            float q = (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
            const float rstd = 1.0f / sqrtf(q * (1.0f / 64.0f) + LN_EPS);
            a0 *= rstd; a1 *= rstd; a2 *= rstd; a3 *= rstd;                       // xhat
            const float z0 = d.x * sq_gelu_grad<FAST>(a0 * gg.x + bb.x), z1 = d.y * sq_gelu_grad<FAST>(a1 * gg.y + bb.y);
            const float z2 = d.z * sq_gelu_grad<FAST>(a2 * gg.z + bb.z), z3 = d.w * sq_gelu_grad<FAST>(a3 * gg.w + bb.w);
            ag[k].x += z0 * a0; ag[k].y += z1 * a1; ag[k].z += z2 * a2; ag[k].w += z3 * a3;
            ab[k].x += z0; ab[k].y += z1; ab[k].z += z2; ab[k].w += z3;
            const float h0 = z0 * gg.x, h1 = z1 * gg.y, h2 = z2 * gg.z, h3 = z3 * gg.w;   // d xhat
            float c1 = (h0 + h1) + (h2 + h3);
            float c2 = (h0 * a0 + h1 * a1) + (h2 * a2 + h3 * a3);
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) { c1 += __shfl_xor(c1, o, 64); c2 += __shfl_xor(c2, o, 64); }
            c1 *= (1.0f / 64.0f); c2 *= (1.0f / 64.0f);
            float4 o4;
            o4.x = rstd * (h0 - c1 - a0 * c2); o4.y = rstd * (h1 - c1 - a1 * c2);
            o4.z = rstd * (h2 - c1 - a2 * c2); o4.w = rstd * (h3 - c1 - a3 * c2);
            if (out_bf16) reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(dx) + (size_t)row * C)[c4] =
                              make_uint2(pack_bf16x2(o4.x, o4.y), pack_bf16x2(o4.z, o4.w));
            else reinterpret_cast<float4*>(reinterpret_cast<float*>(dx) + (size_t)row * C)[c4] = o4;
        }
    }
    const size_t prow = (size_t)blockIdx.x * rpi + rsub;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int c4 = k * 256 + col0;
        reinterpret_cast<float4*>(ws + prow * 2 * C)[c4] = ag[k];
        reinterpret_cast<float4*>(ws + prow * 2 * C + C)[c4] = ab[k];
    }
}


// ---- lean (bf16 stream) forms of the two LayerNorm backward kernels: 16-byte accesses (8 columns per lane), 512-thread blocks and ONE
// partial [dg | db] row per block, so that 512 partial rows no longer mean 8 waves per CU.  The generic kernels above, fed bf16 rows
// through 8-byte loads, got SLOWER than on fp32 rows (22.8 -> 28.2 us, 13.0 -> 21.4 us): with <= 512 blocks of 4 waves they are bound
// by the latency of one row at a time per wave, not by bytes.
__device__ __forceinline__ void unpack8(const u32x4& u, float (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(u[e] << 16); v[2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u); }
}
__device__ __forceinline__ u32x4 pack8(const float (&v)[8]) {
    return u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
}

// one wave per row; lane owns the 8-column chunks i * 64 + lane
template <int MAXC>
__global__ __launch_bounds__(512) void ln_rows_bwd_lean_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const float* __restrict__ g,
                                                               const bf16_t* __restrict__ dres, float* __restrict__ dx, bf16_t* __restrict__ dx_lp,
                                                               float* __restrict__ ws, int R, int D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int D8 = D >> 3;
    float ag[MAXC][8], ab[MAXC][8], gg[MAXC][8];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = i * 64 + lane;
#pragma unroll
        for (int e = 0; e < 8; ++e) { ag[i][e] = 0.f; ab[i][e] = 0.f; gg[i][e] = c < D8 ? g[c * 8 + e] : 0.f; }
    }
    // (requesting the next row's pieces before this row is worked on was tried: 20.9 -> 24.7 us)
    for (int row = blockIdx.x * 8 + wave; row < R; row += gridDim.x * 8) {
        const size_t ro = (size_t)row * D;
        u32x4 xr[MAXC], dr[MAXC], rr[MAXC];
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = i * 64 + lane;
            const bool ok = c < D8;
            xr[i] = ok ? *reinterpret_cast<const u32x4*>(x + ro + c * 8) : u32x4{0, 0, 0, 0};
            dr[i] = ok ? *reinterpret_cast<const u32x4*>(dy + ro + c * 8) : u32x4{0, 0, 0, 0};
            rr[i] = (ok && dres) ? *reinterpret_cast<const u32x4*>(dres + ro + c * 8) : u32x4{0, 0, 0, 0};
        }
        float xv[MAXC][8], dv[MAXC][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            unpack8(xr[i], xv[i]); unpack8(dr[i], dv[i]);
#pragma unroll
            for (int e = 0; e < 8; e += 2) s += xv[i][e] + xv[i][e + 1];
        }
        const float mean = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            if (i * 64 + lane < D8) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { xv[i][e] -= mean; q += xv[i][e] * xv[i][e]; }
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + LN_EPS);
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xv[i][e] *= rstd;                                        // xhat
                ag[i][e] += dv[i][e] * xv[i][e];
                ab[i][e] += dv[i][e];
                dv[i][e] *= gg[i][e];                                    // d xhat
                c1 += dv[i][e];
                c2 += dv[i][e] * xv[i][e];
            }
        c1 = wave_sum(c1) / (float)D;
        c2 = wave_sum(c2) / (float)D;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = i * 64 + lane;
            if (c < D8) {
                float o[8], r8[8];
                unpack8(rr[i], r8);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rstd * (dv[i][e] - c1 - xv[i][e] * c2) + r8[e];
                if (dx) {
                    reinterpret_cast<float4*>(dx + ro + c * 8)[0] = make_float4(o[0], o[1], o[2], o[3]);
                    reinterpret_cast<float4*>(dx + ro + c * 8)[1] = make_float4(o[4], o[5], o[6], o[7]);
                }
                if (dx_lp) *reinterpret_cast<u32x4*>(dx_lp + ro + c * 8) = pack8(o);
            }
        }
    }
    // one partial row [dg | db] per block: waves 1..7 hand their sums to wave 0 through LDS, added in wave order
    __shared__ float red[7][2][512];
    float* wg = ws + (size_t)blockIdx.x * 2 * D;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        if (wave > 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { red[wave - 1][0][lane * 8 + e] = ag[i][e]; red[wave - 1][1][lane * 8 + e] = ab[i][e]; }
        }
        __syncthreads();
        const int c = i * 64 + lane;
        if (wave == 0 && c < D8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float a = ag[i][e], b = ab[i][e];
#pragma unroll
                for (int w = 0; w < 7; ++w) { a += red[w][0][lane * 8 + e]; b += red[w][1][lane * 8 + e]; }
                wg[c * 8 + e] = a; wg[D + c * 8 + e] = b;
            }
        }
        __syncthreads();
    }
}

// thread owns 8 fixed columns and walks rows; 8 lanes = one 64-wide group; 512 / (C / 8) rows per block iteration
template <bool FAST>
__global__ __launch_bounds__(512) void ln64_gelu_bwd_lean_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const float* __restrict__ g,
                                                                 const float* __restrict__ b, void* __restrict__ dx, int out_bf16,
                                                                 float* __restrict__ ws, int R, int C) {
    const int tpr = C >> 3, rpi = 512 / tpr;
    const int rsub = threadIdx.x / tpr, col = (threadIdx.x % tpr) * 8;
    float ag[8], ab[8], gg[8], bb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ag[e] = 0.f; ab[e] = 0.f; gg[e] = g[col + e]; bb[e] = b[col + e]; }
    for (int row = blockIdx.x * rpi + rsub; row < R; row += gridDim.x * rpi) {
        const size_t ro = (size_t)row * C + col;
        float v[8], d[8];
        unpack8(*reinterpret_cast<const u32x4*>(x + ro), v);
        unpack8(*reinterpret_cast<const u32x4*>(dy + ro), d);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e += 2) s += v[e] + v[e + 1];
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mean = s * (1.0f / 64.0f);
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[e] -= mean; q += v[e] * v[e]; }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        const float rstd = 1.0f / sqrtf(q * (1.0f / 64.0f) + LN_EPS);
        float h[8], c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[e] *= rstd;                                                // xhat
            const float z = d[e] * sq_gelu_grad<FAST>(v[e] * gg[e] + bb[e]);
            ag[e] += z * v[e]; ab[e] += z;
            h[e] = z * gg[e];                                            // d xhat
            c1 += h[e]; c2 += h[e] * v[e];
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) { c1 += __shfl_xor(c1, o, 64); c2 += __shfl_xor(c2, o, 64); }
        c1 *= (1.0f / 64.0f); c2 *= (1.0f / 64.0f);
        float o8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o8[e] = rstd * (h[e] - c1 - v[e] * c2);
        if (out_bf16) *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(dx) + ro) = pack8(o8);
        else {
            reinterpret_cast<float4*>(reinterpret_cast<float*>(dx) + ro)[0] = make_float4(o8[0], o8[1], o8[2], o8[3]);
            reinterpret_cast<float4*>(reinterpret_cast<float*>(dx) + ro)[1] = make_float4(o8[4], o8[5], o8[6], o8[7]);
        }
    }
    // one partial row per block: row slots 1.. hand their sums to slot 0 through LDS, added in slot order
    extern __shared__ float red64[];              // [(rpi - 1)][2][C]
    if (rsub > 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { red64[((size_t)(rsub - 1) * 2) * C + col + e] = ag[e]; red64[((size_t)(rsub - 1) * 2 + 1) * C + col + e] = ab[e]; }
    }
    __syncthreads();
    if (rsub == 0) {
        float* wg = ws + (size_t)blockIdx.x * 2 * C;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = ag[e], bsum = ab[e];
            for (int w = 0; w < rpi - 1; ++w) { a += red64[((size_t)w * 2) * C + col + e]; bsum += red64[((size_t)w * 2 + 1) * C + col + e]; }
            wg[col + e] = a; wg[C + col + e] = bsum;
        }
    }
}

int colsum_impl(const void* x, int dtype, int R, int C, int ld, float* ws, float* out, hipStream_t s, float* out2 = nullptr,
                int split = 0) {
    if (dtype == SQ_F32 && R <= 512) {
        hipLaunchKernelGGL(colsum_small_kernel, dim3((C + 15) / 16), dim3(256), 0, s, (const float*)x, R, C, ld, out, out2, split);
        SQ_LAUNCH_CHECK();
        return SQ_OK;
    }
    int nsplit = (R + 63) / 64;
    if (nsplit > COLSUM_SPLITS) nsplit = COLSUM_SPLITS;
    if (nsplit < 1) nsplit = 1;
    const int rows_per_split = (R + nsplit - 1) / nsplit;
    const dim3 grid((C + 63) / 64, nsplit), block(256);
    if (dtype == SQ_BF16) hipLaunchKernelGGL(colsum_stage1<bf16_t>, grid, block, 0, s, (const bf16_t*)x, R, C, ld, rows_per_split, ws);
    else hipLaunchKernelGGL(colsum_stage1<float>, grid, block, 0, s, (const float*)x, R, C, ld, rows_per_split, ws);
    SQ_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_stage2, dim3((C + 63) / 64), dim3(256), 0, s, ws, nsplit, C, out, out2, split);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}

}  // namespace

size_t sq_colsum_ws_floats(int C) { return (size_t)COLSUM_SPLITS * C; }

int sq_colsum_jobs_add(sq_colsum_jobs* jobs, const float* x, int R, int C, int ld, float* out, float* out2, int split) {
    SQ_REQUIRE(jobs->n < SQ_MAX_COLSUM_JOBS && R > 0 && R <= 512 && C > 0 && ld >= C, "colsum_multi: job %d R=%d C=%d", jobs->n, R, C);
    sq_colsum_job& J = jobs->job[jobs->n++];
    J.x = x; J.out = out; J.out2 = out2; J.R = R; J.C = C; J.ld = ld; J.split = split; J.blk0 = jobs->blocks;
    jobs->blocks += (C + 15) / 16;
    return SQ_OK;
}

int sq_k_colsum_multi(const sq_colsum_jobs& jobs, hipStream_t s) {
    if (jobs.n == 0) return SQ_OK;
    hipLaunchKernelGGL(colsum_multi_kernel, dim3(jobs.blocks), dim3(256), 0, s, jobs);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}

int sq_k_colsum(const void* x, int dtype, int R, int C, int ld, float* ws, float* out, hipStream_t s) {
    SQ_REQUIRE(R > 0 && C > 0 && ld >= C, "colsum: R=%d C=%d ld=%d", R, C, ld);
    return colsum_impl(x, dtype, R, C, ld, ws, out, s);
}

int sq_k_group_sum(const void* x, int dtype, int G, int N, int C, float scale, float* out, hipStream_t s) {
    if (C % 4 == 0) {
        const int items = G * (C / 4);
        if (dtype == SQ_BF16) hipLaunchKernelGGL(group_sum4_kernel<bf16_t>, dim3((items + 63) / 64), dim3(256), 0, s, (const bf16_t*)x, G, N, C / 4, scale, out, (bf16_t*)nullptr);
        else hipLaunchKernelGGL(group_sum4_kernel<float>, dim3((items + 63) / 64), dim3(256), 0, s, (const float*)x, G, N, C / 4, scale, out, (float*)nullptr);
        SQ_LAUNCH_CHECK();
        return SQ_OK;
    }
    const int total = G * C;
    if (dtype == SQ_BF16) hipLaunchKernelGGL(group_sum_kernel<bf16_t>, dim3((total + 255) / 256), dim3(256), 0, s, (const bf16_t*)x, G, N, C, scale, out);
    else hipLaunchKernelGGL(group_sum_kernel<float>, dim3((total + 255) / 256), dim3(256), 0, s, (const float*)x, G, N, C, scale, out);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}

int sq_k_bcast_rows(const float* src, float scale, float* dst, bf16_t* dst_lp, int B, int N, int D, hipStream_t s) {
    SQ_REQUIRE(D % 4 == 0, "bcast_rows: D=%d", D);
    SQ_REQUIRE((size_t)B * N * D / 4 < (1ull << 31), "bcast_rows: tensor too large for 32-bit indexing");
    const uint32_t total4 = (uint32_t)((size_t)B * N * D / 4);
    size_t g = ((size_t)total4 + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(bcast_rows_kernel, dim3((int)g), dim3(256), 0, s, (const float4*)src, scale, (float4*)dst, (uint2*)dst_lp, N, D / 4, total4);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}

int sq_k_batch_sum(const float* x, float* out, int B, int ND, hipStream_t s) {
    hipLaunchKernelGGL(batch_sum_kernel, dim3((ND + 255) / 256), dim3(256), 0, s, x, out, B, ND);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}

size_t sq_ln_bwd_ws_floats(int D) { return (size_t)LN_BWD_PARTIALS * 2 * D + sq_colsum_ws_floats(2 * D); }

int sq_k_ln_rows_bwd(const float* dy, const float* x, const float* g, const float* dres, float* dx, bf16_t* dx_lp, float* dg,
                     float* db, float* ws, int R, int D, hipStream_t s, sq_colsum_jobs* defer) {
    return sq_k_ln_rows_bwd_any(dy, SQ_F32, x, SQ_F32, g, dres, SQ_F32, dx, dx_lp, dg, db, ws, R, D, s, defer);
}

int sq_k_ln_rows_bwd_any(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* g, const void* dres, int dres_dtype,
                         float* dx, bf16_t* dx_lp, float* dg, float* db, float* ws, int R, int D, hipStream_t s, sq_colsum_jobs* defer) {
    SQ_REQUIRE(D % 4 == 0 && D <= 4096 && D > 0, "ln_rows_bwd: D=%d", D);
    SQ_REQUIRE(dx || dx_lp, "ln_rows_bwd: no output");
    const bool lean = dy_dtype == SQ_BF16;
    SQ_REQUIRE((x_dtype == SQ_BF16) == lean && (!dres || (dres_dtype == SQ_BF16) == lean), "ln_rows_bwd: dy, x and dres must share one dtype (all fp32 or all bf16)");
    if (lean && D % 8 == 0) {
        int nb = (R + 7) / 8;
        if (nb > LN_BWD_PARTIALS) nb = LN_BWD_PARTIALS;
        const bf16_t* dyb = (const bf16_t*)dy; const bf16_t* xb = (const bf16_t*)x; const bf16_t* rb = (const bf16_t*)dres;
        if (D <= 1024) hipLaunchKernelGGL(ln_rows_bwd_lean_kernel<2>, dim3(nb), dim3(512), 0, s, dyb, xb, g, rb, dx, dx_lp, ws, R, D);
        else if (D <= 2048) hipLaunchKernelGGL(ln_rows_bwd_lean_kernel<4>, dim3(nb), dim3(512), 0, s, dyb, xb, g, rb, dx, dx_lp, ws, R, D);
        else hipLaunchKernelGGL(ln_rows_bwd_lean_kernel<8>, dim3(nb), dim3(512), 0, s, dyb, xb, g, rb, dx, dx_lp, ws, R, D);
        SQ_LAUNCH_CHECK();
        float* cs = ws + (size_t)LN_BWD_PARTIALS * 2 * D;
        if (defer) return sq_colsum_jobs_add(defer, ws, nb, 2 * D, 2 * D, dg, db, D);
        return colsum_impl(ws, SQ_F32, nb, 2 * D, 2 * D, cs, dg, s, db, D);
    }
    int nblk = (R + 3) / 4;
    if (nblk > LN_BWD_PARTIALS) nblk = LN_BWD_PARTIALS;
    const dim3 grid(nblk), block(256);
#define SQ_LNB(MAXI)                                                                                                                \
    do {                                                                                                                            \
        if (lean) hipLaunchKernelGGL((ln_rows_bwd_kernel<MAXI, true>), grid, block, 0, s, dy, x, g, dres, dx, dx_lp, ws, R, D);      \
        else hipLaunchKernelGGL((ln_rows_bwd_kernel<MAXI, false>), grid, block, 0, s, dy, x, g, dres, dx, dx_lp, ws, R, D);         \
    } while (0)
    if (D <= 1024) SQ_LNB(4); else if (D <= 2048) SQ_LNB(8); else SQ_LNB(16);
#undef SQ_LNB
    SQ_LAUNCH_CHECK();
    // partial rows are [dg | db] of length 2D: one column-sum over nblk*4 partial rows, then split
    float* cs_ws = ws + (size_t)LN_BWD_PARTIALS * 2 * D;
    if (defer) return sq_colsum_jobs_add(defer, ws, nblk, 2 * D, 2 * D, dg, db, D);
    return colsum_impl(ws, SQ_F32, nblk, 2 * D, 2 * D, cs_ws, dg, s, db, D);
}

int sq_k_ln64_gelu_bwd(const float* dy, const float* x, const float* g, const float* b, void* dx, int out_dtype, float* dg,
                       float* db, float* ws, int R, int C, hipStream_t s, sq_colsum_jobs* defer) {
    return sq_k_ln64_gelu_bwd_any(dy, SQ_F32, x, SQ_F32, g, b, dx, out_dtype, dg, db, ws, R, C, s, defer);
}

int sq_k_ln64_gelu_bwd_any(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* g, const float* b, void* dx, int out_dtype,
                           float* dg, float* db, float* ws, int R, int C, hipStream_t s, sq_colsum_jobs* defer) {
    const bool lean = dy_dtype == SQ_BF16;
    SQ_REQUIRE((x_dtype == SQ_BF16) == lean, "ln64_gelu_bwd: dy and x must share one dtype");
    const int C16 = C / 4;
    SQ_REQUIRE(C % 64 == 0 && ((C16 <= 256 && 256 % C16 == 0) || C16 == 512 || C16 == 1024),
               "ln64_gelu_bwd: C=%d (nheads must be a power of two <= 64 for the training path)", C);
    const int ob = out_dtype == SQ_BF16;
    if (lean && C >= 64 && C <= 4096 && 512 % (C / 8) == 0) {
        const int rpi_l = 512 / (C / 8);
        int nb = (R + rpi_l - 1) / rpi_l;
        if (nb > LN_BWD_PARTIALS) nb = LN_BWD_PARTIALS;
        const size_t sh = (size_t)(rpi_l - 1) * 2 * C * 4;
        if (ob) hipLaunchKernelGGL(ln64_gelu_bwd_lean_kernel<true>, dim3(nb), dim3(512), sh, s, (const bf16_t*)dy, (const bf16_t*)x, g, b, dx, ob, ws, R, C);
        else hipLaunchKernelGGL(ln64_gelu_bwd_lean_kernel<false>, dim3(nb), dim3(512), sh, s, (const bf16_t*)dy, (const bf16_t*)x, g, b, dx, ob, ws, R, C);
        SQ_LAUNCH_CHECK();
        float* cs = ws + (size_t)LN_BWD_PARTIALS * 2 * C;
        if (defer) return sq_colsum_jobs_add(defer, ws, nb, 2 * C, 2 * C, dg, db, C);
        return colsum_impl(ws, SQ_F32, nb, 2 * C, 2 * C, cs, dg, s, db, C);
    }
    const int nch = C16 <= 256 ? 1 : C16 / 256;
    const int rpi = C16 <= 256 ? 256 / C16 : 1;
    int nblk = (R + rpi - 1) / rpi;
    if (nblk * rpi > LN_BWD_PARTIALS) nblk = LN_BWD_PARTIALS / rpi;
    const dim3 grid(nblk), block(256);
    // bf16 gradients out: the 7-term erf (consistent with the forward kernel); fp32: erff
#define SQ_L64(NCH, FAST)                                                                                                            \
    do {                                                                                                                            \
        if (lean) hipLaunchKernelGGL((ln64_gelu_bwd_kernel<NCH, FAST, true>), grid, block, 0, s, dy, x, g, b, dx, ob, ws, R, C);     \
        else hipLaunchKernelGGL((ln64_gelu_bwd_kernel<NCH, FAST, false>), grid, block, 0, s, dy, x, g, b, dx, ob, ws, R, C);        \
    } while (0)
    if (ob) { if (nch == 1) SQ_L64(1, true); else if (nch == 2) SQ_L64(2, true); else SQ_L64(4, true); }
    else { if (nch == 1) SQ_L64(1, false); else if (nch == 2) SQ_L64(2, false); else SQ_L64(4, false); }
#undef SQ_L64
    SQ_LAUNCH_CHECK();
    float* cs_ws = ws + (size_t)LN_BWD_PARTIALS * 2 * C;
    if (defer && nblk * rpi <= 512) return sq_colsum_jobs_add(defer, ws, nblk * rpi, 2 * C, 2 * C, dg, db, C);
    return colsum_impl(ws, SQ_F32, nblk * rpi, 2 * C, 2 * C, cs_ws, dg, s, db, C);
}
