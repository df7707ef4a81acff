// Memory-bound helper kernels (gfx950): 16-byte accesses per lane, wave64 reductions.
#include "elementwise.h"

namespace {

constexpr float LN_EPS = 1e-5f;

__global__ void add_pos_kernel(const float4* __restrict__ x, const float4* __restrict__ pos, float4* __restrict__ X,
                               uint2* __restrict__ Xh, uint32_t total4, uint32_t nd4) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += gridDim.x * blockDim.x) {
        const float4 a = x[i], p = pos[i % nd4];
        const float4 v = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
        if (X) X[i] = v;                 // (null: the bf16 stream of inference -- nobody reads the fp32 rows)
        if (Xh) Xh[i] = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
    }
}

// X[b, t, :] = src[idx[b, t], :] + pos[t, :]  (idx < 0: a zero row -- window padding, spatial_vis/visualize.py:72-75):
// the [B, N, D] input of a window batch is never materialised, rows come straight from the tile-feature cache
__global__ void add_pos_gather_kernel(const float4* __restrict__ src, const int32_t* __restrict__ idx, const float4* __restrict__ pos,
                                      float4* __restrict__ X, uint2* __restrict__ Xh, uint32_t total4, uint32_t nd4, uint32_t d4) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += gridDim.x * blockDim.x) {
        const uint32_t row = i / d4, col = i - row * d4;
        const int32_t r = idx[row];
        const float4 p = pos[i % nd4];
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r >= 0) a = src[(size_t)r * d4 + col];
        const float4 v = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
        if (X) X[i] = v;                 // (null: the bf16 stream of inference -- nobody reads the fp32 rows)
        if (Xh) Xh[i] = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
    }
}

// four consecutive elements of a row as floats: fp32 storage (16 bytes) or bf16 storage (8 bytes)
template <bool IN_BF16>
__device__ __forceinline__ float4 load4(const void* base, size_t i4) {
    if constexpr (IN_BF16) {
        const uint2 t = reinterpret_cast<const uint2*>(base)[i4];
        return make_float4(__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16), __uint_as_float(t.y & 0xffff0000u));
    } else {
        return reinterpret_cast<const float4*>(base)[i4];
    }
}

// thread = (b, d4, quarter of the tokens); quarters combined through LDS in a fixed order (deterministic)
template <bool IN_BF16>
__global__ __launch_bounds__(256) void token_mean_kernel(const void* __restrict__ X, float4* __restrict__ out, uint2* __restrict__ outh,
                                                         int B, int N, int D4) {
    __shared__ float4 red[4][64];
    const int tx = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + tx;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < B * D4) {
        const int b = i / D4, d = i - b * D4;
        const size_t p = (size_t)b * N * D4 + d;
        const int per = (N + 3) / 4, lo = q * per, hi = min(N, lo + per);
        for (int n = lo; n < hi; ++n) {
            const float4 v = load4<IN_BF16>(X, p + (size_t)n * D4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    red[q][tx] = acc;
    __syncthreads();
    if (q == 0 && i < B * D4) {
        const float inv = 1.0f / (float)N;
        float4 r;
        r.x = ((red[0][tx].x + red[1][tx].x) + (red[2][tx].x + red[3][tx].x)) * inv;
        r.y = ((red[0][tx].y + red[1][tx].y) + (red[2][tx].y + red[3][tx].y)) * inv;
        r.z = ((red[0][tx].z + red[1][tx].z) + (red[2][tx].z + red[3][tx].z)) * inv;
        r.w = ((red[0][tx].w + red[1][tx].w) + (red[2][tx].w + red[3][tx].w)) * inv;
        out[i] = r;
        if (outh) outh[i] = make_uint2(pack_bf16x2(r.x, r.y), pack_bf16x2(r.z, r.w));
    }
}

// bf16 rows (the residual stream of bf16 inference), 16-byte loads: thread = (b, 8 columns, quarter of the tokens), five
// independent loads in flight per thread.  Per column the additions happen in exactly the order of token_mean_kernel (a quarter's
// tokens in sequence, the quarters as (q0 + q1) + (q2 + q3)): same bits.  The 8-byte / one-load-at-a-time form above ran
// the spatial path's [102400, 1024] reduction at 1.2 TB/s (181 us, 11 % of a layer: rocprofv3, round 4).
__global__ __launch_bounds__(256) void token_mean_bf16x8_kernel(const u32x4* __restrict__ X, float4* __restrict__ out, uint2* __restrict__ outh,
                                                                int B, int N, int D8) {
    __shared__ float red[4][64][8];
    const int tx = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + tx;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (i < B * D8) {
        const int b = i / D8, d = i - b * D8;
        const u32x4* p = X + (size_t)b * N * D8 + d;
        const int per = (N + 3) / 4, lo = q * per, hi = min(N, lo + per);
        int n = lo;
        for (; n + 5 <= hi; n += 5) {
            u32x4 t[5];
#pragma unroll
            for (int u = 0; u < 5; ++u) t[u] = p[(size_t)(n + u) * D8];
#pragma unroll
            for (int u = 0; u < 5; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[2 * e] += __uint_as_float(t[u][e] << 16); acc[2 * e + 1] += __uint_as_float(t[u][e] & 0xffff0000u); }
        }
        for (; n < hi; ++n) {
            const u32x4 t = p[(size_t)n * D8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[2 * e] += __uint_as_float(t[e] << 16); acc[2 * e + 1] += __uint_as_float(t[e] & 0xffff0000u); }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[q][tx][e] = acc[e];
    __syncthreads();
    if (q == 0 && i < B * D8) {
        const float inv = 1.0f / (float)N;
        float r[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = ((red[0][tx][e] + red[1][tx][e]) + (red[2][tx][e] + red[3][tx][e])) * inv;
        out[2 * i] = make_float4(r[0], r[1], r[2], r[3]);
        out[2 * i + 1] = make_float4(r[4], r[5], r[6], r[7]);
        if (outh) {
            outh[2 * i] = make_uint2(pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3]));
            outh[2 * i + 1] = make_uint2(pack_bf16x2(r[4], r[5]), pack_bf16x2(r[6], r[7]));
        }
    }
}

// LayerNorm of bf16 rows with D = 512 NI, bf16 or fp32 out: a lane owns 8 consecutive columns per 512-column block (one 16-byte
// load, one 16-byte bf16 store).  Statistics two-pass in fp32 like ln_rows_kernel; the order of the partial sums differs from
// it (8 columns per lane instead of 4), i.e. results agree to fp32 rounding, not bitwise -- used for the bf16 stream only.
template <int NI>
__global__ __launch_bounds__(256) void ln_rows_bf16x8_kernel(const u32x4* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                             void* __restrict__ y, int out_bf16, int R, float* __restrict__ mean_out,
                                                             float* __restrict__ rstd_out) {
    constexpr int D = 512 * NI;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    float v[NI][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const u32x4 t = x[(size_t)row * (D / 8) + i * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[i][2 * e] = __uint_as_float(t[e] << 16); v[i][2 * e + 1] = __uint_as_float(t[e] & 0xffff0000u); }
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[i][e];
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float a = v[i][e] - mean; q += a * a; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + LN_EPS);
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = i * 512 + lane * 8;
        const float4 g0 = *reinterpret_cast<const float4*>(g + c), g1 = *reinterpret_cast<const float4*>(g + c + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(b + c), b1 = *reinterpret_cast<const float4*>(b + c + 4);
        float o[8];
        o[0] = (v[i][0] - mean) * rstd * g0.x + b0.x; o[1] = (v[i][1] - mean) * rstd * g0.y + b0.y;
        o[2] = (v[i][2] - mean) * rstd * g0.z + b0.z; o[3] = (v[i][3] - mean) * rstd * g0.w + b0.w;
        o[4] = (v[i][4] - mean) * rstd * g1.x + b1.x; o[5] = (v[i][5] - mean) * rstd * g1.y + b1.y;
        o[6] = (v[i][6] - mean) * rstd * g1.z + b1.z; o[7] = (v[i][7] - mean) * rstd * g1.w + b1.w;
        if (out_bf16) {
            reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(y) + (size_t)row * D)[i * 64 + lane] =
                u32x4{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
        } else {
            float* d = reinterpret_cast<float*>(y) + (size_t)row * D + c;
            *reinterpret_cast<float4*>(d) = make_float4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<float4*>(d + 4) = make_float4(o[4], o[5], o[6], o[7]);
        }
    }
}

// one wave per row; row kept in registers between the mean and variance passes
template <int MAXI, bool IN_BF16>
__global__ __launch_bounds__(256) void ln_rows_kernel(const void* __restrict__ x, const float* __restrict__ g,
                                                      const float* __restrict__ b, void* __restrict__ y, int out_bf16,
                                                      int R, int D, float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const int D4 = D >> 2;
    float4 v[MAXI];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int c = i * 64 + lane;
        v[i] = c < D4 ? load4<IN_BF16>(x, (size_t)row * D4 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int c = i * 64 + lane;
        if (c < D4) {
            const float a = v[i].x - mean, bb = v[i].y - mean, cc = v[i].z - mean, dd = v[i].w - mean;
            q += (a * a + bb * bb) + (cc * cc + dd * dd);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + LN_EPS);
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
    const float4* g4 = reinterpret_cast<const float4*>(g);
    const float4* b4 = reinterpret_cast<const float4*>(b);
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int c = i * 64 + lane;
        if (c < D4) {
            const float4 gg = g4[c], bb = b4[c];
            float4 o;
            o.x = (v[i].x - mean) * rstd * gg.x + bb.x;
            o.y = (v[i].y - mean) * rstd * gg.y + bb.y;
            o.z = (v[i].z - mean) * rstd * gg.z + bb.z;
            o.w = (v[i].w - mean) * rstd * gg.w + bb.w;
            if (out_bf16)
                reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(y) + (size_t)row * D)[c] =
                    make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
            else
                reinterpret_cast<float4*>(reinterpret_cast<float*>(y) + (size_t)row * D)[c] = o;
        }
    }
}

// 16 lanes per 64-wide group (4 elements per lane), 4 groups per wave-iteration
template <bool FAST>
__global__ __launch_bounds__(256) void ln64_gelu_kernel(const float4* __restrict__ x, const float4* __restrict__ g,
                                                        const float4* __restrict__ b, void* __restrict__ y, int out_bf16,
                                                        uint32_t ngroups, int C16) {
    const int sub = threadIdx.x & 15;
    for (uint32_t grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 4; grp < ngroups; grp += (gridDim.x * blockDim.x) >> 4) {
        const uint32_t i4 = grp * 16 + sub;         // float4 index (32-bit: 64-bit modulo is very slow)
        const float4 v = x[i4];
        float s = (v.x + v.y) + (v.z + v.w);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mean = s * (1.0f / 64.0f);
        const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
        float q = (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        const float rstd = 1.0f / sqrtf(q * (1.0f / 64.0f) + LN_EPS);
        const int c4 = (int)(i4 % (uint32_t)C16);   // float4 column inside the row
        const float4 gg = g[c4], bb = b[c4];
        float4 o;
        o.x = sq_gelu<FAST>(a0 * rstd * gg.x + bb.x);
        o.y = sq_gelu<FAST>(a1 * rstd * gg.y + bb.y);
        o.z = sq_gelu<FAST>(a2 * rstd * gg.z + bb.z);
        o.w = sq_gelu<FAST>(a3 * rstd * gg.w + bb.w);
        if (out_bf16) reinterpret_cast<uint2*>(y)[i4] = make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
        else reinterpret_cast<float4*>(y)[i4] = o;
    }
}

// ln64_gelu_kernel with the rows built on the fly: x[(b, n), :] = f_tile[idx[b, n], :] + f_pos[n, :] (16 lanes per 64-wide group)
template <bool FAST>
__global__ __launch_bounds__(256) void gather_ln64_gelu_kernel(const float4* __restrict__ f_tile, const float4* __restrict__ f_pos,
                                                               const int32_t* __restrict__ idx, const float4* __restrict__ g,
                                                               const float4* __restrict__ b, void* __restrict__ y, int out_bf16,
                                                               uint32_t ngroups, int C16, int N) {
    const int sub = threadIdx.x & 15;
    const uint32_t heads = (uint32_t)C16 >> 4;       // 64-wide groups per row
    for (uint32_t grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 4; grp < ngroups; grp += (gridDim.x * blockDim.x) >> 4) {
        const uint32_t row = grp / heads, h = grp - row * heads;
        const int c4 = (int)(h * 16 + sub);          // float4 column inside the row
        const int t = idx[row];
        const uint32_t n = row % (uint32_t)N;
        float4 v = f_pos[n * (uint32_t)C16 + c4];
        if (t >= 0) {
            const float4 u = f_tile[(size_t)t * C16 + c4];
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        float s = (v.x + v.y) + (v.z + v.w);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mean = s * (1.0f / 64.0f);
        const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
        float q = (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        const float rstd = 1.0f / sqrtf(q * (1.0f / 64.0f) + LN_EPS);
        const float4 gg = g[c4], bb = b[c4];
        float4 o;
        o.x = sq_gelu<FAST>(a0 * rstd * gg.x + bb.x);
        o.y = sq_gelu<FAST>(a1 * rstd * gg.y + bb.y);
        o.z = sq_gelu<FAST>(a2 * rstd * gg.z + bb.z);
        o.w = sq_gelu<FAST>(a3 * rstd * gg.w + bb.w);
        const size_t i4 = (size_t)row * C16 + c4;
        if (out_bf16) reinterpret_cast<uint2*>(y)[i4] = make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
        else reinterpret_cast<float4*>(y)[i4] = o;
    }
}

__global__ void f32_to_bf16_kernel(const float4* __restrict__ src, uint2* __restrict__ dst, size_t n4, const float* tail_src,
                                   bf16_t* tail_dst, int tail) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t i = gid; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = src[i];
        dst[i] = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
    }
    if (gid < (size_t)tail) tail_dst[gid] = f32_to_bf16(tail_src[gid]);
}

__global__ void bf16_to_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = bf16_to_f32(src[i]);
}
// 8 elements per thread: one 16-byte load, two 16-byte stores (both pointers 16-byte aligned; the launcher checks)
__global__ __launch_bounds__(256) void bf16_to_f32_x8_kernel(const u32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const u32x4 v = src[i];
        dst[2 * i] = f32x4{__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u), __uint_as_float(v[1] << 16), __uint_as_float(v[1] & 0xffff0000u)};
        dst[2 * i + 1] = f32x4{__uint_as_float(v[2] << 16), __uint_as_float(v[2] & 0xffff0000u), __uint_as_float(v[3] << 16), __uint_as_float(v[3] & 0xffff0000u)};
    }
}

// 64x64 tile through LDS (+1 padding), coalesced on both sides; zero-fills dst columns [R, ldd)
template <typename E>
__global__ __launch_bounds__(256) void transpose_kernel(const E* __restrict__ src, int lds_, E* __restrict__ dst, int ldd,
                                                        int R, int C, long long sstride, long long dstride) {
    __shared__ E tile[64][65];
    src += (long long)blockIdx.z * sstride;
    dst += (long long)blockIdx.z * dstride;
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? src[(size_t)r * lds_ + c] : (E)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (r < ldd && c < C) dst[(size_t)c * ldd + r] = tile[tx][i];
    }
}

// many transposes in one launch: block -> job by tile prefix; same 64x64 LDS tile as transpose_kernel
template <typename E>
__global__ __launch_bounds__(256) void transpose_multi_kernel(const sq_transpose_jobs jobs) {
    __shared__ E tile[64][65];
    int j = 0;
    while (j + 1 < jobs.n && (int)blockIdx.x >= jobs.job[j + 1].tile0) ++j;
    const sq_transpose_job& J = jobs.job[j];
    int t = blockIdx.x - J.tile0;
    const int tc = (J.C + 63) / 64, tr = (J.ldd + 63) / 64;
    const int z = t / (tc * tr);
    t -= z * tc * tr;
    const E* src = reinterpret_cast<const E*>(J.src) + (long long)z * J.sstride;
    E* dst = reinterpret_cast<E*>(J.dst) + (long long)z * J.dstride;
    const int c0 = (t % tc) * 64, r0 = (t / tc) * 64;
    const int R = J.R, C = J.C, ldd = J.ldd, lds_ = J.lds;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? src[(size_t)r * lds_ + c] : (E)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (r < ldd && c < C) dst[(size_t)c * ldd + r] = tile[tx][i];
    }
}

// the same for 2-byte elements with 4-byte accesses on both sides (the generic kernel moves 128 bytes per wave instruction): a lane
// reads two neighbouring columns and writes two neighbouring rows.  Needs even lds / ldd / C and 4-byte aligned bases (the launcher
// checks; else the generic kernel runs).
__global__ __launch_bounds__(256) void transpose_multi_u16x2_kernel(const sq_transpose_jobs jobs) {
    __shared__ uint16_t tile[64][66];
    int j = 0;
    while (j + 1 < jobs.n && (int)blockIdx.x >= jobs.job[j + 1].tile0) ++j;
    const sq_transpose_job& J = jobs.job[j];
    int t = blockIdx.x - J.tile0;
    const int tc = (J.C + 63) / 64, tr = (J.ldd + 63) / 64;
    const int z = t / (tc * tr);
    t -= z * tc * tr;
    const uint16_t* src = reinterpret_cast<const uint16_t*>(J.src) + (long long)z * J.sstride;
    uint16_t* dst = reinterpret_cast<uint16_t*>(J.dst) + (long long)z * J.dstride;
    const int c0 = (t % tc) * 64, r0 = (t / tc) * 64;
    const int R = J.R, C = J.C, ldd = J.ldd, lds_ = J.lds;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 lanes x 2 elements per row, 8 rows per pass
    for (int i = ty; i < 64; i += 8) {
        const int r = r0 + i, c = c0 + 2 * tx;
        uint32_t v = 0;
        if (r < R && c < C) v = *reinterpret_cast<const uint32_t*>(src + (size_t)r * lds_ + c);      // C even: c + 1 < C too
        tile[i][2 * tx] = (uint16_t)(v & 0xffffu);
        tile[i][2 * tx + 1] = (uint16_t)(v >> 16);
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 8) {
        const int c = c0 + i, r = r0 + 2 * tx;
        if (r < ldd && c < C) *reinterpret_cast<uint32_t*>(dst + (size_t)c * ldd + r) = (uint32_t)tile[2 * tx][i] | ((uint32_t)tile[2 * tx + 1][i] << 16);
    }
}

__global__ void cast_pad_kernel(const float* __restrict__ src, int lds_, void* __restrict__ dst, int out_bf16, int ldd, int R, int C) {
    const uint32_t total = (uint32_t)R * ldd;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int r = (int)(i / (uint32_t)ldd), c = (int)(i - (uint32_t)r * ldd);
        const float v = c < C ? src[(size_t)r * lds_ + c] : 0.f;
        if (out_bf16) reinterpret_cast<bf16_t*>(dst)[i] = f32_to_bf16(v);
        else reinterpret_cast<float*>(dst)[i] = v;
    }
}

inline int grid_for(size_t work_items, int block, int cap = 256 * 8) {
    size_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > (size_t)cap) g = cap;
    return (int)g;
}

}  // namespace

int sq_k_add_pos(const float* x, const float* pos, float* X, bf16_t* Xh, int B, int N, int D, hipStream_t s) {
    SQ_REQUIRE(D % 4 == 0, "add_pos: D=%d must be a multiple of 4", D);
    SQ_REQUIRE((size_t)B * N * D / 4 < (1ull << 31), "add_pos: tensor too large for 32-bit indexing");
    const uint32_t total4 = (uint32_t)((size_t)B * N * D / 4), nd4 = (uint32_t)((size_t)N * D / 4);
    hipLaunchKernelGGL(add_pos_kernel, dim3(grid_for(total4, 256)), dim3(256), 0, s, (const float4*)x, (const float4*)pos,
                       (float4*)X, (uint2*)Xh, total4, nd4);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}

int sq_k_add_pos_gather(const float* src, const int32_t* idx, const float* pos, float* X, bf16_t* Xh, int B, int N, int D, hipStream_t s) {
    SQ_REQUIRE(D % 4 == 0, "add_pos: D=%d must be a multiple of 4", D);
    SQ_REQUIRE((size_t)B * N * D / 4 < (1ull << 31), "add_pos: tensor too large for 32-bit indexing");
    const uint32_t total4 = (uint32_t)((size_t)B * N * D / 4), nd4 = (uint32_t)((size_t)N * D / 4);
    hipLaunchKernelGGL(add_pos_gather_kernel, dim3(grid_for(total4, 256)), dim3(256), 0, s, (const float4*)src, idx, (const float4*)pos,
                       (float4*)X, (uint2*)Xh, total4, nd4, (uint32_t)(D / 4));
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}

int sq_k_token_mean(const float* X, float* out, bf16_t* outh, int B, int N, int D, hipStream_t s) {
    return sq_k_token_mean_any(X, SQ_F32, out, outh, B, N, D, s);
}

int sq_k_token_mean_any(const void* X, int in_dtype, float* out, bf16_t* outh, int B, int N, int D, hipStream_t s) {
    SQ_REQUIRE(D % 4 == 0, "token_mean: D=%d must be a multiple of 4", D);
    const int total = B * (D / 4);
    if (in_dtype == SQ_BF16 && D % 8 == 0 && (((uintptr_t)X | (uintptr_t)out | (uintptr_t)outh) & 15) == 0)
        hipLaunchKernelGGL(token_mean_bf16x8_kernel, dim3((B * (D / 8) + 63) / 64), dim3(256), 0, s, (const u32x4*)X, (float4*)out, (uint2*)outh, B, N, D / 8);
    else if (in_dtype == SQ_BF16)
        hipLaunchKernelGGL(token_mean_kernel<true>, dim3((total + 63) / 64), dim3(256), 0, s, X, (float4*)out, (uint2*)outh, B, N, D / 4);
    else
        hipLaunchKernelGGL(token_mean_kernel<false>, dim3((total + 63) / 64), dim3(256), 0, s, X, (float4*)out, (uint2*)outh, B, N, D / 4);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}

int sq_k_ln_rows(const float* x, const float* g, const float* b, void* y, int out_dtype, int R, int D, float* mean_out,
                 float* rstd_out, hipStream_t s) {
    return sq_k_ln_rows_any(x, SQ_F32, g, b, y, out_dtype, R, D, mean_out, rstd_out, s);
}

int sq_k_ln_rows_any(const void* x, int in_dtype, const float* g, const float* b, void* y, int out_dtype, int R, int D, float* mean_out,
                     float* rstd_out, hipStream_t s) {
    SQ_REQUIRE(D % 4 == 0 && D <= 4096 && D > 0, "ln_rows: D=%d must be a multiple of 4 and <= 4096", D);
    const dim3 grid((R + 3) / 4), block(256);
    const int ob = out_dtype == SQ_BF16;
    const bool wide = in_dtype == SQ_BF16 && (D == 1024 || D == 2048) && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)g | (uintptr_t)b) & 15) == 0;
    if (wide) {
        if (D == 1024) hipLaunchKernelGGL((ln_rows_bf16x8_kernel<2>), grid, block, 0, s, (const u32x4*)x, g, b, y, ob, R, mean_out, rstd_out);
        else hipLaunchKernelGGL((ln_rows_bf16x8_kernel<4>), grid, block, 0, s, (const u32x4*)x, g, b, y, ob, R, mean_out, rstd_out);
    } else if (in_dtype == SQ_BF16) {
        if (D <= 1024) hipLaunchKernelGGL((ln_rows_kernel<4, true>), grid, block, 0, s, x, g, b, y, ob, R, D, mean_out, rstd_out);
        else if (D <= 2048) hipLaunchKernelGGL((ln_rows_kernel<8, true>), grid, block, 0, s, x, g, b, y, ob, R, D, mean_out, rstd_out);
        else hipLaunchKernelGGL((ln_rows_kernel<16, true>), grid, block, 0, s, x, g, b, y, ob, R, D, mean_out, rstd_out);
    } else {
        if (D <= 1024) hipLaunchKernelGGL((ln_rows_kernel<4, false>), grid, block, 0, s, x, g, b, y, ob, R, D, mean_out, rstd_out);
        else if (D <= 2048) hipLaunchKernelGGL((ln_rows_kernel<8, false>), grid, block, 0, s, x, g, b, y, ob, R, D, mean_out, rstd_out);
        else hipLaunchKernelGGL((ln_rows_kernel<16, false>), grid, block, 0, s, x, g, b, y, ob, R, D, mean_out, rstd_out);
    }
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}

int sq_k_ln64_gelu(const float* x, const float* g, const float* b, void* y, int out_dtype, int R, int C, hipStream_t s) {
    SQ_REQUIRE(C % 64 == 0, "ln64_gelu: C=%d must be a multiple of 64", C);
    SQ_REQUIRE((size_t)R * (C / 4) < (1ull << 31), "ln64_gelu: tensor too large for 32-bit indexing");
    const uint32_t ngroups = (uint32_t)((size_t)R * (C / 64));
    // bf16 output: the 7-term erf (sq_common.h); fp32 (parity mode): erff
    if (out_dtype == SQ_BF16) hipLaunchKernelGGL(ln64_gelu_kernel<true>, dim3(grid_for((size_t)ngroups * 16, 256)), dim3(256), 0, s, (const float4*)x,
                       (const float4*)g, (const float4*)b, y, out_dtype == SQ_BF16, ngroups, C / 4);
    else hipLaunchKernelGGL(ln64_gelu_kernel<false>, dim3(grid_for((size_t)ngroups * 16, 256)), dim3(256), 0, s, (const float4*)x,
                       (const float4*)g, (const float4*)b, y, out_dtype == SQ_BF16, ngroups, C / 4);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}

int sq_k_gather_ln64_gelu(const float* f_tile, const float* f_pos, const int32_t* idx, const float* g, const float* b, void* y, int out_dtype,
                          int B, int N, int C, hipStream_t s) {
    SQ_REQUIRE(f_tile && f_pos && idx && g && b && y, "gather_ln64_gelu: null pointer");
    SQ_REQUIRE(C % 64 == 0 && B >= 1 && N >= 1, "gather_ln64_gelu: C=%d must be a multiple of 64 (B=%d N=%d)", C, B, N);
    SQ_REQUIRE((((uintptr_t)f_tile | (uintptr_t)f_pos | (uintptr_t)g | (uintptr_t)b | (uintptr_t)y) & 15) == 0, "gather_ln64_gelu: operands must be 16-byte aligned");
    SQ_REQUIRE((size_t)B * N * (C / 64) < (1ull << 31), "gather_ln64_gelu: tensor too large for 32-bit group indexing");
    const uint32_t ngroups = (uint32_t)((size_t)B * N * (C / 64));
    if (out_dtype == SQ_BF16) hipLaunchKernelGGL(gather_ln64_gelu_kernel<true>, dim3(grid_for((size_t)ngroups * 16, 256)), dim3(256), 0, s, (const float4*)f_tile,
                       (const float4*)f_pos, idx, (const float4*)g, (const float4*)b, y, 1, ngroups, C / 4, N);
    else hipLaunchKernelGGL(gather_ln64_gelu_kernel<false>, dim3(grid_for((size_t)ngroups * 16, 256)), dim3(256), 0, s, (const float4*)f_tile,
                       (const float4*)f_pos, idx, (const float4*)g, (const float4*)b, y, 0, ngroups, C / 4, N);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}

int sq_k_f32_to_bf16(const float* src, bf16_t* dst, size_t n, hipStream_t s) {
    const size_t n4 = n / 4;
    const int tail = (int)(n - n4 * 4);
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(n4 ? n4 : 1, 256)), dim3(256), 0, s, (const float4*)src, (uint2*)dst,
                       n4, src + n4 * 4, dst + n4 * 4, tail);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}

int sq_k_bf16_to_f32(const bf16_t* src, float* dst, size_t n, hipStream_t s) {
    size_t done = 0;
    if (n >= 8 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0) {
        const size_t n8 = n / 8;
        hipLaunchKernelGGL(bf16_to_f32_x8_kernel, dim3(grid_for(n8, 256)), dim3(256), 0, s, (const u32x4*)src, (f32x4*)dst, n8);
        SQ_LAUNCH_CHECK();
        done = n8 * 8;
    }
    if (done < n) {
        hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(grid_for(n - done, 256)), dim3(256), 0, s, src + done, dst + done, n - done);
        SQ_LAUNCH_CHECK();
    }
    return SQ_OK;
}

int sq_k_transpose(const void* src, int lds_, void* dst, int ldd, int R, int C, int elem_size, int batch,
                   long long sstride, long long dstride, hipStream_t s) {
    SQ_REQUIRE(ldd >= R && lds_ >= C, "transpose: ldd=%d < R=%d or lds=%d < C=%d", ldd, R, lds_, C);
    const dim3 grid((C + 63) / 64, (ldd + 63) / 64, batch), block(256);
    if (elem_size == 2)
        hipLaunchKernelGGL(transpose_kernel<uint16_t>, grid, block, 0, s, (const uint16_t*)src, lds_, (uint16_t*)dst, ldd, R, C, sstride, dstride);
    else if (elem_size == 4)
        hipLaunchKernelGGL(transpose_kernel<uint32_t>, grid, block, 0, s, (const uint32_t*)src, lds_, (uint32_t*)dst, ldd, R, C, sstride, dstride);
    else { sq_set_error("transpose: elem_size %d", elem_size); return SQ_ERR_ARG; }
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}

int sq_transpose_jobs_add(sq_transpose_jobs* jobs, const void* src, int lds_, void* dst, int ldd, int R, int C, int batch,
                          long long sstride, long long dstride) {
    SQ_REQUIRE(jobs->n < SQ_MAX_TRANSPOSE_JOBS, "transpose_multi: more than %d jobs", SQ_MAX_TRANSPOSE_JOBS);
    SQ_REQUIRE(ldd >= R && lds_ >= C && batch >= 1, "transpose_multi: ldd=%d < R=%d or lds=%d < C=%d", ldd, R, lds_, C);
    sq_transpose_job& J = jobs->job[jobs->n++];
    J.src = src; J.dst = dst; J.lds = lds_; J.ldd = ldd; J.R = R; J.C = C; J.sstride = sstride; J.dstride = dstride;
    J.tile0 = jobs->tiles;
    jobs->tiles += ((C + 63) / 64) * ((ldd + 63) / 64) * batch;
    return SQ_OK;
}

int sq_k_transpose_multi(const sq_transpose_jobs& jobs, int elem_size, hipStream_t s) {
    if (jobs.n == 0) return SQ_OK;
    bool pairs = elem_size == 2;
    for (int i = 0; i < jobs.n && pairs; ++i) {
        const sq_transpose_job& J = jobs.job[i];
        pairs = J.lds % 2 == 0 && J.ldd % 2 == 0 && J.C % 2 == 0 && ((uintptr_t)J.src & 3) == 0 && ((uintptr_t)J.dst & 3) == 0 &&
                J.sstride % 2 == 0 && J.dstride % 2 == 0;
    }
    if (pairs) hipLaunchKernelGGL(transpose_multi_u16x2_kernel, dim3(jobs.tiles), dim3(256), 0, s, jobs);
    else if (elem_size == 2) hipLaunchKernelGGL(transpose_multi_kernel<uint16_t>, dim3(jobs.tiles), dim3(256), 0, s, jobs);
    else if (elem_size == 4) hipLaunchKernelGGL(transpose_multi_kernel<uint32_t>, dim3(jobs.tiles), dim3(256), 0, s, jobs);
    else { sq_set_error("transpose_multi: elem_size %d", elem_size); return SQ_ERR_ARG; }
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}

int sq_k_cast_pad(const float* src, int lds_, void* dst, int dst_dtype, int ldd, int R, int C, hipStream_t s) {
    SQ_REQUIRE(ldd >= C && (size_t)R * ldd < (1ull << 31), "cast_pad: ldd=%d < C=%d or tensor too large", ldd, C);
    hipLaunchKernelGGL(cast_pad_kernel, dim3(grid_for((size_t)R * ldd, 256)), dim3(256), 0, s, src, lds_, dst,
                       dst_dtype == SQ_BF16, ldd, R, C);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}
