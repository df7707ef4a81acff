// Per-slide k-Means(100) + cluster means on the GPU, batched over slides.
//
// Stands behind pre_processing/kmean_features.py:96-108:
//     KMeans(n_clusters=100, random_state=0).fit(features).labels_ ; per-label np.mean
// i.e. scikit-learn's KMeans defaults (k-means++ seeding with 2+log(k) local trials, Lloyd,
// max_iter 300, tol 1e-4) with the arithmetic defined in oracle/kmeans_oracle.py:
//   * centring: column mean by adding rows in index order in fp32, X - mean in fp32
//   * seeding distances  fl32(max(0, (-2 x.c + |x|^2) + |c|^2)) with fp64 dot products -- every
//     candidate is a data point, so all of them come from ONE fp64 Gram matrix G = Xc Xc^T
//     (v_mfma_f64_16x16x4_f64); the 99 dependent seeding steps then only read rows of G
//   * potentials fl32(sum_fp64), cumsum in fp64, searchsorted(side=left), first-minimum argmin
//   * Lloyd: |c|^2 - 2 x.c in fp64 on fp32 centres (fp64 MFMA), strict-< argmin, fp64 member
//     sums in index order, fl32(sum/count), empty-cluster relocation, centre-shift / label stop rule
//   * cluster means of the ORIGINAL features: member rows added in index order in fp32, / count
// One workgroup per slide runs the whole seeding loop; slides are independent (grid dimension).
#include "../../include/sequoia_hip.h"
#include "sq_common.h"

namespace {

constexpr int KM_THREADS = 1024;
constexpr int KM_MAX_TRIALS = 8;
constexpr int KM_MAX_K = 256;
constexpr int KM_DOT_SLICES = 8;        // K slices of the centre x point products

struct KmState {          // per slide, device memory
    int iter;             // Lloyd iterations run
    int done;             // 1 once converged / max_iter reached
    int strict;           // labels unchanged between two iterations
    int n_empty;          // empty clusters found in the current iteration
    double tol;           // 1e-4 * mean(var(X, axis=0))
    double shift;         // sum of squared centre shifts of the current iteration
};

// ------------------------------------------------------------------------------------------
// block-wide helpers (blockDim.x multiple of 64, <= 1024)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_reduce_sum(double v, double* sh /*[16]*/) {
    v = wave_sum_f64(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    double r = 0.0;
    for (int i = 0; i < nw; ++i) r += sh[i];     // same order on every thread -> identical result
    return r;
}

__device__ __forceinline__ int block_reduce_sum_int(int v, int* sh /*[16]*/) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    int r = 0;
    for (int i = 0; i < nw; ++i) r += sh[i];
    return r;
}

// T values per thread reduced together: 2 barriers for the whole set (the seeding loop is barrier-latency bound)
template <int T>
__device__ __forceinline__ void block_reduce_sum_n(double (&v)[T], double* sh /*[16 * T]*/) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int t = 0; t < T; ++t) v[t] = wave_sum_f64(v[t]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int t = 0; t < T; ++t) sh[wv * T + t] = v[t];
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < T; ++t) {
        double r = 0.0;
        for (int i = 0; i < nw; ++i) r += sh[i * T + t];
        v[t] = r;
    }
}

template <int T>
__device__ __forceinline__ void block_reduce_sum_n_int(int (&v)[T], int* sh /*[16 * T]*/) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[t] += __shfl_xor(v[t], o, 64);
    }
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int t = 0; t < T; ++t) sh[wv * T + t] = v[t];
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < T; ++t) {
        int r = 0;
        for (int i = 0; i < nw; ++i) r += sh[i * T + t];
        v[t] = r;
    }
}

// ------------------------------------------------------------------------------------------
// 1. centring + tolerance
// ------------------------------------------------------------------------------------------
// Column means in numpy's axis-0 order (rows added one after the other into an fp32 accumulator), X - mean in fp32,
// and the per-column variance the stop tolerance is built from (fp64).
// A block owns 64 columns.  Its 256 threads stream 128-row x 64-column tiles into LDS with 32 loads in flight per
// thread (the old thread-per-column loop had ONE dependent load in flight: 1.75 ms for 8 MB); the order-sensitive
// fp32 sum is then taken by one wave from LDS, row after row.  The fp64 moments only feed `tol`: every thread sums
// its own rows, the four row-lanes of a column are combined in a fixed order.
constexpr int KC_ROWS = 128, KC_COLS = 64;
__global__ __launch_bounds__(256) void km_center_kernel(const float* __restrict__ X, float* __restrict__ Xc, double* __restrict__ colvar, int n, int D) {
    __shared__ float tile[KC_ROWS][KC_COLS];
    __shared__ double part[4][KC_COLS];
    __shared__ float mean_s[KC_COLS];
    const int col = threadIdx.x & 63, rl = threadIdx.x >> 6;        // column inside the block, row-lane 0..3
    const int d = blockIdx.x * KC_COLS + col;
    const bool dok = d < D;
    const size_t base = (size_t)blockIdx.y * n * D;
    float acc = 0.f;                                                 // wave 0 only: the fp32 column sum, rows in order
    double s64 = 0.0;
    for (int r0 = 0; r0 < n; r0 += KC_ROWS) {
        float v[KC_ROWS / 4];
#pragma unroll
        for (int u = 0; u < KC_ROWS / 4; ++u) {
            const int r = r0 + 4 * u + rl;
            v[u] = (dok && r < n) ? X[base + (size_t)r * D + d] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < KC_ROWS / 4; ++u) { tile[4 * u + rl][col] = v[u]; s64 += (double)v[u]; }
        __syncthreads();
        if (rl == 0) {
            const int lim = min(KC_ROWS, n - r0);
            for (int r = 0; r < lim; ++r) acc += tile[r][col];
        }
        __syncthreads();
    }
    part[rl][col] = s64;
    if (rl == 0) mean_s[col] = acc / (float)n;
    __syncthreads();
    const float mean = mean_s[col];
    const double m64 = (((part[0][col] + part[1][col]) + part[2][col]) + part[3][col]) / n;
    __syncthreads();
    double q = 0.0;
    for (int r0 = 0; r0 < n; r0 += KC_ROWS) {
        float v[KC_ROWS / 4];
#pragma unroll
        for (int u = 0; u < KC_ROWS / 4; ++u) {
            const int r = r0 + 4 * u + rl;
            v[u] = (dok && r < n) ? X[base + (size_t)r * D + d] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < KC_ROWS / 4; ++u) {
            const int r = r0 + 4 * u + rl;
            if (dok && r < n) {
                Xc[base + (size_t)r * D + d] = v[u] - mean;
                const double t = (double)v[u] - m64;
                q += t * t;
            }
        }
    }
    part[rl][col] = q;
    __syncthreads();
    if (rl == 0 && dok) colvar[(size_t)blockIdx.y * D + d] = (((part[0][col] + part[1][col]) + part[2][col]) + part[3][col]) / n;
}

__global__ void km_tol_kernel(const double* __restrict__ colvar, KmState* __restrict__ st, int D, double tol_rel) {
    __shared__ double sh[16];
    double s = 0.0;
    for (int d = threadIdx.x; d < D; d += blockDim.x) s += colvar[(size_t)blockIdx.x * D + d];
    s = block_reduce_sum(s, sh);
    if (threadIdx.x == 0) {
        KmState& k = st[blockIdx.x];
        k.iter = 0; k.done = 0; k.strict = 0; k.n_empty = 0; k.shift = 0.0;
        k.tol = s / D * tol_rel;
    }
}

// ------------------------------------------------------------------------------------------
// 2. fp64 MFMA GEMM with fp32 operands:  C[M,N] (f64) = A[M,K] . B[N,K]^T   (batched over slides)
//    block 64x64, 4 waves x (2x2 tiles of 16x16), K-chunk 32 staged in LDS as [k][row] doubles
// ------------------------------------------------------------------------------------------
// ksl > 1: the contraction is cut into ksl slices (grid z = slide * ksl + slice), slice partials land in consecutive
// C planes and the consumer adds them in slice order (short products -- 100 centres x 1000 points -- otherwise run on
// 32 blocks with 64 dependent K steps each).
__global__ __launch_bounds__(256) void km_dgemm_nt_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                          double* __restrict__ C, int M, int N, int K, long long sA,
                                                          long long sB, long long sC, int ksl, int sym) {
    // sym: A == B (Gram matrix): only the blocks on and above the diagonal are computed, each also stores its mirror
    // image -- bit-exact, the products commute and the K order is the same on both sides of the diagonal
    if (sym && blockIdx.x < blockIdx.y) return;
    __shared__ double sa[32][64 + 2];
    __shared__ double sb[32][64 + 2];
    const int slide = blockIdx.z / ksl, slice = blockIdx.z - slide * ksl;
    A += (long long)slide * sA; B += (long long)slide * sB; C += (long long)blockIdx.z * sC;
    const int kper = ((K + ksl - 1) / ksl + 31) / 32 * 32;
    const int k_lo = slice * kper, k_hi = min(K, k_lo + kper);
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    f64x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
    // loader: thread -> (row = tid / 4 within 64, k4 = tid % 4 ... two passes of 16 k each).  The next K step's rows are
    // fetched into registers before this step's MFMAs: a lone 64 x 64 block per CU otherwise spends its time waiting
    // for one round of global loads per step (the 1000 x 1000 Gram matrix: 132 -> 75 us)
    const int lrow = tid >> 2, lk = (tid & 3) * 4;
    const bool arow = m0 + lrow < M, brow = n0 + lrow < N;
    const float* ap = A + (size_t)(arow ? m0 + lrow : 0) * K;
    const float* bp = B + (size_t)(brow ? n0 + lrow : 0) * K;
    float4 va[2], vb[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int kk = k0 + half * 16 + lk;
            va[half] = make_float4(0.f, 0.f, 0.f, 0.f);
            vb[half] = va[half];
            if (arow && kk < k_hi) va[half] = *reinterpret_cast<const float4*>(ap + kk);
            if (brow && kk < k_hi) vb[half] = *reinterpret_cast<const float4*>(bp + kk);
        }
    };
    if (k_lo < k_hi) fetch(k_lo);
    for (int k0 = k_lo; k0 < k_hi; k0 += 32) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int kr = half * 16 + lk;
            sa[kr + 0][lrow] = va[half].x; sa[kr + 1][lrow] = va[half].y; sa[kr + 2][lrow] = va[half].z; sa[kr + 3][lrow] = va[half].w;
            sb[kr + 0][lrow] = vb[half].x; sb[kr + 1][lrow] = vb[half].y; sb[kr + 2][lrow] = vb[half].z; sb[kr + 3][lrow] = vb[half].w;
        }
        __syncthreads();
        if (k0 + 32 < k_hi) fetch(k0 + 32);
#pragma unroll
        for (int ks = 0; ks < 32; ks += 4) {
            const int k = ks + (lane >> 4);
            double fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = sa[k][wm * 32 + i * 16 + (lane & 15)];
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = sb[k][wn * 32 + j * 16 + (lane & 15)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 32 + i * 16 + (lane >> 4) + 4 * r;
                const int n = n0 + wn * 32 + j * 16 + (lane & 15);
                if (m < M && n < N) {
                    C[(size_t)m * N + n] = acc[i][j][r];
                    if (sym && blockIdx.x != blockIdx.y) C[(size_t)n * N + m] = acc[i][j][r];
                }
            }
}

// ------------------------------------------------------------------------------------------
// 3. k-means++ seeding: one workgroup per slide, all k-1 dependent steps inside the kernel
// ------------------------------------------------------------------------------------------
// Squared distances of the seeding step as fp32, once for all pairs: exactly pairwise.py:647-651 per element
// (d = -2 X.Y^T; d += XX; d += YY in fp64 from the Gram matrix; cast fp32; max(., 0)), row c = distances from point c.
// The seeding kernel then reads 4-byte distances (16 bytes per thread and candidate) instead of rebuilding them from
// 8-byte Gram entries inside its 99 dependent steps.
__global__ __launch_bounds__(256) void km_dist_f32_kernel(const double* __restrict__ Gall, float* __restrict__ Dall, int n) {
    const int c = blockIdx.x;
    const double* G = Gall + (size_t)blockIdx.y * n * n;
    float* Dm = Dall + (size_t)blockIdx.y * n * n;
    const double gcc = G[(size_t)c * n + c];
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        double d = -2.0 * G[(size_t)c * n + j];
        d += gcc;
        d += G[(size_t)j * n + j];
        const float f = (float)d;
        Dm[(size_t)c * n + j] = f > 0.f ? f : 0.f;
    }
}

// One step = scan -> candidates -> candidate potentials -> choice, i.e. three block-wide exchanges and ONE round of
// dependent global loads (the candidates' rows of G).  The exchanges go through LDS arrays that alternate with the
// step parity, so each costs a single barrier; the candidate distances stay in registers and the winner's are reused
// for the `closest` update (no second read of G).  Arithmetic (scan association, reduction orders) is unchanged from
// the first version of this kernel, results are bit-identical.
template <int PTS>   // points per thread (n <= PTS * blockDim.x)
__global__ __launch_bounds__(KM_THREADS) void km_seed_kernel(const float* __restrict__ Dall, const double* __restrict__ uniforms,
                                                             int first, int n, int k, int trials, int* __restrict__ seeds) {
    __shared__ double sh[2][16 * KM_MAX_TRIALS];
    __shared__ int shi[2][16 * KM_MAX_TRIALS];
    __shared__ double scan_w[2][16];
    const float* Dm = Dall + (size_t)blockIdx.x * n * n;
    int* out = seeds + (size_t)blockIdx.x * k;
    const bool vec4 = PTS == 4 && (n & 3) == 0;        // a thread's four points are one aligned 16-byte load
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6;

    // thread owns the CONTIGUOUS points [tid*PTS, tid*PTS+PTS) so the scan is a plain blocked scan
    float closest[PTS];
#pragma unroll
    for (int q = 0; q < PTS; ++q) {
        const int j = tid * PTS + q;
        closest[q] = j < n ? Dm[(size_t)first * n + j] : 0.f;
    }
    float pot;
    {
        double part = 0.0;
#pragma unroll
        for (int q = 0; q < PTS; ++q) part += (double)closest[q];
        part = wave_sum_f64(part);
        if (lane == 0) sh[1][wv] = part;
        __syncthreads();
        double r = 0.0;
        for (int i = 0; i < nw; ++i) r += sh[1][i];
        pot = (float)r;
    }
    if (tid == 0) out[0] = first;

    for (int c = 1; c < k; ++c) {
        const int par = c & 1;
        // stable_cumsum (fp64) of closest: per-thread serial, wave scan, block scan
        double run = 0.0, incl[PTS];
#pragma unroll
        for (int q = 0; q < PTS; ++q) { run += (double)closest[q]; incl[q] = run; }
        double wscan = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double t = __shfl_up(wscan, o, 64);
            if (lane >= o) wscan += t;
        }
        if (lane == 63) scan_w[par][wv] = wscan;
        __syncthreads();
        double woff = 0.0;
        for (int i = 0; i < wv; ++i) woff += scan_w[par][i];
        const double excl = woff + wscan - run;          // sum of everything before this thread's points
        // candidates: searchsorted(cumsum, u * pot, side='left') = #{cum < value}, clipped to n-1 (all trials at once)
#pragma unroll
        for (int t = 0; t < KM_MAX_TRIALS; ++t) {
            if (t < trials) {
                const double rv = uniforms[(size_t)(c - 1) * trials + t] * (double)pot;
                int cnt = 0;
#pragma unroll
                for (int q = 0; q < PTS; ++q)
                    cnt += __popcll(__ballot(tid * PTS + q < n && excl + incl[q] < rv));
                if (lane == 0) shi[par][wv * KM_MAX_TRIALS + t] = cnt;
            }
        }
        __syncthreads();
        int cand[KM_MAX_TRIALS];
#pragma unroll
        for (int t = 0; t < KM_MAX_TRIALS; ++t) {
            int r = 0;
            if (t < trials)
                for (int i = 0; i < nw; ++i) r += shi[par][i * KM_MAX_TRIALS + t];
            cand[t] = r > n - 1 ? n - 1 : r;
        }
        // distances to the candidates (one round of loads for all trials), potentials
        float dc[KM_MAX_TRIALS][PTS];
        double p[KM_MAX_TRIALS];
#pragma unroll
        for (int t = 0; t < KM_MAX_TRIALS; ++t) {
            if (t < trials) {
                const float* row = Dm + (size_t)cand[t] * n;
                if constexpr (PTS == 4) {
                    if (vec4 && tid * 4 < n) {
                        const float4 v = *reinterpret_cast<const float4*>(row + tid * 4);
                        dc[t][0] = v.x; dc[t][1] = v.y; dc[t][2] = v.z; dc[t][3] = v.w;
                    } else {
#pragma unroll
                        for (int q = 0; q < PTS; ++q) dc[t][q] = tid * PTS + q < n ? row[tid * PTS + q] : 0.f;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < PTS; ++q) dc[t][q] = tid * PTS + q < n ? row[tid * PTS + q] : 0.f;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < KM_MAX_TRIALS; ++t) {
            p[t] = 0.0;
            if (t < trials) {
#pragma unroll
                for (int q = 0; q < PTS; ++q)
                    if (tid * PTS + q < n) p[t] += (double)fminf(closest[q], dc[t][q]);
            } else {
#pragma unroll
                for (int q = 0; q < PTS; ++q) dc[t][q] = 0.f;
            }
        }
#pragma unroll
        for (int t = 0; t < KM_MAX_TRIALS; ++t) p[t] = wave_sum_f64(p[t]);
        if (lane == 0) {
#pragma unroll
            for (int t = 0; t < KM_MAX_TRIALS; ++t) sh[par][wv * KM_MAX_TRIALS + t] = p[t];
        }
        __syncthreads();
        float best_pot = 0.f;
        int best_t = 0;
#pragma unroll
        for (int t = 0; t < KM_MAX_TRIALS; ++t) {
            if (t < trials) {
                double r = 0.0;
                for (int i = 0; i < nw; ++i) r += sh[par][i * KM_MAX_TRIALS + t];
                const float pt = (float)r;
                if (t == 0 || pt < best_pot) { best_pot = pt; best_t = t; }     // np.argmin: first minimum
            }
        }
        int chosen = cand[0];
#pragma unroll
        for (int t = 1; t < KM_MAX_TRIALS; ++t)
            if (t == best_t) chosen = cand[t];
        pot = best_pot;
#pragma unroll
        for (int q = 0; q < PTS; ++q) {
            float f = dc[0][q];
#pragma unroll
            for (int t = 1; t < KM_MAX_TRIALS; ++t)
                if (t == best_t) f = dc[t][q];
            if (tid * PTS + q < n) closest[q] = fminf(closest[q], f);
        }
        if (tid == 0) out[c] = chosen;
    }
}

__global__ void km_gather_centers_kernel(const float* __restrict__ Xc, const int* __restrict__ seeds, float* __restrict__ centers,
                                         int n, int D, int k) {
    const int c = blockIdx.x, s = blockIdx.y;
    const int src = seeds[(size_t)s * k + c];
    for (int d = threadIdx.x; d < D; d += blockDim.x)
        centers[((size_t)s * k + c) * D + d] = Xc[((size_t)s * n + src) * D + d];
}

// ------------------------------------------------------------------------------------------
// 4. Lloyd iteration pieces (every kernel returns at once for slides that are done)
// ------------------------------------------------------------------------------------------
// |c|^2 in fp64, one wave per centre (lanes stride the row, butterfly sum): its own launch, because inside the assign
// kernel every block recomputed all k norms with 25 dependent row walks per wave (0.25 ms of a 0.3 ms kernel)
__global__ __launch_bounds__(64) void km_center_norms_kernel(const float* __restrict__ centers, double* __restrict__ cnorm,
                                                             const KmState* __restrict__ st, int D, int k, int force) {
    const int c = blockIdx.x, s = blockIdx.y;
    if (st[s].done && !force) return;
    const float* cr = centers + ((size_t)s * k + c) * D;
    double a = 0.0;
    int d = threadIdx.x;
    for (; d + 7 * 64 < D; d += 8 * 64) {          // eight loads in flight, added in the same (ascending d) order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = cr[d + 64 * u];
#pragma unroll
        for (int u = 0; u < 8; ++u) a += (double)v[u] * (double)v[u];
    }
    for (; d < D; d += 64) a += (double)cr[d] * (double)cr[d];
    a = wave_sum_f64(a);
    if (threadIdx.x == 0) cnorm[(size_t)s * k + c] = a;
}

// labels[j] = argmin_c (|c|^2 - 2 x_j.c), strict '<' (first minimum); dots from the fp64 GEMM as ksl planes of
// [k centres][n points] (a thread walks the centres of ITS point: consecutive threads read consecutive doubles)
__global__ __launch_bounds__(1024) void km_assign_kernel(const double* __restrict__ dots, const double* __restrict__ cnorm, int* __restrict__ labels,
                                                         const KmState* __restrict__ st, int n, int D, int k, int force, int ksl) {
    // block = 64 points x 16 centre groups (wave g walks centres [g*k/16, (g+1)*k/16) of its 64 points, all of their
    // slice loads in flight together); the partial minima are merged in centre order, which keeps the first minimum
    constexpr int G = 16, CB = 7;
    __shared__ double cn[KM_MAX_K];
    __shared__ double bval[G][64];
    __shared__ int blab[G][64];
    const int s = blockIdx.y;
    if (st[s].done && !force) return;
    for (int c = threadIdx.x; c < k; c += blockDim.x) cn[c] = cnorm[(size_t)s * k + c];
    __syncthreads();
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    const int c_lo = g * k / G, c_hi = (g + 1) * k / G;
    const size_t plane = (size_t)k * n;
    const double* dr = dots + (size_t)s * ksl * plane + (j < n ? j : 0);
    double best = 0.0;
    int lab = -1;
    for (int c0 = c_lo; c0 < c_hi; c0 += CB) {
        double dot[CB];
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            const int c = c0 + u < c_hi ? c0 + u : c_hi - 1;
            double a = dr[(size_t)c * n];
            for (int q = 1; q < ksl; ++q) a += dr[q * plane + (size_t)c * n];
            dot[u] = a;
        }
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            const int c = c0 + u;
            if (c < c_hi) {
                const double v = cn[c] - 2.0 * dot[u];
                if (lab < 0 || v < best) { best = v; lab = c; }
            }
        }
    }
    bval[g][lane] = best;
    blab[g][lane] = lab;
    __syncthreads();
    if (g == 0 && j < n) {
        double b = bval[0][lane];
        int l = blab[0][lane];
#pragma unroll
        for (int q = 1; q < G; ++q)
            if (blab[q][lane] >= 0 && (l < 0 || bval[q][lane] < b)) { b = bval[q][lane]; l = blab[q][lane]; }
        labels[(size_t)s * n + j] = l;
    }
}

// counting sort of point ids by label, stable in the point index: members[off[c] .. off[c+1])
__global__ __launch_bounds__(KM_THREADS) void km_members_kernel(const int* __restrict__ labels, int* __restrict__ members,
                                                                int* __restrict__ offsets, const KmState* __restrict__ st, int n,
                                                                int k, int force) {
    __shared__ int slab[4096];
    __shared__ int soff[KM_MAX_K + 1];
    const int s = blockIdx.x;
    if (st[s].done && !force) return;
    for (int j = threadIdx.x; j < n; j += blockDim.x) slab[j] = labels[(size_t)s * n + j];
    __syncthreads();
    for (int c = threadIdx.x; c < k; c += blockDim.x) {
        int cnt = 0;
        for (int j = 0; j < n; ++j) cnt += slab[j] == c;
        soff[c + 1] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        soff[0] = 0;
        for (int c = 0; c < k; ++c) soff[c + 1] += soff[c];
    }
    __syncthreads();
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        const int lab = slab[j];
        int rank = 0;
        for (int i = 0; i < j; ++i) rank += slab[i] == lab;
        members[(size_t)s * n + soff[lab] + rank] = j;
    }
    for (int c = threadIdx.x; c <= k; c += blockDim.x) offsets[(size_t)s * (k + 1) + c] = soff[c];
}

// sums[c, d] = sum over members (index order) of Xc in fp64
__global__ void km_sums_kernel(const float* __restrict__ Xc, const int* __restrict__ members, const int* __restrict__ offsets,
                               double* __restrict__ sums, const KmState* __restrict__ st, int n, int D, int k) {
    const int c = blockIdx.x, s = blockIdx.y;
    if (st[s].done) return;
    const int lo = offsets[(size_t)s * (k + 1) + c], hi = offsets[(size_t)s * (k + 1) + c + 1];
    const int* mem = members + (size_t)s * n;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        double a = 0.0;
        int i = lo;
        for (; i + 8 <= hi; i += 8) {              // eight gathered rows in flight, added in member (= index) order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = Xc[((size_t)s * n + mem[i + u]) * D + d];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += (double)v[u];
        }
        for (; i < hi; ++i) a += (double)Xc[((size_t)s * n + mem[i]) * D + d];
        sums[((size_t)s * k + c) * D + d] = a;
    }
}

// _relocate_empty_clusters_dense (_k_means_common.pyx:167-211), one workgroup per slide; rare path
__global__ __launch_bounds__(KM_THREADS) void km_relocate_kernel(const float* __restrict__ Xc, const float* __restrict__ centers,
                                                                 const int* __restrict__ labels, int* __restrict__ offsets,
                                                                 double* __restrict__ sums, double* __restrict__ weights,
                                                                 double* __restrict__ dist_ws, const KmState* __restrict__ st,
                                                                 int n, int D, int k) {
    __shared__ int empty[KM_MAX_K];
    __shared__ int n_empty;
    __shared__ int far_idx;
    const int s = blockIdx.x;
    if (st[s].done) return;
    const int* off = offsets + (size_t)s * (k + 1);
    double* w = weights + (size_t)s * k;
    for (int c = threadIdx.x; c < k; c += blockDim.x) w[c] = (double)(off[c + 1] - off[c]);
    __syncthreads();
    if (threadIdx.x == 0) {
        int e = 0;
        for (int c = 0; c < k; ++c)
            if (w[c] == 0.0) empty[e++] = c;
        n_empty = e;
    }
    __syncthreads();
    if (n_empty == 0) return;
    // distances of every point to its (old) centre: ((X - centers_old[labels])**2).sum(axis=1)
    double* dist = dist_ws + (size_t)s * n;
    for (int j = threadIdx.x >> 6; j < n; j += blockDim.x >> 6) {
        const float* xr = Xc + ((size_t)s * n + j) * D;
        const float* cr = centers + ((size_t)s * k + labels[(size_t)s * n + j]) * D;
        double a = 0.0;
        for (int d = threadIdx.x & 63; d < D; d += 64) { const float t = xr[d] - cr[d]; a += (double)(t * t); }
        a = wave_sum_f64(a);
        if ((threadIdx.x & 63) == 0) dist[j] = a;
    }
    __syncthreads();
    for (int e = 0; e < n_empty; ++e) {
        if (threadIdx.x == 0) {          // farthest remaining point (descending distance, ties by index)
            int bi = -1; double bd = -1.0;
            for (int j = 0; j < n; ++j)
                if (dist[j] > bd) { bd = dist[j]; bi = j; }
            far_idx = bd > 0.0 ? bi : -1;
            if (bi >= 0) dist[bi] = -2.0;
        }
        __syncthreads();
        const int fi = far_idx;
        if (fi < 0) break;               // np.max(distances) == 0: relocation is pointless
        const int new_id = empty[e], old_id = labels[(size_t)s * n + fi];
        for (int d = threadIdx.x; d < D; d += blockDim.x) {
            const double x = (double)Xc[((size_t)s * n + fi) * D + d];
            sums[((size_t)s * k + old_id) * D + d] -= x;
            sums[((size_t)s * k + new_id) * D + d] = x;
        }
        __syncthreads();
        if (threadIdx.x == 0) { w[new_id] = 1.0; w[old_id] -= 1.0; }
        __syncthreads();
    }
}

// _average_centers + _center_shift: one block per (cluster, slide) writes the new centre row into the OTHER centre
// buffer (a cluster left empty copies the heaviest cluster's new row, so rows cannot be updated in place) and its
// share of the squared shift; slides that are done copy their rows through, so both buffers stay valid for them.
__global__ __launch_bounds__(256) void km_update_centers_kernel(const float* __restrict__ centers, float* __restrict__ centers_new,
                                                                const double* __restrict__ sums, const double* __restrict__ weights,
                                                                double* __restrict__ shift_part, const KmState* __restrict__ st,
                                                                int D, int k) {
    __shared__ double sh[16];
    __shared__ int amax_s;
    const int c = blockIdx.x, s = blockIdx.y;
    const float* old_row = centers + ((size_t)s * k + c) * D;
    float* new_row = centers_new + ((size_t)s * k + c) * D;
    if (st[s].done) {
        for (int d = threadIdx.x; d < D; d += blockDim.x) new_row[d] = old_row[d];
        return;
    }
    const double* w = weights + (size_t)s * k;
    if (threadIdx.x == 0) {
        int a = 0;
        for (int i = 1; i < k; ++i)
            if (w[i] > w[a]) a = i;      // np.argmax: first maximum
        amax_s = a;
    }
    __syncthreads();
    const int src = w[c] > 0.0 ? c : amax_s;
    const double ws = w[src];
    const double* srow = sums + ((size_t)s * k + src) * D;
    double shift = 0.0;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        const float nv = (float)(srow[d] / ws);
        const double df = (double)nv - (double)old_row[d];
        shift += df * df;
        new_row[d] = nv;
    }
    shift = block_reduce_sum(shift, sh);
    if (threadIdx.x == 0) shift_part[(size_t)s * k + c] = shift;
}

// the stop rule of _kmeans_single_lloyd; one workgroup per slide
__global__ __launch_bounds__(KM_THREADS) void km_finish_iter_kernel(const double* __restrict__ shift_part, const int* __restrict__ labels,
                                                                    int* __restrict__ labels_old, KmState* __restrict__ st, int n, int k,
                                                                    int max_iter) {
    __shared__ int shi[16];
    const int s = blockIdx.x;
    if (st[s].done) return;
    int diff = 0;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        const int l = labels[(size_t)s * n + j];
        diff += l != labels_old[(size_t)s * n + j];
        labels_old[(size_t)s * n + j] = l;
    }
    diff = block_reduce_sum_int(diff, shi);
    if (threadIdx.x == 0) {
        double shift = 0.0;
        for (int c = 0; c < k; ++c) shift += shift_part[(size_t)s * k + c];      // cluster order
        KmState& ks = st[s];
        ks.iter += 1;
        ks.shift = shift;
        if (diff == 0) { ks.strict = 1; ks.done = 1; }
        else if (shift <= ks.tol) ks.done = 1;
        else if (ks.iter >= max_iter) ks.done = 1;
    }
}

__global__ void km_restore_strict_kernel(int* __restrict__ labels, const int* __restrict__ labels_old, const KmState* __restrict__ st, int n) {
    const int s = blockIdx.y;
    if (!st[s].strict) return;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) labels[(size_t)s * n + j] = labels_old[(size_t)s * n + j];
}

__global__ void km_count_done_kernel(const KmState* __restrict__ st, int S, int* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int d = 0, q = 0;
        for (int s = 0; s < S; ++s) { d += st[s].done; q += st[s].strict; }
        out[0] = d;
        out[1] = q;
    }
}

__global__ void km_report_kernel(const KmState* __restrict__ st, int S, int* __restrict__ n_iter) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < S && n_iter) n_iter[s] = st[s].iter;
}

// kmean_features.py:99-105: np.mean(features[labels == c], axis=0): fp32 adds in index order, / count
__global__ void km_cluster_means_kernel(const float* __restrict__ X, const int* __restrict__ members, const int* __restrict__ offsets,
                                        float* __restrict__ out, int n, int D, int k) {
    const int c = blockIdx.x, s = blockIdx.y;
    const int lo = offsets[(size_t)s * (k + 1) + c], hi = offsets[(size_t)s * (k + 1) + c + 1];
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float a = 0.f;
        for (int i = lo; i < hi; ++i) a += X[((size_t)s * n + members[(size_t)s * n + i]) * D + d];
        out[((size_t)s * k + c) * D + d] = a / (float)(hi - lo);      // empty cluster -> 0/0 = NaN like numpy
    }
}

struct KmBufs {
    float* Xc; double* colvar; double* G; float* centers; float* centers2; double* shift_part; double* dots; double* sums; double* weights; double* dist;
    int* seeds; int* labels_old; int* members; int* offsets; int* done_count; KmState* st;
    float* dist32;
    size_t bytes;
};

void km_bufs(int S, int n, int D, int k, char* base, KmBufs* o) {
    size_t off = 0;
    auto take = [&](size_t bytes) { off = sq_align_up(off, 256); char* p = base ? base + off : nullptr; off += bytes; return (void*)p; };
    o->Xc = (float*)take((size_t)S * n * D * 4);
    o->colvar = (double*)take((size_t)S * D * 8);
    o->G = (double*)take((size_t)S * n * n * 8);
    o->centers = (float*)take((size_t)S * k * D * 4);
    o->centers2 = (float*)take((size_t)S * k * D * 4);
    o->shift_part = (double*)take((size_t)S * k * 8 * 2);      // [S][k] shift shares, then [S][k] centre norms
    o->dots = (double*)take((size_t)S * KM_DOT_SLICES * n * k * 8);
    o->sums = (double*)take((size_t)S * k * D * 8);
    o->weights = (double*)take((size_t)S * k * 8);
    o->dist = (double*)take((size_t)S * n * 8);
    o->seeds = (int*)take((size_t)S * k * 4);
    o->labels_old = (int*)take((size_t)S * n * 4);
    o->members = (int*)take((size_t)S * n * 4);
    o->offsets = (int*)take((size_t)S * (k + 1) * 4);
    o->done_count = (int*)take(256);
    o->st = (KmState*)take((size_t)S * sizeof(KmState));
    o->dist32 = (float*)take((size_t)S * n * n * 4);
    o->bytes = sq_align_up(off, 256);
}

}  // namespace

extern "C" size_t sq_kmeans_workspace_bytes(int n_slides, int n_samples, int dim, int n_clusters) {
    if (n_slides < 1 || n_samples < 1 || dim < 1 || n_clusters < 1) return 0;
    KmBufs b;
    km_bufs(n_slides, n_samples, dim, n_clusters, nullptr, &b);
    return b.bytes;
}

extern "C" int sq_kmeans_fit(const float* X, int S, int n, int D, int k, int first_center, const double* uniforms,
                             int n_local_trials, int max_iter, double tol, int32_t* labels, float* cluster_features,
                             int32_t* seed_indices, int32_t* n_iter, void* workspace, size_t workspace_bytes,
                             sq_stream_t stream_) {
    hipStream_t st = (hipStream_t)stream_;
    SQ_REQUIRE(X && labels && uniforms && workspace, "kmeans: null pointer");
    SQ_REQUIRE(S >= 1 && n >= k && k >= 1 && k <= KM_MAX_K, "kmeans: need n_samples >= n_clusters and 1 <= n_clusters <= %d (n=%d k=%d)", KM_MAX_K, n, k);
    SQ_REQUIRE(n <= 4096, "kmeans: n_samples=%d > 4096 per slide (max_patch_number default is 4000)", n);
    SQ_REQUIRE(D % 4 == 0, "kmeans: dim=%d must be a multiple of 4", D);
    SQ_REQUIRE(n_local_trials >= 1 && n_local_trials <= KM_MAX_TRIALS, "kmeans: n_local_trials=%d", n_local_trials);
    SQ_REQUIRE(first_center >= 0 && first_center < n, "kmeans: first_center=%d", first_center);
    KmBufs b;
    km_bufs(S, n, D, k, (char*)workspace, &b);
    if (b.bytes > workspace_bytes) {
        sq_set_error("kmeans: workspace %zu < required %zu", workspace_bytes, b.bytes);
        return SQ_ERR_WORKSPACE;
    }
    hipLaunchKernelGGL(km_center_kernel, dim3((D + KC_COLS - 1) / KC_COLS, S), dim3(256), 0, st, X, b.Xc, b.colvar, n, D);
    SQ_LAUNCH_CHECK();
    hipLaunchKernelGGL(km_tol_kernel, dim3(S), dim3(256), 0, st, b.colvar, b.st, D, tol);
    SQ_LAUNCH_CHECK();
    // Gram matrix of the centred data (fp64)
    hipLaunchKernelGGL(km_dgemm_nt_kernel, dim3((n + 63) / 64, (n + 63) / 64, S), dim3(256), 0, st, b.Xc, b.Xc, b.G, n, n, D,
                       (long long)n * D, (long long)n * D, (long long)n * n, 1, 1);
    SQ_LAUNCH_CHECK();
    hipLaunchKernelGGL(km_dist_f32_kernel, dim3(n, S), dim3(256), 0, st, b.G, b.dist32, n);
    SQ_LAUNCH_CHECK();
    if (n <= KM_THREADS)       // 4 waves x 4 points per thread: cheaper barriers and wave exchanges than 16 waves x 1
        hipLaunchKernelGGL(km_seed_kernel<4>, dim3(S), dim3(256), 0, st, b.dist32, uniforms, first_center, n, k, n_local_trials, b.seeds);
    else if (n <= KM_THREADS)
        hipLaunchKernelGGL(km_seed_kernel<1>, dim3(S), dim3(KM_THREADS), 0, st, b.dist32, uniforms, first_center, n, k, n_local_trials, b.seeds);
    else if (n <= 2 * KM_THREADS)
        hipLaunchKernelGGL(km_seed_kernel<2>, dim3(S), dim3(KM_THREADS), 0, st, b.dist32, uniforms, first_center, n, k, n_local_trials, b.seeds);
    else
        hipLaunchKernelGGL(km_seed_kernel<4>, dim3(S), dim3(KM_THREADS), 0, st, b.dist32, uniforms, first_center, n, k, n_local_trials, b.seeds);
    SQ_LAUNCH_CHECK();
    if (seed_indices) SQ_HIP_CHECK(hipMemcpyAsync(seed_indices, b.seeds, (size_t)S * k * 4, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(km_gather_centers_kernel, dim3(k, S), dim3(256), 0, st, b.Xc, b.seeds, b.centers, n, D, k);
    SQ_LAUNCH_CHECK();
    SQ_HIP_CHECK(hipMemsetAsync(b.labels_old, 0xff, (size_t)S * n * 4, st));          // labels_old = -1

    float* cur = b.centers;          // centres of the iteration in flight; the update writes the other buffer
    float* nxt = b.centers2;
    const int ksl = D >= 32 * KM_DOT_SLICES ? KM_DOT_SLICES : 1;
    auto e_step = [&](int force) -> int {
        // dots[slice][centre][point] = centres . points (fp64), K sliced so that the launch fills the chip
        hipLaunchKernelGGL(km_dgemm_nt_kernel, dim3((n + 63) / 64, (k + 63) / 64, S * ksl), dim3(256), 0, st, cur, b.Xc, b.dots, k, n, D,
                           (long long)k * D, (long long)n * D, (long long)k * n, ksl, 0);
        SQ_LAUNCH_CHECK();
        hipLaunchKernelGGL(km_center_norms_kernel, dim3(k, S), dim3(64), 0, st, cur, b.shift_part + (size_t)S * k, b.st, D, k, force);
        SQ_LAUNCH_CHECK();
        hipLaunchKernelGGL(km_assign_kernel, dim3((n + 63) / 64, S), dim3(1024), 0, st, b.dots, b.shift_part + (size_t)S * k, labels, b.st, n, D, k, force, ksl);
        SQ_LAUNCH_CHECK();
        hipLaunchKernelGGL(km_members_kernel, dim3(S), dim3(KM_THREADS), 0, st, labels, b.members, b.offsets, b.st, n, k, force);
        SQ_LAUNCH_CHECK();
        return SQ_OK;
    };
    int it = 0;
    int done[2] = {0, 0};            // slides finished / of those, strictly converged (labels unchanged)
    while (it < max_iter) {
        const int burst = it == 0 ? 2 : 4;      // iterations between host checks of the done flags
        for (int q = 0; q < burst && it < max_iter; ++q, ++it) {
            if (int e = e_step(0)) return e;
            hipLaunchKernelGGL(km_sums_kernel, dim3(k, S), dim3(256), 0, st, b.Xc, b.members, b.offsets, b.sums, b.st, n, D, k);
            SQ_LAUNCH_CHECK();
            hipLaunchKernelGGL(km_relocate_kernel, dim3(S), dim3(KM_THREADS), 0, st, b.Xc, cur, labels, b.offsets, b.sums,
                               b.weights, b.dist, b.st, n, D, k);
            SQ_LAUNCH_CHECK();
            hipLaunchKernelGGL(km_update_centers_kernel, dim3(k, S), dim3(256), 0, st, cur, nxt, b.sums, b.weights, b.shift_part, b.st, D, k);
            SQ_LAUNCH_CHECK();
            hipLaunchKernelGGL(km_finish_iter_kernel, dim3(S), dim3(KM_THREADS), 0, st, b.shift_part, labels, b.labels_old, b.st, n, k, max_iter);
            SQ_LAUNCH_CHECK();
            float* t = cur; cur = nxt; nxt = t;
        }
        hipLaunchKernelGGL(km_count_done_kernel, dim3(1), dim3(64), 0, st, b.st, S, b.done_count);
        SQ_LAUNCH_CHECK();
        SQ_HIP_CHECK(hipMemcpyAsync(done, b.done_count, 8, hipMemcpyDeviceToHost, st));
        SQ_HIP_CHECK(hipStreamSynchronize(st));
        if (done[0] == S) break;
    }
    // final E-step for the slides that did not converge strictly (labels must match the final centres).
    // For strictly converged slides the labels of the last E-step are already the answer and the
    // centres moved by a label-preserving update, so re-running the E-step for everyone is only
    // correct for the non-strict ones: the assign kernel is forced, then strict slides are restored.
    // When every slide converged strictly (the common case) nothing is left to do: labels and member lists are
    // those of the last E-step (_kmeans.py: the extra E-step runs only `if not strict_convergence`).
    if (!(done[0] == S && done[1] == S)) {
        // strict slides keep labels (== labels_old after the last update); non-strict get a fresh E-step
        if (int e = e_step(1)) return e;
        // restore labels of strict slides from labels_old and rebuild their member lists
        hipLaunchKernelGGL(km_restore_strict_kernel, dim3((n + 255) / 256, S), dim3(256), 0, st, labels, (const int*)b.labels_old, (const KmState*)b.st, n);
        SQ_LAUNCH_CHECK();
        hipLaunchKernelGGL(km_members_kernel, dim3(S), dim3(KM_THREADS), 0, st, labels, b.members, b.offsets, b.st, n, k, 1);
        SQ_LAUNCH_CHECK();
    }
    if (cluster_features) {
        hipLaunchKernelGGL(km_cluster_means_kernel, dim3(k, S), dim3(256), 0, st, X, b.members, b.offsets, cluster_features, n, D, k);
        SQ_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(km_report_kernel, dim3((S + 63) / 64), dim3(64), 0, st, b.st, S, n_iter);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}
