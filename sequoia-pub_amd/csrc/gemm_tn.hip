// TN GEMM for weight gradients on gfx950:   C[M,N] = alpha * sum_k A[k,m] * B[k,n]
//
// dW = dY^T . X contracts over TOKENS, and both operands are stored token-major (row = token) by the
// forward pass.  Instead of materialising transposed copies, the token-major tiles are DMA'd into LDS
// as they are (rows of 128 outputs = 256 B bf16 / 512 B fp32, fully coalesced) and the MFMA fragments are
// read "down the columns":
//   bf16: v_mfma_f32_16x16x32_bf16 -- lane (i = l&15, g = l>>4) needs k = 8g..8g+7 of column i:
//         two ds_read_b64_tr_b16 (gfx950's transposing LDS read, 4 rows x 16 columns per 16-lane group)
//         per fragment; 16-B chunks XOR-swizzled on the DMA source by 2*((row&3) | ((row>>3)&1)<<2) so the
//         eight 32-byte row pieces one half-wave reads fall in eight different bank slots
//   fp32: v_mfma_f32_16x16x4_f32   -- lane (i, g) needs row g of column i: one conflict-free ds_read_b32
//         (chunks swizzled by (row&3)<<2)
// Block = 128 x 128 outputs, 4 waves (2x2) x 4x4 MFMA tiles, contraction walked 64 (bf16) / 32 (fp32) rows
// per step with two LDS buffers; grid.y slices the contraction (split-K, deterministic reduction).
#include "gemm.h"
#include "gemm_epi.h"

#include <cstdio>

namespace {

constexpr uint32_t OOB = 0x80000000u;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_base, uint32_t voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)lds_base, 16, voffset, soffset, 0, 0);
}

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 lds_bf16x4;

// ds_read_b64_tr_b16: within each 16-lane group, lane L supplies the address of 4 contiguous bf16 and lane i
// receives element (i & 3) of the words supplied by lanes 4j + (i >> 2), j = 0..3 -- a 4 x 16 transpose
__device__ __forceinline__ bf16x4 tr_read(const char* p) {
    const auto v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) lds_bf16x4*)p);
    bf16x4 r;
    r[0] = v[0]; r[1] = v[1]; r[2] = v[2]; r[3] = v[3];
    return r;
}

template <typename T>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const GemmArgs p) {
    constexpr bool LP = sizeof(T) == 2;
    constexpr int BT = 128;                        // outputs per tile side
    constexpr int ROWB = BT * (int)sizeof(T);      // bytes per LDS row (256 / 512)
    constexpr int KR = LP ? 64 : 32;               // contraction rows per step
    constexpr int TILE_BYTES = KR * ROWB;          // 16 KiB per operand
    constexpr int CHUNKS = ROWB / 16;              // 16-B chunks per row (16 / 32)
    constexpr int ROWS_PER_INSTR = 1024 / ROWB;    // rows one wave DMA instruction fills (4 / 2)
    constexpr int NINSTR = KR / (4 * ROWS_PER_INSTR);   // DMA instructions per thread per operand (4 / 4)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // with a bias gradient requested, round_up(tiles_m, 8) extra blocks at the FRONT of the grid sum A's columns
    // instead of multiplying (they are latency-bound, so they must not form the tail; a multiple of 8 keeps
    // block x of the product on XCD x % 8 = its B column tile when N = 1024, i.e. every B tile in one L2)
    const int tiles_n = (p.N + BT - 1) / BT, tiles_m = (p.M + BT - 1) / BT;
    const int cs_blocks = p.colsum_a ? (tiles_m + 7) / 8 * 8 : 0;
    const bool cs_block = (int)blockIdx.x < cs_blocks;
    if (cs_block && (int)blockIdx.x >= tiles_m) return;
    const int bx = (int)blockIdx.x - cs_blocks;
    // tile -> XCD: blocks go round-robin over the 8 XCDs (each with its own L2).  Both operands are long (K rows),
    // so an XCD should own a compact 4 x 2 patch of tiles (it then reads 4 A strips + 2 B strips per 8 tiles) rather
    // than a whole row or column of the tile grid (1 + 8 strips).  Bijective when the patches divide evenly.
    int mt = bx / tiles_n, nt = bx % tiles_n;
    if ((tiles_m & 3) == 0 && (tiles_n & 1) == 0 && (((tiles_m >> 2) * (tiles_n >> 1)) & 7) == 0) {
        const int q = bx & 7, i = bx >> 3;
        const int patch = q + 8 * (i >> 3), j = i & 7;
        const int pn = tiles_n >> 1;
        mt = (patch / pn) * 4 + (j >> 1);
        nt = (patch % pn) * 2 + (j & 1);
    }
    const int m0 = (cs_block ? (int)blockIdx.x : mt) * BT;
    const int n0 = cs_block ? 0 : nt * BT;
    const int z = blockIdx.z;

    // grouped launch: member z brings its own operand / result pointers (same shape and leading dimensions for all)
    const bool grouped = p.ngroup > 0;
    const int zs = grouped ? 0 : z;                 // batch index for the strided form
    const T* Ab = reinterpret_cast<const T*>(grouped ? sq_group_pick(p.gA, z) : p.A) + (long long)zs * p.sA;
    const T* Bb = reinterpret_cast<const T*>(grouped ? sq_group_pick(p.gB, z) : p.B) + (long long)zs * p.sB;
    float* const colsum_dst = grouped ? sq_group_pick(p.gcs, z) : p.colsum_a;
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, (int)(p.a_bytes - (size_t)zs * p.sA * sizeof(T)), 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, (int)(p.b_bytes - (size_t)zs * p.sB * sizeof(T)), 0x00020000);

    // DMA lane mapping: instruction j of wave w fills rows (j*4 + w) * ROWS_PER_INSTR + lane / CHUNKS
    const int lrow = lane / CHUNKS, lchunk = lane % CHUNKS;
    auto key = [](int row) { return LP ? (((row & 3) | (((row >> 3) & 1) << 2)) << 1) : ((row & 3) << 2); };

    const int nk_all = (p.K + KR - 1) / KR;
    const int per = (nk_all + p.splitk - 1) / p.splitk;
    const int kt_lo = blockIdx.y * per, kt_hi = min(nk_all, kt_lo + per);

    auto issue = [&](int kt, int buf) {
        char* sa = smem + buf * (2 * TILE_BYTES);
        char* sb = sa + TILE_BYTES;
#pragma unroll
        for (int j = 0; j < NINSTR; ++j) {
            const int rbase = (j * 4 + wave) * ROWS_PER_INSTR;
            const int row = rbase + lrow;                       // row inside the step
            const int k = kt * KR + row;                        // contraction index
            const int gchunk = lchunk ^ key(row);               // source chunk landing at LDS chunk lchunk
            const int ca = m0 + gchunk * (16 / (int)sizeof(T)), cb = n0 + gchunk * (16 / (int)sizeof(T));
            const uint32_t oa = (k < p.K && ca < p.M) ? (uint32_t)(((long long)k * p.lda + ca) * (long long)sizeof(T)) : OOB;
            const uint32_t ob = (k < p.K && cb < p.N) ? (uint32_t)(((long long)k * p.ldb + cb) * (long long)sizeof(T)) : OOB;
            glds16(rsA, sa + rbase * ROWB, oa, 0);
            glds16(rsB, sb + rbase * ROWB, ob, 0);
        }
    };

    const int li = lane & 15, lg = lane >> 4;
    // bf16 transpose-read addressing: lane (g = lane>>4, j = (lane>>2)&3, c = lane&3) hands the hardware the 8 bytes
    // at row 8g + j, columns 4c..4c+3 of a 16-column block; it receives column lane&15 of the group's four rows
    int tr_a[4], tr_b[4];
    {
        const int j = (lane >> 2) & 3, c = lane & 3;
        const int row = 8 * lg + j, kx = key(row);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int qa = wm * 8 + 2 * t + (c >> 1), qb = wn * 8 + 2 * t + (c >> 1);     // 16-byte chunk of the column
            tr_a[t] = row * ROWB + ((qa ^ kx) << 4) + ((c & 1) << 3);
            tr_b[t] = row * ROWB + ((qb ^ kx) << 4) + ((c & 1) << 3);
        }
    }

    if (cs_block) {
        // colsum_a[m] = sum_k A[k, m]: the A tiles multiplied by a fragment of ones (every column of the 16x16
        // result holds the same sums); waves with wn == 1 only help with the DMA
        auto issue_a = [&](int kt, int buf) {
            char* sa = smem + buf * (2 * TILE_BYTES);
#pragma unroll
            for (int j = 0; j < NINSTR; ++j) {
                const int rbase = (j * 4 + wave) * ROWS_PER_INSTR;
                const int row = rbase + lrow, k = kt * KR + row;
                const int ca = m0 + (lchunk ^ key(row)) * (16 / (int)sizeof(T));
                const uint32_t oa = (k < p.K && ca < p.M) ? (uint32_t)(((long long)k * p.lda + ca) * (long long)sizeof(T)) : OOB;
                glds16(rsA, sa + rbase * ROWB, oa, 0);
            }
        };
        f32x4 cs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) cs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (kt_lo < kt_hi) {
            issue_a(kt_lo, 0);
            __syncthreads();
            for (int kt = kt_lo; kt < kt_hi; ++kt) {
                const int cur = (kt - kt_lo) & 1;
                if (kt + 1 < kt_hi) issue_a(kt + 1, cur ^ 1);
                const char* sa = smem + cur * (2 * TILE_BYTES);
                if (wn == 0) {
                    if constexpr (LP) {
                        bf16x8 ones;
#pragma unroll
                        for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
#pragma unroll
                        for (int s = 0; s < 2; ++s)
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                const bf16x4 a0 = tr_read(sa + s * 32 * ROWB + tr_a[t]), a1 = tr_read(sa + (s * 32 + 4) * ROWB + tr_a[t]);
                                cs[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7), ones, cs[t], 0, 0, 0);
                            }
                    } else {
#pragma unroll
                        for (int s = 0; s < 8; ++s) {
                            const int row = 4 * s + lg, kx = (row & 3) << 2;
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                const int ca = wm * 64 + t * 16 + li;
                                const float fa = *reinterpret_cast<const float*>(sa + row * ROWB + (((ca >> 2) ^ kx) << 4) + ((ca & 3) << 2));
                                cs[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, 1.0f, cs[t], 0, 0, 0);
                            }
                        }
                    }
                }
                __syncthreads();
            }
        }
        if (wn == 0 && li == 0) {      // K-slices go behind the C partials in the split-K scratch
            float* dst = p.splitk > 1 ? p.splitk_ws + (size_t)p.batch * p.splitk * (size_t)p.M * p.N + ((size_t)z * p.splitk + blockIdx.y) * p.M : colsum_dst;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * 64 + i * 16 + lg * 4 + r;
                    if (m < p.M) dst[m] = cs[i][r];
                }
        }
        return;
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) {
        const char* sa = smem + buf * (2 * TILE_BYTES);
        const char* sb = sa + TILE_BYTES;
        if constexpr (LP) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {                       // two k-steps of 32 rows
                bf16x8 fa[4], fb[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    // rows 32s + 8g + 4h + j of column block t: h = 0 / 1 are the low / high half of the fragment
                    const bf16x4 a0 = tr_read(sa + s * 32 * ROWB + tr_a[t]), a1 = tr_read(sa + (s * 32 + 4) * ROWB + tr_a[t]);
                    const bf16x4 b0 = tr_read(sb + s * 32 * ROWB + tr_b[t]), b1 = tr_read(sb + (s * 32 + 4) * ROWB + tr_b[t]);
                    fa[t] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                    fb[t] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 8; ++s) {                       // eight k-steps of 4 rows
                const int row = 4 * s + lg;
                const int kx = (row & 3) << 2;                  // = lg << 2
                float fa[4], fb[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int ca = wm * 64 + t * 16 + li, cb = wn * 64 + t * 16 + li;
                    fa[t] = *reinterpret_cast<const float*>(sa + row * ROWB + (((ca >> 2) ^ kx) << 4) + ((ca & 3) << 2));
                    fb[t] = *reinterpret_cast<const float*>(sb + row * ROWB + (((cb >> 2) ^ kx) << 4) + ((cb & 3) << 2));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        }
    };

    if (kt_lo < kt_hi) {
        issue(kt_lo, 0);
        __syncthreads();
        for (int kt = kt_lo; kt < kt_hi; ++kt) {
            const int cur = (kt - kt_lo) & 1;
            if (kt + 1 < kt_hi) issue(kt + 1, cur ^ 1);
            compute(cur);
            __syncthreads();
        }
    }

    // epilogue: 16x16 MFMA C/D layout: col = lane&15, row = (lane>>4)*4 + reg.  Stage through LDS (64 KiB).
    float* stage = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * 64 + i * 16 + lg * 4 + r;
                const int col = wn * 64 + j * 16 + li;
                stage[row * BT + col] = acc[i][j][r];
            }
    __syncthreads();
    const int c8 = tid & 15, rbase = tid >> 4;
    const int n = n0 + c8 * 8;
    if (n >= p.N) return;
    if (p.splitk > 1) {
        float* part = p.splitk_ws + ((size_t)z * p.splitk + blockIdx.y) * (size_t)p.M * p.N;
#pragma unroll 1
        for (int u = 0; u < 8; ++u) {
            const int row = rbase + u * 16, m = m0 + row;
            if (m < p.M) {
                *reinterpret_cast<f32x4*>(part + (size_t)m * p.N + n) = *reinterpret_cast<const f32x4*>(stage + row * BT + c8 * 8);
                *reinterpret_cast<f32x4*>(part + (size_t)m * p.N + n + 4) = *reinterpret_cast<const f32x4*>(stage + row * BT + c8 * 8 + 4);
            }
        }
        return;
    }
    const bool vec = p.vec_epi != 0;
    const int cnt = min(8, p.N - n);
#pragma unroll 1
    for (int u = 0; u < 8; ++u) {
        const int row = rbase + u * 16, m = m0 + row;
        if (m >= p.M) continue;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(stage + row * BT + c8 * 8);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(stage + row * BT + c8 * 8 + 4);
        float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        epi_apply<3>(p, zs, m, n, v, cnt, vec, nullptr, nullptr, grouped ? sq_group_pick(p.gC, z) : nullptr);
    }
}

}  // namespace

int g_tn_force_split = 0;      // experiment knob (sq_dbg_set key 4)

int sq_launch_gemm_tn(const GemmArgs& a_in, int dtype, hipStream_t stream) {
    GemmArgs a = a_in;
    const int epc = dtype == SQ_BF16 ? 8 : 4;
    SQ_REQUIRE(dtype == SQ_BF16 || dtype == SQ_F32, "gemm_tn: dtype %d", dtype);
    SQ_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.batch > 0, "gemm_tn: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
    // M may be ragged: A columns in [M, round_up(M, 8)) are read (they must exist: lda >= round_up(M, 8)) but
    // the corresponding output rows are never written
    SQ_REQUIRE(a.N % 8 == 0 && a.lda >= (a.M + 7) / 8 * 8, "gemm_tn: N=%d must be a multiple of 8 and lda=%d >= round_up(M=%d, 8)", a.N, a.lda, a.M);
    SQ_REQUIRE(a.lda % epc == 0 && a.ldb % epc == 0 && ((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.B & 15) == 0 &&
               (a.sA % epc) == 0 && (a.sB % epc) == 0, "gemm_tn: operands must be 16-byte aligned (lda=%d ldb=%d)", a.lda, a.ldb);
    SQ_REQUIRE(a.a_bytes > 0 && a.a_bytes < (1ull << 31) && a.b_bytes > 0 && a.b_bytes < (1ull << 31), "gemm_tn: operand extents must be < 2 GiB");
    SQ_REQUIRE(!a.conv, "gemm_tn: no convolution view");
    if (a.ngroup) {
        SQ_REQUIRE(a.ngroup >= 1 && a.ngroup <= 4 && a.batch == 1, "gemm_tn: a group has 1..4 members and no batch (got %d, batch %d)", a.ngroup, a.batch);
        bool any_cs = false, all_cs = true;
        for (int i = 0; i < a.ngroup; ++i) {
            SQ_REQUIRE(a.gA[i] && a.gB[i] && a.gC[i] && ((uintptr_t)a.gA[i] & 15) == 0 && ((uintptr_t)a.gB[i] & 15) == 0 && ((uintptr_t)a.gC[i] & 15) == 0,
                       "gemm_tn: group member %d: null or misaligned pointer", i);
            any_cs = any_cs || a.gcs[i]; all_cs = all_cs && a.gcs[i];
        }
        SQ_REQUIRE(any_cs == all_cs, "gemm_tn: bias-gradient outputs must be given for every group member or for none");
        a.A = a.gA[0]; a.B = a.gB[0]; a.C = a.gC[0]; a.colsum_a = a.gcs[0];
        a.batch = a.ngroup;
    }
    SQ_REQUIRE(!a.colsum_a || a.batch == 1 || a.ngroup, "gemm_tn: colsum_a needs batch == 1");
    {
        auto al = [](const void* ptr, int ld, long long st, int elem) {
            return ptr == nullptr || (((uintptr_t)ptr % 16) == 0 && (ld * elem) % 16 == 0 && ((st * elem) % 16) == 0);
        };
        bool ok = al(a.bias, 4, a.sBias, 4) && al(a.rowbias, a.ldrb, a.sRb, 4) && al(a.Cpre, a.ldpre, a.sPre, a.pre_dtype == SQ_F32 ? 4 : 2) &&
                  al(a.gelu_grad_of, a.ldgg, a.sGg, a.gg_dtype == SQ_F32 ? 4 : 2) && al(a.C2, a.ldc2, a.sC2, 2);
        ok = ok && al(a.C, a.ldc, a.sC, a.out_dtype == SQ_F32 ? 4 : 2) && al(a.res, a.ldres, a.sRes, a.res_dtype == SQ_F32 ? 4 : 2);
        a.vec_epi = ok ? 1 : 0;
    }
    const int kr = dtype == SQ_BF16 ? 64 : 32;
    const int nk = (a.K + kr - 1) / kr;
    const long long tiles = (long long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    a.splitk = 1;
    const long long cs_blocks = a.colsum_a ? ((a.M + 127) / 128 + 7) / 8 * 8 : 0;
    if (a.splitk_ws && tiles * a.batch < 256 && nk >= 4) {
        // 2 blocks (64 KiB LDS each) per CU x 256 CUs: the whole grid should be resident at once
        long long s = 512 / ((tiles + cs_blocks) * a.batch);
        if (s < 1) s = 1;
        if (s > nk / 2) s = nk / 2;
        if (s > 32) s = 32;
        while (s > 1 && (size_t)s * ((size_t)a.M * a.N * a.batch + (a.colsum_a ? (size_t)a.M * a.batch : 0)) * sizeof(float) > a.splitk_ws_bytes) --s;
        if (s > 1) a.splitk = (int)s;
    }
    if (g_tn_force_split > 0 && a.splitk_ws && (size_t)g_tn_force_split * ((size_t)a.M * a.N * a.batch + (size_t)a.M * a.batch) * 4 <= a.splitk_ws_bytes) a.splitk = g_tn_force_split;
    int prof = -1;
    if (sq_prof_on()) {
        const double es = dtype == SQ_BF16 ? 2.0 : 4.0;
        char name[96];
        snprintf(name, sizeof(name), "gemmtn_%s_M%d_N%d_K%d_b%d", dtype == SQ_BF16 ? "bf16" : "f32", a.M, a.N, a.K, a.batch);
        prof = sq_prof_begin(name, 2.0 * a.M * (double)a.N * a.K * a.batch,
                             ((double)a.K * (a.M + a.N) * es + (double)a.M * a.N * 4.0) * a.batch, stream);
    }
    const long long grid_x = tiles + cs_blocks;
    dim3 grid((unsigned)grid_x, a.splitk, a.batch), block(256);
    if (dtype == SQ_BF16) hipLaunchKernelGGL(gemm_tn_kernel<bf16_t>, grid, block, 65536, stream, a);
    else hipLaunchKernelGGL(gemm_tn_kernel<float>, grid, block, 65536, stream, a);
    SQ_LAUNCH_CHECK();
    int rc = SQ_OK;
    if (a.splitk > 1) rc = sq_launch_splitk_reduce(a, stream);
    if (prof >= 0) sq_prof_end(prof, stream);
    return rc;
}
