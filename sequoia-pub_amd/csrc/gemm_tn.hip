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
#include <cstdlib>

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

// the same read as inline assembly (ring form below): LDS byte address + immediate offset
template <int OFF>
__device__ __forceinline__ u32x2 tr_read_asm(uint32_t addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
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


// ---------------------------------------------------------------------------------------------------------------------
// Ring form (bf16, both output extents multiples of 128): what the large weight gradients of a training step run on.
//
// The two-buffer kernel above keeps ONE K-step in flight: with a single 4-wave block per CU (256 tiles of a grouped
// launch = one per CU) every step exposes the full L2 / HBM round trip in front of its 32 MFMAs -- 1.0 us per 64-row step,
// 135 us for the four 1024 x 1024 x 6400 gradients of a layer (398 TF).  Here:
//   * eight waves per 128 x 128 tile: waves 0-3 and 4-7 each hold the whole 2 x 2 arrangement of 64 x 64 patches and take
//     rows 0-31 / 32-63 of every 64-row step (an in-block K split: two waves per SIMD, one's transposing LDS reads run under
//     the other's MFMAs; the two accumulator sets are added through LDS in the epilogue, fixed order);
//   * FOUR stages of 32 KiB (A rows + B rows, token-major as stored), three steps in flight, one counted
//     `s_waitcnt vmcnt(8)` per step; steps beyond the slice's end are issued out of range (the buffer descriptor returns
//     zeros), so the count is the same in every iteration; ONE raw `s_barrier` per step, the two wave sets issue their LDS-DMA
//     requests at opposite ends of the step, fragment reads run a step ahead of the MFMAs (see "Schedule" in the kernel);
//   * tile walk: XCD x (blocks x, x + 8, ...) owns a contiguous run of the work list, and the list is ordered in 8 x 4
//     patches of tiles -- the 32 blocks resident on an XCD read 8 A strips + 4 B strips (12 x 1.6 MB for K = 6400) instead
//     of 4 x (4 + 2) = 24;
//   * the bias gradient (column sums of A) rides in the tiles themselves: the 128 columns of a tile row are eight 16-wide
//     fragment blocks, block s is summed by tile column s % tiles_n with one extra MFMA (fragment x ones) per step in two of
//     its eight waves -- no extra blocks in the grid.
constexpr int RG_BT = 128, RG_ROWB = 256, RG_KR = 64, RG_TILE = RG_KR * RG_ROWB, RG_STAGE = 2 * RG_TILE, RG_NST = 4;
constexpr int RG_SLD = RG_BT + 4;                         // fp32 stage row pitch (floats)
constexpr int RG_LDS = RG_NST * RG_STAGE;                 // 128 KiB (the epilogue's 128 x 132 fp32 stage + column sums fit inside)

__global__ __launch_bounds__(512) void gemm_tn_ring_kernel(const GemmArgs p, const int tiles_m, const int tiles_n, const int total) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;

    // work item of this block: XCD-contiguous runs of the list [member][slice][patch of 8 x 4 tiles][tile]
    const int per = (total + 7) >> 3;
    const int L = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    if (L >= total) return;
    const int tpm = tiles_m * tiles_n;
    const int zs = L / tpm, r = L - zs * tpm;
    const int z = zs / p.splitk, slice = zs - z * p.splitk;
    int mt, nt;
    if ((tiles_m & 7) == 0 && (tiles_n & 3) == 0) {
        const int patch = r >> 5, q = r & 31, pn = tiles_n >> 2;
        mt = (patch / pn) * 8 + (q >> 2);
        nt = (patch % pn) * 4 + (q & 3);
    } else {
        mt = r / tiles_n;
        nt = r - mt * tiles_n;
    }
    const int m0 = mt * RG_BT, n0 = nt * RG_BT;

    const bool grouped = p.ngroup > 0;
    const bf16_t* Ab = reinterpret_cast<const bf16_t*>(grouped ? sq_group_pick(p.gA, z) : p.A);
    const bf16_t* Bb = reinterpret_cast<const bf16_t*>(grouped ? sq_group_pick(p.gB, z) : p.B);
    float* const colsum_dst = grouped ? sq_group_pick(p.gcs, z) : p.colsum_a;
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, (int)p.a_bytes, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, (int)p.b_bytes, 0x00020000);

    const int nk_all = (p.K + RG_KR - 1) / RG_KR;
    const int per_k = (nk_all + p.splitk - 1) / p.splitk;
    const int kt_lo = slice * per_k, kt_hi = min(nk_all, kt_lo + per_k);
    const int nk = max(kt_hi - kt_lo, 0);

    // DMA lane mapping: a wave instruction fills 4 rows of 256 B; instruction j of wave w -> rows (8j + w) * 4 ..+3
    const int lrow = lane >> 4, lchunk = lane & 15;
    auto key = [](int row) { return ((row & 3) | (((row >> 3) & 1) << 2)) << 1; };
    uint32_t src_a[2], src_b[2];      // byte offset of this lane's 16 bytes inside a step (row part), per instruction
    int row_j[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (j * 8 + wave) * 4 + lrow;
        const int gchunk = lchunk ^ key(row);
        row_j[j] = row;
        src_a[j] = (uint32_t)(((long long)row * p.lda + m0 + gchunk * 8) * 2);
        src_b[j] = (uint32_t)(((long long)row * p.ldb + n0 + gchunk * 8) * 2);
    }
    const uint32_t step_a = (uint32_t)p.lda * RG_KR * 2, step_b = (uint32_t)p.ldb * RG_KR * 2;
    auto issue = [&](int kt, int buf) {
        char* sa = smem + buf * RG_STAGE;
        char* sb = sa + RG_TILE;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool ok = kt < kt_hi && kt * RG_KR + row_j[j] < p.K;
            const uint32_t oa = ok ? src_a[j] + (uint32_t)kt * step_a : OOB;
            const uint32_t ob = ok ? src_b[j] + (uint32_t)kt * step_b : OOB;
            glds16(rsA, sa + (j * 8 + wave) * 4 * RG_ROWB, oa, 0);
            glds16(rsB, sb + (j * 8 + wave) * 4 * RG_ROWB, ob, 0);
        }
    };

    // transposing fragment reads (as in the kernel above); this wave's rows of a step start at 32 * kg
    const int li = lane & 15, lg = lane >> 4;
    int tr_a[4], tr_b[4];
    {
        const int j = (lane >> 2) & 3, c = lane & 3;
        const int row = 8 * lg + j, kx = key(row);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int qa = wm * 8 + 2 * t + (c >> 1), qb = wn * 8 + 2 * t + (c >> 1);
            tr_a[t] = (kg * 32 + row) * RG_ROWB + ((qa ^ kx) << 4) + ((c & 1) << 3);
            tr_b[t] = (kg * 32 + row) * RG_ROWB + ((qb ^ kx) << 4) + ((c & 1) << 3);
        }
    }
    // column sums: fragment block s = wm * 4 + t of this tile row belongs to tile column s % tiles_n
    int csmask = 0;
    if (colsum_dst && wn == 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if ((wm * 4 + t) % tiles_n == nt) csmask |= 1 << t;
    }
    csmask = __builtin_amdgcn_readfirstlane(csmask);

    f32x4 acc[4][4], cs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        cs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

    // The fragment reads are inline assembly: hipcc treats the transposing-read BUILTIN as possibly aliasing the LDS-DMA
    // requests in flight and puts `s_waitcnt vmcnt(0)` in front of it -- which would drain the ring every step.
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    // Schedule.  Per step a CU stages 32 KiB through its vector-memory path (64 B/clk: 512 cycles -- and a wave that is issuing
    // LDS-DMA requests issues nothing else meanwhile), reads 64 KiB of fragments (512 cycles at the 128 B/clk two waves per SIMD get
    // out of `ds_read_b64_tr_b16`; no bank conflicts by the counters) and runs 512 cycles of MFMAs
    // per SIMD; a block-wide barrier costs ~165 cycles.  Measured with ablation switches (tools/tn_probe.py abl): when every wave
    // does request -> read -> multiply in order the three costs ADD (1560 cycles per step); with two barriers per step and the
    // wave sets half a step apart a phase is max(request + read, multiply) + barrier (1240 cycles).  What runs: ONE barrier per
    // step; behind it waves 0-3 issue their requests FIRST and waves 4-7 LAST, so that on every SIMD (waves w and w + 4 share one)
    // one wave sits in the memory path while the other owns the matrix pipe; the fragment reads of step `it` are issued ahead of
    // the MFMAs of step it-1 (two register sets) and waited for behind them, when they have long returned.
    typedef u32x2 frag_set[4][2];
    frag_set ra0, rb0, ra1, rb1;       // fragments of two consecutive steps (set = step parity; separate objects: no run-time indexing)
    const int dbg = p.dbg;             // ablation switches (tools/tn_probe.py abl): 2 no requests after the prologue, 4 no MFMA
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) ra0[t][h] = rb0[t][h] = ra1[t][h] = rb1[t][h] = u32x2{0u, 0u};
    auto read_frags = [&](int buf, frag_set& ra, frag_set& rb) {
        const uint32_t sbase = lds0 + (uint32_t)buf * RG_STAGE;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ra[t][0] = tr_read_asm<0>(sbase + tr_a[t]); ra[t][1] = tr_read_asm<4 * RG_ROWB>(sbase + tr_a[t]);
            rb[t][0] = tr_read_asm<RG_TILE>(sbase + tr_b[t]); rb[t][1] = tr_read_asm<RG_TILE + 4 * RG_ROWB>(sbase + tr_b[t]);
        }
    };
    auto frags_landed = [&](frag_set& ra, frag_set& rb) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[1][0]), "+v"(ra[1][1]), "+v"(ra[2][0]), "+v"(ra[2][1]), "+v"(ra[3][0]), "+v"(ra[3][1]) :: "memory");
        asm volatile("" : "+v"(rb[0][0]), "+v"(rb[0][1]), "+v"(rb[1][0]), "+v"(rb[1][1]), "+v"(rb[2][0]), "+v"(rb[2][1]), "+v"(rb[3][0]), "+v"(rb[3][1]) :: "memory");
    };
    auto multiply = [&](const frag_set& ra, const frag_set& rb) {
        bf16x8 fa[4], fb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const u32x4 ua = {ra[t][0][0], ra[t][0][1], ra[t][1][0], ra[t][1][1]}, ub = {rb[t][0][0], rb[t][0][1], rb[t][1][0], rb[t][1][1]};
            fa[t] = __builtin_bit_cast(bf16x8, ua);
            fb[t] = __builtin_bit_cast(bf16x8, ub);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        if (csmask) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (csmask & (1 << t)) cs[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[t], ones, cs[t], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto request = [&](int it) {
        if (!(dbg & 2)) issue(kt_lo + it + RG_NST - 1, (it + RG_NST - 1) & (RG_NST - 1));
    };
    // one step: its fragments go to (ra, rb); the previous step's are in (pa, pb)
    auto step = [&](int it, frag_set& ra, frag_set& rb, const frag_set& pa, const frag_set& pb) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // this wave's requests of step `it` have landed (two later steps x 4 may be out)
        __builtin_amdgcn_s_barrier();                           // everybody's; and step it-1's buffer has been read by every wave
        if (kg == 0) request(it);                               // waves 0-3 sit in the memory path first ...
        read_frags(it & (RG_NST - 1), ra, rb);
        __builtin_amdgcn_sched_barrier(0);
        if (it > 0 && !(dbg & 4)) multiply(pa, pb);
        __builtin_amdgcn_sched_barrier(0);
        frags_landed(ra, rb);                                   // (in front of the next barrier: behind it the buffer may be refilled)
        if (kg != 0) request(it);                               // ... waves 4-7 last, under the others' MFMAs
    };

#pragma unroll
    for (int s = 0; s < RG_NST - 1; ++s) issue(kt_lo + s, s);
    for (int it = 0; it < nk; it += 2) {
        step(it, ra0, rb0, ra1, rb1);
        if (it + 1 < nk) step(it + 1, ra1, rb1, ra0, rb0);
    }
    if (nk > 0 && !(dbg & 4)) {
        if ((nk - 1) & 1) multiply(ra1, rb1); else multiply(ra0, rb0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (out-of-range requests still write zeros: none may land in the stage below)
    __syncthreads();

    // ---- epilogue: waves 4-7 park their sums in LDS, waves 0-3 add theirs on top (16x16 C/D layout: col = lane&15, row = 4*(lane>>4) + reg)
    float* stage = reinterpret_cast<float*>(smem);
    float* cstage = stage + RG_BT * RG_SLD;                     // [128] column sums of the other K half
    if (kg == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) stage[(wm * 64 + i * 16 + lg * 4 + rr) * RG_SLD + wn * 64 + j * 16 + li] = acc[i][j][rr];
        if (csmask && li == 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (csmask & (1 << t))
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) cstage[wm * 64 + t * 16 + lg * 4 + rr] = cs[t][rr];
        }
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    float* q = stage + (wm * 64 + i * 16 + lg * 4 + rr) * RG_SLD + wn * 64 + j * 16 + li;
                    *q = acc[i][j][rr] + *q;
                }
        if (csmask && li == 0) {
            float* dst = p.splitk > 1 ? p.splitk_ws + (size_t)p.batch * p.splitk * (size_t)p.M * p.N + ((size_t)z * p.splitk + slice) * p.M : colsum_dst;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (csmask & (1 << t))
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int ml = wm * 64 + t * 16 + lg * 4 + rr;
                        dst[m0 + ml] = cs[t][rr] + cstage[ml];
                    }
        }
    }
    __syncthreads();
    const int c8 = tid & 15, rbase = tid >> 4;                  // 32 rows x 16 chunks of 8 columns per pass
    const int n = n0 + c8 * 8;
    if (p.splitk > 1) {
        float* part = p.splitk_ws + ((size_t)z * p.splitk + slice) * (size_t)p.M * p.N;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = rbase + u * 32, m = m0 + row;
            *reinterpret_cast<f32x4*>(part + (size_t)m * p.N + n) = *reinterpret_cast<const f32x4*>(stage + row * RG_SLD + c8 * 8);
            *reinterpret_cast<f32x4*>(part + (size_t)m * p.N + n + 4) = *reinterpret_cast<const f32x4*>(stage + row * RG_SLD + c8 * 8 + 4);
        }
        return;
    }
    const bool vec = p.vec_epi != 0;
    void* const c_ovr = grouped ? sq_group_pick(p.gC, z) : nullptr;
#pragma unroll 1
    for (int u = 0; u < 4; ++u) {
        const int row = rbase + u * 32, m = m0 + row;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(stage + row * RG_SLD + c8 * 8);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(stage + row * RG_SLD + c8 * 8 + 4);
        float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        epi_apply<3>(p, 0, m, n, v, 8, vec, nullptr, nullptr, c_ovr);
    }
}

}  // namespace

int g_tn_force_split = 0;      // experiment knob (sq_dbg_set key 4)
extern int g_dbg;               // sq_dbg_set key 1 (gemm.hip)
int g_tn_ring = 1;             // sq_dbg_set key 15 (tests / probes): 0 turns the ring form off

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
    // large bf16 gradients: the four-stage ring form (one 8-wave block per CU); sq_dbg_set(15, 0) turns it off
    if (g_tn_ring && dtype == SQ_BF16 && a.M % 128 == 0 && a.N % 128 == 0 && (a.batch == 1 || a.ngroup) && nk >= 8 &&
        tiles * a.batch >= 8) {
        const long long work = tiles * a.batch;
        const size_t per_slice = ((size_t)a.M * a.N * a.batch + (a.colsum_a ? (size_t)a.M * a.batch : 0)) * sizeof(float);
        if (a.splitk_ws && work < 256) {
            long long s = 256 / work;                          // one round of blocks, one block per CU
            if (s > nk / 8) s = nk / 8;
            while (s > 1 && (size_t)s * per_slice > a.splitk_ws_bytes) --s;
            if (s > 1) a.splitk = (int)s;
        }
        if (g_tn_force_split > 0 && a.splitk_ws && (size_t)g_tn_force_split * per_slice <= a.splitk_ws_bytes) a.splitk = g_tn_force_split;
        static SqDevOnce attr;
        if (attr.needed()) {
            SQ_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_ring_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RG_LDS));
            attr.done();
        }
        int prof = -1;
        if (sq_prof_on()) {
            char name[96];
            snprintf(name, sizeof(name), "gemmtn_bf16_M%d_N%d_K%d_b%d", a.M, a.N, a.K, a.batch);
            prof = sq_prof_begin(name, 2.0 * a.M * (double)a.N * a.K * a.batch, ((double)a.K * (a.M + a.N) * 2.0 + (double)a.M * a.N * 4.0) * a.batch, stream);
        }
        const int total = (int)(work * a.splitk);
        a.dbg |= g_dbg;
        const dim3 grid((unsigned)((total + 7) / 8 * 8)), block(512);
        hipLaunchKernelGGL(gemm_tn_ring_kernel, grid, block, RG_LDS, stream, a, a.M / 128, a.N / 128, total);
        SQ_LAUNCH_CHECK();
        int rc = SQ_OK;
        if (a.splitk > 1) rc = sq_launch_splitk_reduce(a, stream);
        if (prof >= 0) sq_prof_end(prof, stream);
        return rc;
    }
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
