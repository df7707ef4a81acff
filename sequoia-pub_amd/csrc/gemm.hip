// MFMA NT GEMM / implicit-GEMM convolution for gfx950 (MI355X).
//
// Tile: 256 threads = 4 waves (2x2); each wave owns WTM x WTN MFMA tiles of 32x32.
// K is walked in tiles of 128 bytes per row (64 bf16 / 32 fp32), staged
// global -> VGPR -> LDS (register prefetch of tile t+1 under the MFMAs of tile t, two
// LDS buffers, one barrier per K-tile).  LDS rows are 128 B, 16-B chunks XOR-swizzled
// by (row>>1)&7 so every ds_read_b128 lane group touches 16 distinct 16-B slots.
// Loads go through buffer descriptors: rows past M/N, K tails and convolution padding
// are redirected to an out-of-range offset and read back as zero.
//
// fp32 path: v_mfma_f32_32x32x2_f32 (exact fp32 fma chain) -- the parity mode.
// bf16 path: v_mfma_f32_32x32x16_bf16, fp32 accumulate -- the perf mode.
#include "gemm.h"
#include <cstdio>

namespace {

constexpr uint32_t OOB = 0x80000000u;

__device__ __forceinline__ u32x4 lds_read128(const char* p) { return *reinterpret_cast<const u32x4*>(p); }

template <typename T> struct Mma;

template <> struct Mma<float> {
    // chunk = 4 floats: element e of lane-half h is k = 4*(2s+h)+e; A and B use the same
    // assignment, so the contraction is a permutation of the K-tile.
    static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& acc) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[e]), __uint_as_float(b[e]), acc, 0, 0, 0);
    }
};

template <> struct Mma<bf16_t> {
    static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& acc) {
        union { u32x4 u; bf16x8 h; } ua, ub;
        ua.u = a; ub.u = b;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.h, ub.h, acc, 0, 0, 0);
    }
};

template <typename T, int WTM, int WTN, bool CONV>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const GemmArgs p) {
    constexpr int BM = 64 * WTM, BN = 64 * WTN;
    constexpr int EPC = 16 / (int)sizeof(T);   // elements per 16-byte chunk
    constexpr int BK = 8 * EPC;                // K elements per tile (128 B)
    constexpr int RA = BM / 32, RB = BN / 32;  // chunks per thread per tile
    constexpr int TILE_BYTES = (BM + BN) * 128;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware bijective remap: each XCD (block id % 8) walks a contiguous run of tiles
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int nwg = tiles_m * tiles_n;
    int t;
    {
        const int b = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = b & 7, idx = b >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (t / tiles_n) * BM;
    const int n0 = (t % tiles_n) * BN;
    const int z = blockIdx.z;

    const T* Ab = reinterpret_cast<const T*>(p.A) + (long long)z * p.sA;
    const T* Bb = reinterpret_cast<const T*>(p.B) + (long long)z * p.sB;
    const size_t a_rem = p.a_bytes - (size_t)z * p.sA * sizeof(T);
    const size_t b_rem = p.b_bytes - (size_t)z * p.sB * sizeof(T);
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, (int)a_rem, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, (int)b_rem, 0x00020000);

    const int c16 = tid & 7;      // 16-B chunk inside the 128-B K-tile row
    const int r0 = tid >> 3;      // row inside a 32-row group

    // per-thread row bookkeeping.  Plain GEMM: the byte offset of each row's chunk is loop-invariant
    // (the K-tile offset travels in the scalar soffset operand), so the K loop issues its loads
    // back-to-back with no address arithmetic.  Conv: one offset per (row, tap).
    uint32_t a_off[RA], b_off[RB];      // byte offsets at k = 0 (OOB when the row is out of range)
    int a_ih0[RA], a_iw0[RA];
    uint32_t a_pix[RA];
    bool a_ok[RA];
#pragma unroll
    for (int j = 0; j < RA; ++j) {
        const int m = m0 + r0 + 32 * j;
        a_ok[j] = m < p.M;
        if constexpr (CONV) {
            const int ohw = p.OH * p.OW;
            const int img = m / ohw;
            const int rem = m - img * ohw;
            const int oh = rem / p.OW, ow = rem - oh * p.OW;
            a_ih0[j] = oh * p.stride - p.pad;
            a_iw0[j] = ow * p.stride - p.pad;
            a_pix[j] = (uint32_t)(img * p.H * p.W);
            a_off[j] = 0;
        } else {
            a_ih0[j] = a_iw0[j] = 0;
            a_pix[j] = 0;
            a_off[j] = a_ok[j] ? ((uint32_t)m * (uint32_t)p.lda + (uint32_t)(c16 * EPC)) * (uint32_t)sizeof(T) : OOB;
        }
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        const int n = n0 + r0 + 32 * j;
        b_off[j] = n < p.N ? ((uint32_t)n * (uint32_t)p.ldb + (uint32_t)(c16 * EPC)) * (uint32_t)sizeof(T) : OOB;
    }

    u32x4 ra[RA], rb[RB];

    auto issue_loads = [&](int kt) {
        const int k0 = kt * BK;
        const bool k_ok = k0 + c16 * EPC < p.K;          // only false in a ragged last K-tile
        uint32_t oa[RA], ob[RB];
        int soff;
        if constexpr (CONV) {
            const int tap = k0 / p.Cin;                  // a K-tile never straddles taps (Cin % BK == 0)
            const int cin0 = k0 - tap * p.Cin + c16 * EPC;
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
            for (int j = 0; j < RA; ++j) {
                const int ih = a_ih0[j] + kh, iw = a_iw0[j] + kw;
                const bool ok = a_ok[j] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                const uint32_t off = ((a_pix[j] + (uint32_t)(ih * p.W + iw)) * (uint32_t)p.Cin + (uint32_t)cin0) * (uint32_t)sizeof(T);
                oa[j] = ok ? off : OOB;
            }
            soff = 0;
#pragma unroll
            for (int j = 0; j < RB; ++j) ob[j] = k_ok ? b_off[j] + (uint32_t)(k0 * (int)sizeof(T)) : OOB;
        } else {
            soff = k0 * (int)sizeof(T);
#pragma unroll
            for (int j = 0; j < RA; ++j) oa[j] = k_ok ? a_off[j] : OOB;
#pragma unroll
            for (int j = 0; j < RB; ++j) ob[j] = k_ok ? b_off[j] : OOB;
        }
#pragma unroll
        for (int j = 0; j < RA; ++j) ra[j] = __builtin_amdgcn_raw_buffer_load_b128(rsA, oa[j], soff, 0);
#pragma unroll
        for (int j = 0; j < RB; ++j) rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rsB, ob[j], soff, 0);
    };

    auto store_lds = [&](int buf) {
        char* sa = smem + buf * TILE_BYTES;
        char* sb = sa + BM * 128;
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const int row = r0 + 32 * j;
            *reinterpret_cast<u32x4*>(sa + row * 128 + ((c16 ^ ((row >> 1) & 7)) << 4)) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            const int row = r0 + 32 * j;
            *reinterpret_cast<u32x4*>(sb + row * 128 + ((c16 ^ ((row >> 1) & 7)) << 4)) = rb[j];
        }
    };

    f32x16 acc[WTM][WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;

    auto compute = [&](int buf) {
        const char* sa = smem + buf * TILE_BYTES;
        const char* sb = sa + BM * 128;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int chunk = 2 * s + lh;
            u32x4 fa[WTM], fb[WTN];
#pragma unroll
            for (int i = 0; i < WTM; ++i) {
                const int row = wm * (WTM * 32) + i * 32 + l31;
                fa[i] = lds_read128(sa + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < WTN; ++j) {
                const int row = wn * (WTN * 32) + j * 32 + l31;
                fb[j] = lds_read128(sb + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j) Mma<T>::run(fa[i], fb[j], acc[i][j]);
        }
    };

    const int nk = (p.K + BK - 1) / BK;
    issue_loads(0);
    store_lds(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more && !(p.dbg & 2)) issue_loads(kt + 1);
        if (!(p.dbg & 4)) compute(kt & 1);
        if (more) store_lds((kt + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue ---------------------------------------------------------------------------
    // Accumulators go through LDS (the K-loop buffers are free after the last barrier) so that the
    // epilogue is a short rolled loop of 16-byte row-major accesses instead of 64 unrolled scalar
    // stores per lane.  C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    float* stage = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * (WTM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int col = wn * (WTN * 32) + j * 32 + l31;
                stage[row * BN + col] = acc[i][j][r];
            }
    __syncthreads();

    const float* bias = p.bias ? p.bias + (long long)z * p.sBias : nullptr;
    const float* rowbias = p.rowbias ? p.rowbias + (long long)z * p.sRb : nullptr;
    const float* res32 = (p.res && p.res_dtype == SQ_F32) ? reinterpret_cast<const float*>(p.res) + (long long)z * p.sRes : nullptr;
    const bf16_t* res16 = (p.res && p.res_dtype == SQ_BF16) ? reinterpret_cast<const bf16_t*>(p.res) + (long long)z * p.sRes : nullptr;
    float* c32 = p.out_dtype == SQ_F32 ? reinterpret_cast<float*>(p.C) + (long long)z * p.sC : nullptr;
    bf16_t* c16p = p.out_dtype == SQ_BF16 ? reinterpret_cast<bf16_t*>(p.C) + (long long)z * p.sC : nullptr;
    bf16_t* c2 = p.C2 ? p.C2 + (long long)z * p.sC2 : nullptr;
    float* cpre = p.Cpre ? p.Cpre + (long long)z * p.sPre : nullptr;
    const float* gg = p.gelu_grad_of ? p.gelu_grad_of + (long long)z * p.sGg : nullptr;
    if (p.dbg & 1) return;

    constexpr int BN4 = BN / 4;
    const bool vec = p.vec_epi != 0;
#pragma unroll 1
    for (int idx = tid; idx < BM * BN4; idx += 256) {
        const int row = idx / BN4, c4 = idx - row * BN4;
        const int m = m0 + row, n = n0 + c4 * 4;
        if (m >= p.M || n >= p.N) continue;
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(stage + row * BN + c4 * 4);
        float v[4] = {p.alpha * a4[0], p.alpha * a4[1], p.alpha * a4[2], p.alpha * a4[3]};
        const int cnt = min(4, p.N - n);
        const long long rb_row = rowbias ? (long long)(m / p.rows_per_group) * p.ldrb : 0;
        if (vec && cnt == 4) {
            if (bias) { const f32x4 t = *reinterpret_cast<const f32x4*>(bias + n); v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3]; }
            if (rowbias) { const f32x4 t = *reinterpret_cast<const f32x4*>(rowbias + rb_row + n); v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3]; }
            if (res32) { const f32x4 t = *reinterpret_cast<const f32x4*>(res32 + (long long)m * p.ldres + n); v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3]; }
            if (res16) {
                const u32x2 t = *reinterpret_cast<const u32x2*>(res16 + (long long)m * p.ldres + n);
                v[0] += __uint_as_float(t[0] << 16); v[1] += __uint_as_float(t[0] & 0xffff0000u);
                v[2] += __uint_as_float(t[1] << 16); v[3] += __uint_as_float(t[1] & 0xffff0000u);
            }
            if (cpre) { f32x4 t = {v[0], v[1], v[2], v[3]}; *reinterpret_cast<f32x4*>(cpre + (long long)m * p.ldpre + n) = t; }
            if (p.act == SQ_ACT_GELU) { v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]); }
            else if (p.act == SQ_ACT_RELU) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            if (gg) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(gg + (long long)m * p.ldgg + n);
                v[0] *= gelu_erf_grad(t[0]); v[1] *= gelu_erf_grad(t[1]); v[2] *= gelu_erf_grad(t[2]); v[3] *= gelu_erf_grad(t[3]);
            }
            if (c32) { f32x4 t = {v[0], v[1], v[2], v[3]}; *reinterpret_cast<f32x4*>(c32 + (long long)m * p.ldc + n) = t; }
            if (c16p) { u32x2 t = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])}; *reinterpret_cast<u32x2*>(c16p + (long long)m * p.ldc + n) = t; }
            if (c2) { u32x2 t = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])}; *reinterpret_cast<u32x2*>(c2 + (long long)m * p.ldc2 + n) = t; }
        } else {
            for (int e = 0; e < cnt; ++e) {
                float x = v[e];
                const int ne = n + e;
                if (bias) x += bias[ne];
                if (rowbias) x += rowbias[rb_row + ne];
                if (res32) x += res32[(long long)m * p.ldres + ne];
                if (res16) x += bf16_to_f32(res16[(long long)m * p.ldres + ne]);
                if (cpre) cpre[(long long)m * p.ldpre + ne] = x;
                if (p.act == SQ_ACT_GELU) x = gelu_erf(x);
                else if (p.act == SQ_ACT_RELU) x = fmaxf(x, 0.f);
                if (gg) x *= gelu_erf_grad(gg[(long long)m * p.ldgg + ne]);
                if (c32) c32[(long long)m * p.ldc + ne] = x;
                if (c16p) c16p[(long long)m * p.ldc + ne] = f32_to_bf16(x);
                if (c2) c2[(long long)m * p.ldc2 + ne] = f32_to_bf16(x);
            }
        }
    }
}

template <typename T, int WTM, int WTN>
int launch_cfg(const GemmArgs& a, hipStream_t stream) {
    constexpr int BM = 64 * WTM, BN = 64 * WTN;
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    const size_t lds = 2 * (BM + BN) * 128;
    dim3 grid(tiles, 1, a.batch), block(256);
    if (a.conv) hipLaunchKernelGGL((gemm_nt_kernel<T, WTM, WTN, true>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((gemm_nt_kernel<T, WTM, WTN, false>), grid, block, lds, stream, a);
    SQ_LAUNCH_CHECK();
    return SQ_OK;
}

int g_force_tile = 0, g_dbg = 0;

template <typename T>
int launch_t(const GemmArgs& a_in, hipStream_t stream) {
    GemmArgs a = a_in;
    a.dbg |= g_dbg;
    if (g_force_tile == 22) return launch_cfg<T, 2, 2>(a, stream);
    if (g_force_tile == 21) return launch_cfg<T, 2, 1>(a, stream);
    if (g_force_tile == 12) return launch_cfg<T, 1, 2>(a, stream);
    if (g_force_tile == 11) return launch_cfg<T, 1, 1>(a, stream);
    // tile choice: fill >= 256 CUs when the problem allows; narrow N (Cout 64) gets BN = 64
    const long long t128 = (long long)((a.M + 127) / 128) * ((a.N + 127) / 128) * a.batch;
    if (a.N <= 64) {
        if ((long long)((a.M + 127) / 128) * a.batch >= 256) return launch_cfg<T, 2, 1>(a, stream);
        return launch_cfg<T, 1, 1>(a, stream);
    }
    if (a.M <= 64) return launch_cfg<T, 1, 2>(a, stream);
    if (t128 >= 512) return launch_cfg<T, 2, 2>(a, stream);
    const long long t64n = (long long)((a.M + 127) / 128) * ((a.N + 63) / 64) * a.batch;
    if (t64n >= 384) return launch_cfg<T, 2, 1>(a, stream);
    return launch_cfg<T, 1, 1>(a, stream);
}

}  // namespace

extern "C" int sq_dbg_set(int key, int value) {
    if (key == 0) g_force_tile = value;
    else if (key == 1) g_dbg = value;
    else return SQ_ERR_ARG;
    return SQ_OK;
}

int sq_launch_gemm(const GemmArgs& a, int dtype, hipStream_t stream) {
    const int epc = dtype == SQ_BF16 ? 8 : 4;
    SQ_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.batch > 0, "gemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
    SQ_REQUIRE(a.K % epc == 0 && a.ldb % epc == 0, "gemm: K=%d / ldb=%d must be multiples of %d", a.K, a.ldb, epc);
    SQ_REQUIRE(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.B & 15) == 0, "gemm: A/B must be 16-byte aligned");
    SQ_REQUIRE(a.a_bytes > 0 && a.a_bytes < (1ull << 31) && a.b_bytes > 0 && a.b_bytes < (1ull << 31),
               "gemm: operand extents must be in (0, 2 GiB): %zu %zu", a.a_bytes, a.b_bytes);
    if (a.conv) {
        SQ_REQUIRE(a.Cin % (8 * epc) == 0, "conv: Cin=%d must be a multiple of the K-tile (%d)", a.Cin, 8 * epc);
    } else {
        SQ_REQUIRE(a.lda % epc == 0, "gemm: lda=%d must be a multiple of %d", a.lda, epc);
    }
    if (dtype != SQ_BF16 && dtype != SQ_F32) {
        sq_set_error("gemm: unknown dtype %d", dtype);
        return SQ_ERR_ARG;
    }
    GemmArgs av = a;
    {   // 16-byte epilogue accesses need every leading dimension / base / batch stride 4-element aligned
        auto al = [](const void* ptr, int ld, long long st, int elem) {
            return ptr == nullptr || (((uintptr_t)ptr % 16) == 0 && (ld * elem) % 16 == 0 && ((st * elem) % 16) == 0);
        };
        const int oe = a.out_dtype == SQ_BF16 ? 2 : 4, re = a.res_dtype == SQ_BF16 ? 2 : 4;
        // bf16 outputs are written 8 bytes at a time: 8-byte alignment is enough for them
        auto al8 = [](const void* ptr, int ld, long long st) {
            return ptr == nullptr || (((uintptr_t)ptr % 8) == 0 && (ld * 2) % 8 == 0 && ((st * 2) % 8) == 0);
        };
        bool ok = al(a.bias, 4, a.sBias, 4) && al(a.rowbias, a.ldrb, a.sRb, 4) && al(a.Cpre, a.ldpre, a.sPre, 4) &&
                  al(a.gelu_grad_of, a.ldgg, a.sGg, 4) && al8(a.C2, a.ldc2, a.sC2);
        ok = ok && (oe == 4 ? al(a.C, a.ldc, a.sC, 4) : al8(a.C, a.ldc, a.sC));
        ok = ok && (re == 4 ? al(a.res, a.ldres, a.sRes, 4) : al8(a.res, a.ldres, a.sRes));
        av.vec_epi = ok ? 1 : 0;
    }
    int prof = -1;
    if (sq_prof_on()) {
        // algorithmic work of this launch: 2*M*N*K flops; operands read once + output written once
        const double es = dtype == SQ_BF16 ? 2.0 : 4.0;
        const double flops = 2.0 * a.M * (double)a.N * a.K * a.batch;
        const double a_elems = a.conv ? (double)a.M / (a.OH * a.OW) * a.H * a.W * a.Cin : (double)a.M * a.K;
        const double bytes = (a_elems * es + (double)a.N * a.K * es) * a.batch +
                             (double)a.M * a.N * a.batch * (a.out_dtype == SQ_BF16 ? 2.0 : 4.0);
        char name[96];
        snprintf(name, sizeof(name), "%s_%s_M%d_N%d_K%d_b%d", a.conv ? "conv" : "gemm", dtype == SQ_BF16 ? "bf16" : "f32",
                 a.M, a.N, a.K, a.batch);
        prof = sq_prof_begin(name, flops, bytes, stream);
    }
    const int rc = dtype == SQ_BF16 ? launch_t<bf16_t>(av, stream) : launch_t<float>(av, stream);
    if (prof >= 0) sq_prof_end(prof, stream);
    return rc;
}
