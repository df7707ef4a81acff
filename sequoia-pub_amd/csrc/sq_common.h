// Shared device/host helpers for libsequoia_hip (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define SQ_OK 0
#define SQ_ERR_ARG -1
#define SQ_ERR_HIP -2
#define SQ_ERR_WORKSPACE -3
#define SQ_ERR_UNSUPPORTED -4

#define SQ_F32 0
#define SQ_BF16 1
#define SQ_F16X3 3    // the same with fp16 planes: fp32-class accuracy (2^-23), finite range 65504 (x3_fmt.h)
#define SQ_BF16X3 2   // split bf16: every fp32 value as hi + lo bf16 planes, three MFMAs per product (gemm_x3.hip)

void sq_set_error(const char* fmt, ...);
bool sq_prof_on();

// Helper streams of the library (per device, created on first use, non-blocking) with a pool of timing-less events:
// independent branches of a pass run beside its main chain; the caller's stream hands work over with events.
struct SqSideStream { hipStream_t stream; hipEvent_t* events; int n_events; };
SqSideStream* sq_side_stream(int which, int n_events);      // which = 0, 1; nullptr on failure
bool sq_env_flag(const char* name);                           // set and not "0"
int sq_prof_begin(const char* name, double flops, double bytes, hipStream_t st);
void sq_prof_end(int idx, hipStream_t st);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE setting: a launcher's attribute block has to run once for
// every device the process drives, not once per process.  Setting it twice from two threads is harmless.
struct SqDevOnce {
    unsigned long long mask = 0;
    static unsigned long long bit() { int d = 0; (void)hipGetDevice(&d); return 1ull << (d & 63); }
    bool needed() const { return (__atomic_load_n(&mask, __ATOMIC_ACQUIRE) & bit()) == 0; }
    void done() { __atomic_fetch_or(&mask, bit(), __ATOMIC_RELEASE); }
};

#define SQ_HIP_CHECK(expr)                                                        \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess) {                                                   \
            sq_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),   \
                         __FILE__, __LINE__);                                     \
            return SQ_ERR_HIP;                                                    \
        }                                                                         \
    } while (0)

#define SQ_REQUIRE(cond, ...)                                                     \
    do {                                                                          \
        if (!(cond)) {                                                            \
            sq_set_error(__VA_ARGS__);                                            \
            return SQ_ERR_ARG;                                                    \
        }                                                                         \
    } while (0)

#define SQ_LAUNCH_CHECK()                                                         \
    do {                                                                          \
        hipError_t _e = hipGetLastError();                                        \
        if (_e != hipSuccess) {                                                   \
            sq_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), \
                         __FILE__, __LINE__);                                     \
            return SQ_ERR_HIP;                                                    \
        }                                                                         \
    } while (0)

typedef uint16_t bf16_t;   // storage type for bfloat16

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// round-to-nearest-even through gfx950's v_cvt_pk_bf16_f32 (one instruction per PAIR; the integer
// add-and-shift formulation costs ~5 VALU ops per value and made bf16-output epilogues VALU-bound)
typedef float sq_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 sq_bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    const __bf16 h = (__bf16)f;
    return __builtin_bit_cast(bf16_t, h);
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const sq_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, sq_bf16x2));
}

// exact (erf) GELU: torch.nn.GELU() default, tformer_lin.py:20-24,57
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// d/dx of exact GELU
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// bf16 compute mode, GELU' (backward): erf by Abramowitz & Stegun 7.1.26 (|error| < 1.5e-7, far below the bf16 rounding of the result)
// sharing exp(-u^2) with the Gaussian density: ~12 VALU ops per element instead of ~40 for erff + expf.  GEMM
// epilogues with GELU / GELU' were VALU-bound on the exact forms.  fp32 (parity) mode keeps erff.
__device__ __forceinline__ void sq_erf_exp_fast(float u, float& erf_u, float& e) {
    const float a = fabsf(u);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * a);
    e = __expf(-a * a);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    erf_u = copysignf(1.0f - poly * e, u);
}
// bf16 compute mode, forward: erf as an odd polynomial on the clamped argument, erf(u) ~ z P(z^2), z = med3(u, -3, 3), degree 8 in
// z^2 (near-minimax fit, |error| < 2.8e-5 everywhere incl. the clamp: 1 - erf(3) = 2.2e-5) -- 12 full-rate VALU operations, no
// transcendental (v_exp / v_rcp issue at quarter rate: the A&S form above costs ~21 slots).  GELU epilogues of large products are
// VALU time with the matrix pipes idle; the result is rounded to bf16 (2^-9 relative) right behind it.
__device__ __forceinline__ float sq_erf_poly(float u) {
    const float z = __builtin_amdgcn_fmed3f(u, -3.0f, 3.0f);
    const float t = z * z;
    float p = 4.074212256e-08f;
    p = fmaf(p, t, -1.944823225e-06f); p = fmaf(p, t, 4.106052802e-05f); p = fmaf(p, t, -5.110368947e-04f);
    p = fmaf(p, t, 4.235427361e-03f); p = fmaf(p, t, -2.510286123e-02f); p = fmaf(p, t, 1.110793352e-01f);
    p = fmaf(p, t, -3.753148615e-01f); p = fmaf(p, t, 1.128268480e+00f);
    return p * z;
}
template <bool FAST>
__device__ __forceinline__ float sq_gelu(float x) {
    if constexpr (FAST) {
        // left of the clamp (x < -3 sqrt 2) the polynomial's 1 + erf stays at 2.2e-5 and 0.5 x (1 + erf) would GROW with |x| instead of
        // decaying: 0 there (exact value between -4.7e-5 and 0), so the absolute error is <= 4.7e-5 for every x.  Right of the clamp the
        // result is x (1 - 1.1e-5): a relative error below bf16's.
        const float u = x * 0.70710678118654752440f;
        const float g = 0.5f * x * (1.0f + sq_erf_poly(u));
        return u < -3.0f ? 0.0f : g;
    } else {
        return gelu_erf(x);
    }
}
template <bool FAST>
__device__ __forceinline__ float sq_gelu_grad(float x) {
    if constexpr (FAST) {
        float er, e;
        sq_erf_exp_fast(x * 0.70710678118654752440f, er, e);            // e = exp(-x^2 / 2)
        return 0.5f * (1.0f + er) + x * 0.39894228040143267794f * e;
    } else {
        return gelu_erf_grad(x);
    }
}

// Sum over aligned groups of 4 / 8 consecutive lanes on DPP moves (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror), every lane of
// a group ending with the same bits as the xor-butterfly `v += __shfl_xor(v, 1); ... 2; ... 4` (only commutativity separates the two).
// hipcc turns __shfl_xor into ds_bpermute_b32 -- an LDS round trip per step, six dependent ones per LayerNorm(64) row in the GEMM
// epilogues (round 5: 96 of them in gemm_p8's combiner epilogue).
__device__ __forceinline__ float sq_dpp_add(float v, int ctrl_is /* 0: xor 1, 1: xor 2, 2: the other quad of the 8-group */) {
    const int x = __float_as_int(v);
    const int y = ctrl_is == 0 ? __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false)
                : ctrl_is == 1 ? __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false)
                               : __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, false);
    return v + __int_as_float(y);
}
__device__ __forceinline__ float group4_sum(float v) { return sq_dpp_add(sq_dpp_add(v, 0), 1); }
__device__ __forceinline__ float group8_sum(float v) { return sq_dpp_add(sq_dpp_add(sq_dpp_add(v, 0), 1), 2); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <typename T> struct sq_type;
template <> struct sq_type<float> { static constexpr int id = SQ_F32; };
template <> struct sq_type<bf16_t> { static constexpr int id = SQ_BF16; };

static inline size_t sq_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int sq_dtype_size(int dtype) { return dtype == SQ_BF16 ? 2 : 4; }   // SQ_BF16X3: two 2-byte planes = 4
