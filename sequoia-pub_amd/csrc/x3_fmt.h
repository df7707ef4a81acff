// Plane formats of the split ("x3") mode: a fp32 value v travels as two 16-bit planes, hi = cvt(v) and lo = cvt(v - hi).
//   X3Fmt<false>: bf16 planes  -- fp32's exponent range, 8 + 8 significant bits: |v - hi - lo| <= 2^-18 |v|  (SQ_BF16X3)
//   X3Fmt<true>:  fp16 planes  -- 11 + 11 significant bits: fp32-class (2^-23) for |v| >= 2^-3, absolute error <= 2^-25
//                 below that (the lo plane turns subnormal), finite range |v| < 65504 (beyond: inf / NaN propagate to the
//                 output, by design not clamped)                                                            (SQ_F16X3)
//                 Subnormal planes: v_mfma_f32_32x32x16_f16 on gfx950 HONOURS subnormal inputs against a normal partner, bit for
//                 bit (subnormal lo planes, subnormal hi planes, the smallest subnormal 2^-24, negative ones); the product of TWO
//                 subnormal inputs is dropped (2^-20 x 2^-20 and 2^-24 x 2^-20 observed lost) -- that product is < 2^-28 in stored
//                 units, an eighth of the 2^-25 above.  Pinned by tests/test_gpu_x3.py::test_f16x3_subnormal_planes_*.
// Both feed 16-bit MFMAs at the same rate; a product is a_hi.b_hi + a_hi.b_lo + a_lo.b_hi with fp32 accumulation.
#pragma once
#include "sq_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

template <bool F16> struct X3Fmt;

template <> struct X3Fmt<false> {
    static __device__ __forceinline__ uint32_t pack2(float a, float b) { return pack_bf16x2(a, b); }
    static __device__ __forceinline__ float lo_f(uint32_t u) { return __uint_as_float(u << 16); }            // element 0 of a packed pair
    static __device__ __forceinline__ float hi_f(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }    // element 1
    static __device__ __forceinline__ float one(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
    static __device__ __forceinline__ float sum_lo(uint32_t h, uint32_t l) { return lo_f(h) + lo_f(l); }     // element 0 of hi + lo
    static __device__ __forceinline__ float sum_hi(uint32_t h, uint32_t l) { return hi_f(h) + hi_f(l); }
    static __device__ __forceinline__ uint32_t rest2(float a, float b, uint32_t h) { return pack2(a - lo_f(h), b - hi_f(h)); }   // the lo plane of (a, b) given their hi plane
    static __device__ __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x16& acc) {
        union { u32x4 u; bf16x8 h; } ua, ub;
        ua.u = a; ub.u = b;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.h, ub.h, acc, 0, 0, 0);
    }
};

template <> struct X3Fmt<true> {
    static __device__ __forceinline__ uint32_t pack2(float a, float b) {
        const sq_f32x2 v = {a, b};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));       // round to nearest even
    }
    static __device__ __forceinline__ float lo_f(uint32_t u) { return (float)__builtin_bit_cast(f16x2, u)[0]; }
    static __device__ __forceinline__ float hi_f(uint32_t u) { return (float)__builtin_bit_cast(f16x2, u)[1]; }
    static __device__ __forceinline__ float one(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
    // hi + lo and v - hi on v_fma_mix_f32 (an fp32 FMA whose operands may be either half of a packed fp16 pair): ONE instruction where
    // v_cvt_f32_f16 + v_add / v_sub_f32 are two or three, same bits -- the conversions are exact and x * (+-1) + y rounds once, as the
    // addition does.  hipcc does not form it on its own (round 5: 2720 v_cvt_f32_f16 in chain_x3.hip's ISA, no v_fma_mix); the
    // epilogues of the split modes are VALU time the matrix pipes wait for.
    static __device__ __forceinline__ float sum_lo(uint32_t h, uint32_t l) {
        float r; asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(h), "v"(l)); return r;
    }
    static __device__ __forceinline__ float sum_hi(uint32_t h, uint32_t l) {
        float r; asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(h), "v"(l)); return r;
    }
    static __device__ __forceinline__ uint32_t rest2(float a, float b, uint32_t h) {
        float ra, rb;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(h), "v"(a));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(h), "v"(b));
        return pack2(ra, rb);
    }
    static __device__ __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x16& acc) {
        union { u32x4 u; f16x8 h; } ua, ub;
        ua.u = a; ub.u = b;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ua.h, ub.h, acc, 0, 0, 0);
    }
};

// ReLU of the split modes: lets NaN through.  An activation beyond fp16's range splits into hi = +inf, lo = -inf, the next
// product turns that into NaN -- and fmaxf(NaN, 0) = 0 would hide it in the very next epilogue.  With this form the NaN
// reaches the pooled features, where avgpool7_x3_kernel raises the caller's non-finite flag (resnet.hip).
__device__ __forceinline__ float x3_relu(float v) { return v < 0.f ? 0.f : v; }

// 8 fp32 values -> packed hi / lo planes (16 bytes each)
template <bool F16>
__device__ __forceinline__ void x3_split8(const float (&v)[8], u32x4& hi, u32x4& lo) {
    using F = X3Fmt<F16>;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hi[e] = F::pack2(v[2 * e], v[2 * e + 1]);
        lo[e] = F::rest2(v[2 * e], v[2 * e + 1], hi[e]);
    }
}
// hi + lo (exact in fp32: both planes are multiples of one ulp of the value they split)
template <bool F16>
__device__ __forceinline__ void x3_join8(const u32x4& hi, const u32x4& lo, float (&v)[8]) {
    using F = X3Fmt<F16>;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[2 * e] = F::sum_lo(hi[e], lo[e]);
        v[2 * e + 1] = F::sum_hi(hi[e], lo[e]);
    }
}
