// Shared GEMM epilogue (NT and TN kernels, split-K reduction).
#pragma once
#include "gemm.h"

// Epilogue of one 8-wide chunk (row m, columns n..n+cnt-1); v holds the raw accumulators.
// Vector path: 2 x 16-byte fp32 accesses / one 16-byte bf16 access per operand.
// EPI selects which transcendental paths are compiled in: 0 none/ReLU only, 1 + GELU activation,
// 2 + GELU' multiply, 3 both (erff expands to >100 instructions per element, so lean kernels leave it out);
// FAST (bf16 compute mode) takes the polynomial erf (forward) / the A&S erf sharing its exponential with the density (GELU')
template <int EPI, bool FAST = false>
static __device__ __forceinline__ void epi_apply(const GemmArgs& p, int z, int m, int n, float (&v)[8], int cnt, bool vec,
                                          const float* pre_res = nullptr, const float* pre_bias = nullptr, void* c_ovr = nullptr) {
    const float* bias = p.bias ? p.bias + (long long)z * p.sBias : nullptr;
    const float* rowbias = p.rowbias ? p.rowbias + (long long)z * p.sRb : nullptr;
    const float* res32 = (p.res && p.res_dtype == SQ_F32) ? reinterpret_cast<const float*>(p.res) + (long long)z * p.sRes : nullptr;
    const bf16_t* res16 = (p.res && p.res_dtype == SQ_BF16) ? reinterpret_cast<const bf16_t*>(p.res) + (long long)z * p.sRes : nullptr;
    void* const cbase = c_ovr ? c_ovr : p.C;          // c_ovr: a grouped launch's member result (z is 0 then)
    float* c32 = p.out_dtype == SQ_F32 ? reinterpret_cast<float*>(cbase) + (long long)z * p.sC : nullptr;
    bf16_t* c16p = p.out_dtype == SQ_BF16 ? reinterpret_cast<bf16_t*>(cbase) + (long long)z * p.sC : nullptr;
    bf16_t* c2 = p.C2 ? p.C2 + (long long)z * p.sC2 : nullptr;
    float* cpre = (p.Cpre && p.pre_dtype == SQ_F32) ? reinterpret_cast<float*>(p.Cpre) + (long long)z * p.sPre : nullptr;
    bf16_t* cpre16 = (p.Cpre && p.pre_dtype == SQ_BF16) ? reinterpret_cast<bf16_t*>(p.Cpre) + (long long)z * p.sPre : nullptr;
    const float* gg = (p.gelu_grad_of && p.gg_dtype == SQ_F32) ? reinterpret_cast<const float*>(p.gelu_grad_of) + (long long)z * p.sGg : nullptr;
    const bf16_t* gg16 = (p.gelu_grad_of && p.gg_dtype == SQ_BF16) ? reinterpret_cast<const bf16_t*>(p.gelu_grad_of) + (long long)z * p.sGg : nullptr;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
    const long long rb_row = rowbias ? (long long)(m / p.rows_per_group) * p.ldrb : 0;
    auto add8 = [&](const float* src) {
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(src), t1 = *reinterpret_cast<const f32x4*>(src + 4);
        v[0] += t0[0]; v[1] += t0[1]; v[2] += t0[2]; v[3] += t0[3]; v[4] += t1[0]; v[5] += t1[1]; v[6] += t1[2]; v[7] += t1[3];
    };
    auto st8 = [&](float* dst) {
        const f32x4 t0 = {v[0], v[1], v[2], v[3]}, t1 = {v[4], v[5], v[6], v[7]};
        *reinterpret_cast<f32x4*>(dst) = t0; *reinterpret_cast<f32x4*>(dst + 4) = t1;
    };
    auto st8h = [&](bf16_t* dst) {
        const u32x4 t = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
        *reinterpret_cast<u32x4*>(dst) = t;
    };
    if (vec && cnt == 8) {
        if (pre_bias) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += pre_bias[e];
        } else if (bias) {
            add8(bias + n);
        }
        if (rowbias) add8(rowbias + rb_row + n);
        if (pre_res) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += pre_res[e];
        } else if (res32) {
            add8(res32 + (long long)m * p.ldres + n);
        } else if (res16) {
            const u32x4 t = *reinterpret_cast<const u32x4*>(res16 + (long long)m * p.ldres + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] += __uint_as_float(t[e] << 16); v[2 * e + 1] += __uint_as_float(t[e] & 0xffff0000u); }
        }
        if (cpre) st8(cpre + (long long)m * p.ldpre + n);
        if (cpre16) st8h(cpre16 + (long long)m * p.ldpre + n);
        if ((EPI & 1) && p.act == SQ_ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = sq_gelu<FAST>(v[e]);
        } else if (p.act == SQ_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if ((EPI & 2) && gg) {
            const float* gs = gg + (long long)m * p.ldgg + n;
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(gs), t1 = *reinterpret_cast<const f32x4*>(gs + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] *= sq_gelu_grad<FAST>(t0[e]); v[4 + e] *= sq_gelu_grad<FAST>(t1[e]); }
        }
        if ((EPI & 2) && gg16) {
            const u32x4 t = *reinterpret_cast<const u32x4*>(gg16 + (long long)m * p.ldgg + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[2 * e] *= sq_gelu_grad<FAST>(__uint_as_float(t[e] << 16));
                v[2 * e + 1] *= sq_gelu_grad<FAST>(__uint_as_float(t[e] & 0xffff0000u));
            }
        }
        if (c32) st8(c32 + (long long)m * p.ldc + n);
        if (c16p) st8h(c16p + (long long)m * p.ldc + n);
        if (c2) st8h(c2 + (long long)m * p.ldc2 + n);
    } else {
        for (int e = 0; e < cnt; ++e) {
            float x = v[e];
            const int ne = n + e;
            if (bias) x += bias[ne];
            if (rowbias) x += rowbias[rb_row + ne];
            if (res32) x += res32[(long long)m * p.ldres + ne];
            if (res16) x += bf16_to_f32(res16[(long long)m * p.ldres + ne]);
            if (cpre) cpre[(long long)m * p.ldpre + ne] = x;
            if (cpre16) cpre16[(long long)m * p.ldpre + ne] = f32_to_bf16(x);
            if ((EPI & 1) && p.act == SQ_ACT_GELU) x = sq_gelu<FAST>(x);
            else if (p.act == SQ_ACT_RELU) x = fmaxf(x, 0.f);
            if ((EPI & 2) && gg) x *= sq_gelu_grad<FAST>(gg[(long long)m * p.ldgg + ne]);
            if ((EPI & 2) && gg16) x *= sq_gelu_grad<FAST>(bf16_to_f32(gg16[(long long)m * p.ldgg + ne]));
            if (c32) c32[(long long)m * p.ldc + ne] = x;
            if (c16p) c16p[(long long)m * p.ldc + ne] = f32_to_bf16(x);
            if (c2) c2[(long long)m * p.ldc2 + ne] = f32_to_bf16(x);
        }
    }
}

