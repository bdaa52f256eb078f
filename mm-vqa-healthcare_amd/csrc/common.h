// Shared device helpers for the gfx950 kernels of libm3ae_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "m3ae_hip.h"

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define DEVINL __device__ __forceinline__

DEVINL float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// round-to-nearest-even; the plain __bf16 cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaN a NaN
DEVINL bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
// two floats -> one dword of two bf16 (lo in bits 0..15): a single v_cvt_pk_bf16_f32
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
DEVINL uint32_t pack2bf(float lo, float hi) {
    const bf16x2_t v = __builtin_convertvector((f32x2){lo, hi}, bf16x2_t);
    return __builtin_bit_cast(uint32_t, v);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static DEVINL float ld(const float* p) { return *p; }
    static DEVINL void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static DEVINL float ld(const bf16_t* p) { return bf2f(*p); }
    static DEVINL void st(bf16_t* p, float v) { *p = f2bf(v); }
};

DEVINL float act_fwd(float x, int act) {
    switch (act) {
        case M3AE_ACT_GELU: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
        case M3AE_ACT_QUICKGELU: return x / (1.0f + expf(-1.702f * x));
        case M3AE_ACT_TANH: return tanhf(x);
        case M3AE_ACT_RELU: return x > 0.f ? x : 0.f;
        default: return x;
    }
}
// derivative of act at the PRE-activation value x
DEVINL float act_bwd(float x, int act) {
    switch (act) {
        case M3AE_ACT_GELU: {
            float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
            float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
            return cdf + x * pdf;
        }
        case M3AE_ACT_QUICKGELU: {
            float s = 1.0f / (1.0f + expf(-1.702f * x));
            return s * (1.0f + 1.702f * x * (1.0f - s));
        }
        case M3AE_ACT_TANH: {
            float t = tanhf(x);
            return 1.0f - t * t;
        }
        case M3AE_ACT_MULAUX: return x;  // the saved value IS the derivative
        case M3AE_ACT_RELU: return x > 0.f ? 1.f : 0.f;
        default: return 1.0f;
    }
}

// ---- fast erf-GELU for the bf16 MFMA epilogues -------------------------------------------------------------
// erf by Abramowitz-Stegun 7.1.26 (|abs err| < 1.5e-7, far below bf16 resolution): one v_exp + one v_rcp + 6 FMAs,
// instead of libm erff (~50 instructions per element, which made the K = 768 GELU GEMMs epilogue-bound).
// The same exp(-x^2/2) term serves the Gaussian pdf of the derivative.  Parity (fp32) mode never uses these.
DEVINL void gelu_terms_fast(float x, float& cdf, float& pdf) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));  // raw v_rcp_f32 (1 ulp), not an IEEE division
    const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);  // = exp(-x^2 / 2)
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float half_erfc = 0.5f * poly * t * e;  // 0.5 erfc(|x| / sqrt 2)
    cdf = x < 0.f ? half_erfc : 1.0f - half_erfc;
    pdf = 0.39894228040143267794f * e;
}
DEVINL float act_fwd_fast(float x, int act) {
    if (act == M3AE_ACT_GELU) {
        float cdf, pdf;
        gelu_terms_fast(x, cdf, pdf);
        return x * cdf;
    }
    if (act == M3AE_ACT_QUICKGELU) return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.45546695959f * x));
    return act_fwd(x, act);
}
DEVINL float act_bwd_fast(float x, int act) {
    if (act == M3AE_ACT_GELU) {
        float cdf, pdf;
        gelu_terms_fast(x, cdf, pdf);
        return fmaf(x, pdf, cdf);
    }
    if (act == M3AE_ACT_QUICKGELU) {
        const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.45546695959f * x));
        return s * (1.0f + 1.702f * x * (1.0f - s));
    }
    return act_bwd(x, act);
}

// Two elements at a time: the FMA / multiply chain compiles to packed fp32 instructions (v_pk_fma_f32, v_pk_mul_f32:
// two lanes-worth of work per issue slot); the two transcendentals per element stay scalar-per-lane.
typedef float f32x2 __attribute__((ext_vector_type(2)));
DEVINL f32x2 splat2(float v) { return (f32x2){v, v}; }
DEVINL void gelu_terms_fast2(f32x2 x, f32x2& cdf, f32x2& pdf) {
    const f32x2 ax = __builtin_elementwise_abs(x);
    const f32x2 den = __builtin_elementwise_fma(splat2(0.3275911f * 0.70710678118654752440f), ax, splat2(1.0f));
    const f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
    const f32x2 xx = x * x * splat2(-0.72134752044448170368f);
    const f32x2 e = {__builtin_amdgcn_exp2f(xx[0]), __builtin_amdgcn_exp2f(xx[1])};
    f32x2 poly = __builtin_elementwise_fma(splat2(1.061405429f), t, splat2(-1.453152027f));
    poly = __builtin_elementwise_fma(poly, t, splat2(1.421413741f));
    poly = __builtin_elementwise_fma(poly, t, splat2(-0.284496736f));
    poly = __builtin_elementwise_fma(poly, t, splat2(0.254829592f));
    const f32x2 h = splat2(0.5f) * poly * t * e;
    const f32x2 om = splat2(1.0f) - h;
    cdf = (f32x2){x[0] < 0.f ? h[0] : om[0], x[1] < 0.f ? h[1] : om[1]};
    pdf = splat2(0.39894228040143267794f) * e;
}
// y[0..n) = act(x[0..n)) / act'(x[0..n)), n even
template <int N> DEVINL void act_fwd_fast_n(float* x, int act) {
    if (act == M3AE_ACT_GELU) {
#pragma unroll
        for (int t = 0; t < N; t += 2) {
            f32x2 c, p, v = {x[t], x[t + 1]};
            gelu_terms_fast2(v, c, p);
            v = v * c;
            x[t] = v[0]; x[t + 1] = v[1];
        }
    } else {
#pragma unroll
        for (int t = 0; t < N; ++t) x[t] = act_fwd_fast(x[t], act);
    }
}
// x <- act(x), d <- act'(x): one evaluation of the erf / sigmoid terms serves both
template <int N> DEVINL void act_fwd_grad_fast_n(float* x, float* d, int act) {
    if (act == M3AE_ACT_GELU) {
#pragma unroll
        for (int t = 0; t < N; t += 2) {
            f32x2 c, p, v = {x[t], x[t + 1]};
            gelu_terms_fast2(v, c, p);
            const f32x2 g = __builtin_elementwise_fma(v, p, c);
            v = v * c;
            x[t] = v[0]; x[t + 1] = v[1];
            d[t] = g[0]; d[t + 1] = g[1];
        }
    } else if (act == M3AE_ACT_QUICKGELU) {
#pragma unroll
        for (int t = 0; t < N; ++t) {
            const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.45546695959f * x[t]));
            d[t] = s * fmaf(1.702f * x[t], 1.0f - s, 1.0f);
            x[t] = x[t] * s;
        }
    } else {
#pragma unroll
        for (int t = 0; t < N; ++t) { d[t] = act_bwd(x[t], act); x[t] = act_fwd(x[t], act); }
    }
}
template <int N> DEVINL void act_bwd_mul_fast_n(float* x, const float* pre, int act) {  // x *= act'(pre)
    if (act == M3AE_ACT_GELU) {
#pragma unroll
        for (int t = 0; t < N; t += 2) {
            f32x2 c, p, u = {pre[t], pre[t + 1]}, v = {x[t], x[t + 1]};
            gelu_terms_fast2(u, c, p);
            v = v * __builtin_elementwise_fma(u, p, c);
            x[t] = v[0]; x[t + 1] = v[1];
        }
    } else {
#pragma unroll
        for (int t = 0; t < N; ++t) x[t] *= act_bwd_fast(pre[t], act);
    }
}

// ---- dropout: counter-based keep mask ---------------------------------------------------------------------
// Every dropout site is a 2-D array [rows][cols]; element (row, col) has the index idx = row * ld + col with
// ld = cols rounded up to a multiple of 4 (drop_ld).  Four consecutive indices share ONE 32-bit hash
//     h(g) = lowbias32((lo32(g) ^ lo32(seed)) + hi32(g) * 0x9E3779B9 + hi32(seed)),  g = idx >> 2,
// and element j = idx & 3 keeps iff rotr(h, 8 j) >= p * 2^32: each rotation is uniform on 32 bits (exact keep
// probability), and the four decisions read four different top bytes of h.  Stateless: forward and backward
// regenerate the same mask from (seed, idx).  A lane that owns 4 aligned consecutive elements (GEMM / LayerNorm
// epilogues, the attention score fragments) pays one hash (2 integer multiplies) per 4 elements.
DEVINL uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// salt (ABI 3): optional DEVICE pointer to a 32-bit per-step value folded into the key by drop_resolve() at kernel entry.  A step
// captured in a hipGraph replays with the SAME kernel arguments (seeds included); the caller bumps *salt between replays and every
// site draws a fresh mask.  NULL (eager runs: the host draws a fresh seed per site and step) leaves the key as it is.
struct DropState { uint32_t key_lo, key_hi, thr; float inv_keep; const uint32_t* salt; };
static inline DropState make_drop(float p, uint64_t seed, const void* salt = nullptr) {
    DropState d;
    d.salt = (const uint32_t*)salt;
    d.key_lo = (uint32_t)seed; d.key_hi = (uint32_t)(seed >> 32);
    double t = (double)p * 4294967296.0;
    d.thr = t >= 4294967295.0 ? 4294967295U : (uint32_t)t;
    d.inv_keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    return d;
}
DEVINL void drop_resolve(DropState& d) {   // once per kernel, before the first hash (wave-uniform scalar load)
    if (d.salt) d.key_hi += *d.salt * 0x85EBCA6BU;
}
DEVINL DropState make_drop_dev(float p, uint64_t seed, const void* salt = nullptr) {  // same as make_drop, callable on the device
    DropState d;
    d.salt = (const uint32_t*)salt;
    d.key_lo = (uint32_t)seed; d.key_hi = (uint32_t)(seed >> 32);
    const float t = p * 4294967296.0f;
    d.thr = t >= 4294967040.0f ? 4294967295U : (uint32_t)t;
    d.inv_keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    return d;
}
__host__ __device__ static inline int64_t drop_ld(int64_t cols) { return (cols + 3) & ~(int64_t)3; }
DEVINL uint32_t drop_hash(const DropState& d, uint64_t group) {
    return lowbias32((((uint32_t)group) ^ d.key_lo) + (uint32_t)(group >> 32) * 0x9E3779B9U + d.key_hi);
}
DEVINL uint32_t rotr32(uint32_t x, uint32_t r) { return __builtin_amdgcn_alignbit(x, x, r); }
DEVINL bool drop_keep(const DropState& d, uint64_t idx) {
    return rotr32(drop_hash(d, idx >> 2), 8u * ((uint32_t)idx & 3u)) >= d.thr;
}
DEVINL float drop_apply(const DropState& d, uint64_t idx, float x) { return drop_keep(d, idx) ? x * d.inv_keep : 0.f; }
// x[0..3] are the elements idx4 .. idx4 + 3, idx4 % 4 == 0
DEVINL void drop_apply4(const DropState& d, uint64_t idx4, float* x) {
    const uint32_t h = drop_hash(d, idx4 >> 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = rotr32(h, 8u * j) >= d.thr ? x[j] * d.inv_keep : 0.f;
}

DEVINL float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
DEVINL float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

static inline int hip_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
