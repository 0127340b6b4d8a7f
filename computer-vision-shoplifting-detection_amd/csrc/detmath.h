// Canonical fp32 transcendental of the engine (DESIGN.md "canonical arithmetic"): a libm-free exp built from
// fmaf / mul / add only, so that a CPU restatement (oracle/det_oracle.c states the same sequence) reproduces
// every SiLU, sigmoid and DFL softmax of the GPU path bit for bit.  <= 1.4 ulp from the true exp.
#pragma once
#include <hip/hip_runtime.h>

namespace mi355 {

__device__ __forceinline__ float det_expf(float x) {
    x = __builtin_fminf(__builtin_fmaxf(x, -104.0f), 89.0f);
    const float t = __builtin_fmaf(x, 1.44269504088896341f, 12582912.0f);   // round(x * log2 e), ties to even
    const float n = t - 12582912.0f;
    float r = __builtin_fmaf(n, -0.693145751953125f, x);
    r = __builtin_fmaf(n, -1.428606765330187045e-06f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    float e = __builtin_fmaf(p, r2, r);
    e = e + 1.0f;
    const int ni = (int)n;
    const int n1 = ni / 2, n2 = ni - n1;
    const float s1 = __int_as_float((n1 + 127) << 23), s2 = __int_as_float((n2 + 127) << 23);
    return (e * s1) * s2;
}

// exp for SiLU only: the argument is clamped to [-87.25, 87], where 2^n is a normal float, so one exact scaling replaces
// det_expf's two half-steps, and where 1 + e < 2^126, so that 1 / (1 + e) is a normal float too (det_silu's division relies on
// it).  Inside that range the bits are det_expf's; outside it the difference vanishes in 1 + e (below 2^-125) or is a quotient
// of magnitude 1e-36 (v < -87).  The clamp is one v_med3_f32 (fminf(fmaxf(x, lo), hi) for every non-NaN x).
__device__ __forceinline__ float det_expf_silu(float x) {
    x = __builtin_amdgcn_fmed3f(x, -87.25f, 87.0f);
    const float t = __builtin_fmaf(x, 1.44269504088896341f, 12582912.0f);
    const float n = t - 12582912.0f;
    float r = __builtin_fmaf(n, -0.693145751953125f, x);
    r = __builtin_fmaf(n, -1.428606765330187045e-06f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    float e = __builtin_fmaf(p, r2, r);
    e = e + 1.0f;
    return e * __int_as_float(((int)n + 127) << 23);
}

// v / d, correctly rounded (= the IEEE quotient the oracle's C division gives), for finite v and d in [1, 2^126): the
// reciprocal-refinement sequence the compiler emits for fp32 division (v_rcp_f32, one Newton step on r, two residual
// corrections of q) WITHOUT its range scaling and fix-up instructions (v_div_scale x2, v_div_fixup: 3 of 11), which only act on
// operands outside that range.  The fp32 matrix instructions and the vector ALU share the SIMD's issue cycles (PMC:
// SQ_VALU_MFMA_COEXEC_CYCLES = 0, time = 32 x MFMAs + 2 x VALU instructions), so every instruction of the epilogue counts.
// Not IEEE for v = +-inf (NaN instead of +-inf) and v = -0 (+0): an overflowed activation is garbage either way, and a conv
// output (+0 + partial sums + bias) is never -0.
__device__ __forceinline__ float det_div_ge1(float v, float d) {
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = v * r;
    float m = __builtin_fmaf(-d, q, v);
    q = __builtin_fmaf(m, r, q);
    m = __builtin_fmaf(-d, q, v);
    return __builtin_fmaf(m, r, q);
}

__device__ __forceinline__ float det_silu(float v) { return det_div_ge1(v, 1.0f + det_expf_silu(-v)); }
__device__ __forceinline__ float det_sigmoid(float v) { return 1.0f / (1.0f + det_expf(-v)); }

}  // namespace mi355
