// Transcendentals of the numeric spec, shared by every kernel file that
// evaluates a sigmoid / tanh / exp / log: fixed polynomials built only from fmaf, rint, IEEE
// divide and exponent-field arithmetic, never libm/ocml, so the HIP path and the CPU oracle
// (oracle/csrc/ocrs_oracle.c restates them independently) agree bit for bit.
#pragma once
#include <hip/hip_runtime.h>

namespace ocrs {
namespace k {

__device__ __forceinline__ float spec_expf(float x) {
    if (x != x) return x;
    x = x > 88.0f ? 88.0f : x;
    x = x < -87.0f ? -87.0f : x;
    float kf = rintf(x * 1.44269504088896341f);
    float r = fmaf(kf, -0.693145751953125f, x);
    r = fmaf(kf, -1.42860682030941723212e-6f, r);
    float p = 1.98412698412698413e-4f;
    p = fmaf(p, r, 1.38888888888888894e-3f);
    p = fmaf(p, r, 8.33333333333333322e-3f);
    p = fmaf(p, r, 4.16666666666666644e-2f);
    p = fmaf(p, r, 1.66666666666666657e-1f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    int bits = __float_as_int(p) + (((int)kf) << 23);
    return __int_as_float(bits);
}

__device__ __forceinline__ float spec_logf(float s) {
    unsigned u = __float_as_uint(s);
    int e = (int)((u >> 23) & 0xffu) - 127;
    float m = __uint_as_float((u & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356237309515f) { m = m * 0.5f; e += 1; }
    float t = (m - 1.0f) / (m + 1.0f);
    float t2 = t * t;
    float p = 1.11111111111111105e-1f;
    p = fmaf(p, t2, 1.42857142857142849e-1f);
    p = fmaf(p, t2, 0.2f);
    p = fmaf(p, t2, 3.33333333333333315e-1f);
    p = fmaf(p, t2, 1.0f);
    float lm = (2.0f * t) * p;
    return fmaf((float)e, 0.693147180559945286f, lm);
}

__device__ __forceinline__ float spec_sigmoidf(float x) { return 1.0f / (1.0f + spec_expf(-x)); }
__device__ __forceinline__ float spec_tanhf(float x) {
    float t = spec_expf(2.0f * x);
    return (t - 1.0f) / (t + 1.0f);
}

}  // namespace k
}  // namespace ocrs
