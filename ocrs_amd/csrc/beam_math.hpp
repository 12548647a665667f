// float64 log-sum-exp of the CTC prefix beam search (rten::ctc::CtcDecoder::decode_beam as called at
// ocrs/src/recognition.rs:512-514), shared by the host implementation (ctc_beam.cpp) and the HIP kernel
// (kernels_beam.hip) and restated operation for operation in oracle/pipeline.py.
//
// libm / ocml / CPython's math module each round exp and log their own way in the last place; a beam decision
// (which of two prefixes survives the pruning) could then differ between host, device and oracle.  So the two
// transcendentals are FIXED polynomials evaluated with plain IEEE double multiplies, adds and one divide, in a
// fixed order (every translation unit is built with -ffp-contract=off; CPython floats are IEEE doubles and
// never fuse) — the same idea as spec_math.hpp for fp32.  Accuracy ~1e-16 relative, far below what separates
// two hypotheses.
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define OCRS_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define OCRS_HD inline
#endif

namespace ocrs {
namespace beam {

constexpr double kNegInf = -__builtin_huge_val();

// exp(d) for d <= 0.  k = rint(d * log2(e)); r = d - k*ln2 (two-piece constant); Taylor polynomial of degree 13 in
// Horner form; scaled by 2^k through the exponent field.  Below -700 the result cannot change 1 + exp(d): 0.
OCRS_HD double exp_nonpos(double d) {
    if (!(d > -700.0)) return 0.0;
    const double kf = __builtin_rint(d * 1.4426950408889634);
    double r = d - kf * 0.6931471803691238;          // ln2 high part (32 significant bits)
    r = r - kf * 1.9082149292705877e-10;             // ln2 low part
    double p = 1.6059043836821613e-10;               // 1/13!
    p = p * r + 2.08767569878681e-09;                // 1/12!
    p = p * r + 2.505210838544172e-08;               // 1/11!
    p = p * r + 2.755731922398589e-07;               // 1/10!
    p = p * r + 2.7557319223985893e-06;              // 1/9!
    p = p * r + 2.48015873015873e-05;                // 1/8!
    p = p * r + 0.0001984126984126984;               // 1/7!
    p = p * r + 0.001388888888888889;                // 1/6!
    p = p * r + 0.008333333333333333;                // 1/5!
    p = p * r + 0.041666666666666664;                // 1/4!
    p = p * r + 0.16666666666666666;                 // 1/3!
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    const int64_t k = (int64_t)kf;                   // -1010 <= k <= 0
    uint64_t bits = (uint64_t)(1023 + k) << 52;      // 2^k, a normal number
    double scale;
    __builtin_memcpy(&scale, &bits, sizeof scale);
    return p * scale;
}

// log(1 + x) for 0 <= x <= 1:  2 atanh(z), z = x / (2 + x) in [0, 1/3]; odd series up to z^33, Horner in z^2.
OCRS_HD double log1p_unit(double x) {
    const double z = x / (2.0 + x);
    const double z2 = z * z;
    double p = 1.0 / 33.0;
    p = p * z2 + 1.0 / 31.0;
    p = p * z2 + 1.0 / 29.0;
    p = p * z2 + 1.0 / 27.0;
    p = p * z2 + 1.0 / 25.0;
    p = p * z2 + 1.0 / 23.0;
    p = p * z2 + 1.0 / 21.0;
    p = p * z2 + 1.0 / 19.0;
    p = p * z2 + 1.0 / 17.0;
    p = p * z2 + 1.0 / 15.0;
    p = p * z2 + 1.0 / 13.0;
    p = p * z2 + 1.0 / 11.0;
    p = p * z2 + 1.0 / 9.0;
    p = p * z2 + 1.0 / 7.0;
    p = p * z2 + 1.0 / 5.0;
    p = p * z2 + 1.0 / 3.0;
    p = p * z2 + 1.0;
    return 2.0 * (z * p);
}

// log(exp(a) + exp(b)) with -inf as the empty sum; symmetric in (a, b)
OCRS_HD double lse(double a, double b) {
    if (a == kNegInf) return b;
    if (b == kNegInf) return a;
    const double m = a > b ? a : b;
    const double lo = a > b ? b : a;
    return m + log1p_unit(exp_nonpos(lo - m));
}

}  // namespace beam
}  // namespace ocrs
