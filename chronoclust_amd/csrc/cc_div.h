// chronoclust_amd/csrc: IEEE double division of several numerators by ONE denominator.
//
// The reference divides CF1 and CF2 of every dimension by the same weight (microcluster.py:213-233 through
// mc_functions.py:14-22): 2 d quotients per tentative add.  The compiler expands each `x / y` on gfx950 into
//     ys = div_scale(y), xs = div_scale(x); r = rcp(ys); two Newton steps on r (four FMAs);
//     q = xs * r; e = fma(-ys, q, xs); result = div_fixup(div_fmas(e, r, q))
// - fourteen instructions, nine of which depend on y alone.  When neither operand needs scaling (div_scale returns its
// operand unchanged, div_fmas is a plain FMA, div_fixup passes the value through) the quotient is
//     fma(fma(-y, x * r, x), r, x * r)      with r = the twice-refined reciprocal of y
// bit for bit, so r can be computed once per denominator (cc_div_prepare) and each further quotient costs three
// instructions (cc_div_apply).  cc_div_fast_ok states when no scaling happens (V_DIV_SCALE_F64 in the CDNA3 ISA guide:
// it scales for a denormal or huge denominator, for a numerator with a biased exponent <= 53 and for quotients near the
// ends of the exponent range); callers test it and take the compiler's division otherwise.  tests/test_div_exact.py
// compares the two forms on the GPU over random and edge-case operands.
#pragma once

// the twice-refined reciprocal the compiler's expansion multiplies the numerator with
__device__ __forceinline__ double cc_div_prepare(double y)
{
    double r = __builtin_amdgcn_rcp(y);
    double e = __builtin_fma(-y, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-y, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}

// x / y given r = cc_div_prepare(y); exact (== x / y) under cc_div_fast_ok(x, y) or x == +0.0
__device__ __forceinline__ double cc_div_apply(double x, double y, double r)
{
    const double q = x * r;
    const double e = __builtin_fma(-y, q, x);
    return __builtin_fma(e, r, q);
}

// denominators 1 <= y < 2^60 (weights: counts of points, possibly decayed - the callers pass y >= 1 only), numerators
// 2^-900 <= |x| <= 2^700 or +0.0: no operand is scaled, no intermediate leaves the normal range
__device__ __forceinline__ bool cc_div_den_ok(double y) { return y >= 1.0 && y < 0x1p60; }
__device__ __forceinline__ bool cc_div_num_ok(double x)
{
    const double a = __builtin_fabs(x);
    return (a >= 0x1p-900 && a <= 0x1p700) || (a == 0.0 && !__builtin_signbit(x));
}
