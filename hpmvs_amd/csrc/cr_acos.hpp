// cr_acos.hpp -- double acos rounded to nearest, for the BOBYQA start point.
//
// parametersFromCenterNorm (reference src/hpmvs/PatchOptimizer.cpp:416-446) computes
// x[1] = acos(u) / angleScale in float64 with the host C library and hands it to NLopt as is.
// BOBYQA on this objective is chaotic (DESIGN.md "Why the objective is bit-exact"): a start point that
// is one ulp off changes the final float centre/normal of about one patch in 400.  glibc 2.35's acos
// (sysdeps/ieee754/dbl-64/e_asin.c, error bound 0.5x ulp) returns the correctly rounded value for all
// but ~0.06% of arguments, while the device math library's acos is allowed a few ulp.  So the device
// refines its library value y0 with one Newton step on cos carried in double-double arithmetic
// (error ~1e-30, far below the distance to a rounding boundary that matters) and rounds once:
//     y = y0 + (cos(y0) - u) / sin(y0)
// tests/test_cpu_oracle.py::test_cr_acos_matches_host_libm measures the agreement with the host libm
// from deliberately perturbed y0 (>= 99.9%).  Pure arithmetic (+, *, /, fma): identical on host and
// device.
#pragma once
#include <math.h>
#if defined(__HIPCC__)
#define CRA_HD __host__ __device__ inline
#else
#define CRA_HD inline
#endif

namespace cra {

struct dd {
    double hi, lo;
};

CRA_HD dd two_sum(double a, double b) {
    const double s = a + b, bb = s - a;
    return {s, (a - (s - bb)) + (b - bb)};
}
CRA_HD dd quick_two_sum(double a, double b) {
    const double s = a + b;
    return {s, b - (s - a)};
}
CRA_HD dd two_prod(double a, double b) {
    const double p = a * b;
    return {p, fma(a, b, -p)};
}
CRA_HD dd add(dd a, dd b) {
    dd s = two_sum(a.hi, b.hi);
    s.lo += a.lo + b.lo;
    return quick_two_sum(s.hi, s.lo);
}
CRA_HD dd add_d(dd a, double b) {
    dd s = two_sum(a.hi, b);
    s.lo += a.lo;
    return quick_two_sum(s.hi, s.lo);
}
CRA_HD dd mul(dd a, dd b) {
    dd p = two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return quick_two_sum(p.hi, p.lo);
}
CRA_HD dd neg(dd a) { return {-a.hi, -a.lo}; }
// a / den for a small exactly representable den
CRA_HD dd div_d(dd a, double den) {
    const double q = a.hi / den;
    const dd p = two_prod(q, den);
    const double r = ((a.hi - p.hi) - p.lo + a.lo) / den;
    return quick_two_sum(q, r);
}

// sin and cos of a double-double |x| <= pi/8 + eps: Taylor series, 13 terms each (x^27/27! < 1e-38)
CRA_HD void sincos_small(dd x, dd& s, dd& c) {
    const dd x2 = mul(x, x);
    dd ts = x, tc = {1.0, 0.0};
    s = x;
    c = tc;
#pragma unroll 1
    for (int k = 1; k <= 13; k++) {
        const double k2 = 2.0 * k;
        tc = neg(div_d(mul(tc, x2), (k2 - 1.0) * k2));
        c = add(c, tc);
        ts = neg(div_d(mul(ts, x2), k2 * (k2 + 1.0)));
        s = add(s, ts);
    }
}

// sin(y), cos(y) for a double y in [0, pi], argument reduced by multiples of pi/4 held in double-double
CRA_HD void sincos_0_pi(double y, dd& s, dd& c) {
    const dd rt2 = {0.70710678118654757, -4.8336466567264567e-17};  // sqrt(1/2)
    int oct = (int)((y + 0.39269908169872414) / 0.78539816339744828);  // nearest multiple of pi/4
    oct = oct < 0 ? 0 : (oct > 4 ? 4 : oct);
    // -oct * pi/4 as double-double
    dd shift = {0.0, 0.0};
    if (oct == 1) shift = {-0.78539816339744828, -3.0616169978683830e-17};
    else if (oct == 2) shift = {-1.5707963267948966, -6.1232339957367660e-17};
    else if (oct == 3) shift = {-2.3561944901923448, -9.1848509936051484e-17};
    else if (oct == 4) shift = {-3.1415926535897931, -1.2246467991473532e-16};
    dd sr, cr;
    sincos_small(add_d(shift, y), sr, cr);
    switch (oct) {
        case 0: s = sr; c = cr; break;
        case 1: s = mul(rt2, add(sr, cr)); c = mul(rt2, add(cr, neg(sr))); break;
        case 2: s = cr; c = neg(sr); break;
        case 3: s = mul(rt2, add(cr, neg(sr))); c = neg(mul(rt2, add(sr, cr))); break;
        default: s = neg(sr); c = neg(cr); break;
    }
}

// acos(u) rounded to nearest for |u| <= 1, given any approximation y0 accurate to a few ulp
CRA_HD double acos_cr(double u, double y0) {
    if (u == 1.0) return 0.0;
    if (u == -1.0) return 3.141592653589793;
    if (!(u < 1.0 && u > -1.0)) return y0;
    dd s, c;
    sincos_0_pi(y0, s, c);
    const dd num = add_d(c, -u);
    const double delta = (num.hi + num.lo) / (s.hi + s.lo);  // |delta| ~ ulp(y0); the y0*delta^2 term is < 1e-30
    return y0 + delta;
}

}  // namespace cra
