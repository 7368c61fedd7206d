// selftest_obj.h -- analytic 3-variable objectives for the BOBYQA self-test (hpmvs_selftest_bobyqa and the host
// reference run of tests/test_gpu_bobyqa_selftest.py).  Only + - * / sqrt floor fabs: every operation is
// correctly rounded on the host and on the GPU alike (both built with -ffp-contract=off), so the two sides see
// bit-identical objective values and any difference in the trajectories is a difference in the optimiser.
#pragma once
#if defined(__HIPCC__)
#define ST_HD __host__ __device__ inline
#else
#define ST_HD static inline
#endif
#include <math.h>

enum { ST_PARAMS = 8 };

ST_HD double selftest_obj(int kind, const double* p, const double* x) {
    switch (kind) {
    case 0: {  // ill-conditioned quadratic + sawtooth "noise" (rounding-error stand-in): drives bobyqb_ into rescue_
        const double t = x[0] * 1e6 + x[1] * 1e5;
        const double saw = t - floor(t);
        const double d0 = x[0] - p[0], d1 = x[1] - p[1], d2 = x[2] - p[2];
        return p[3] * (d0 * d0) + p[4] * (d1 * d1) + p[5] * (d2 * d2) + p[6] * saw;
    }
    case 1: {  // Rosenbrock chain (NLopt test/testfuncs.c style polynomial)
        const double a = x[1] - x[0] * x[0], b = 1.0 - x[0], c = x[2] - x[1] * x[1], d = 1.0 - x[1];
        return p[3] * (a * a) + b * b + p[3] * (c * c) + d * d;
    }
    case 2:  // the photometric objective's plateau value everywhere
        return 2.0;
    case 3: {  // steep one-sided growth (8th power of a ramp) + product coupling: the other rescue_ family
        double r = 1.0 + p[3] * 1e-3 * (x[0] - p[0]);
        if (r < 0.0) r = 0.0;
        const double r2 = r * r, r4 = r2 * r2;
        const double q = x[1] * x[2] - p[1];
        return r4 * r4 + q * q;
    }
    case 4: {  // |.| valley + sqrt term: non-smooth, bound-active when p[1], p[2] lie outside the box
        const double d1 = x[1] - p[1], d2 = x[2] - p[2];
        const double w = x[0] * x[1];
        return fabs(x[0] - p[0]) * p[3] + p[4] * (d1 * d1) + p[5] * (d2 * d2) + p[6] * sqrt(1.0 + w * w);
    }
    default: {  // plain bowl whose minimum lies far outside the angle bounds: ends ROUNDOFF_LIMITED / at the bounds
        const double d0 = x[0] - p[0], d1 = x[1] - p[1], d2 = x[2] - p[2];
        return d0 * d0 + d1 * d1 + d2 * d2;
    }
    }
}
