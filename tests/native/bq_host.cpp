// Host build of the product's BOBYQA state machine (hpmvs_amd/csrc/bobyqa3.hpp) behind a callback
// interface, so CPU tests can pin it against the genuine NLopt library (oracle/_ref).
#include <cstring>
#include <vector>
static int g_rescue_calls = 0;
#define BQ_ON_RESCUE() (++g_rescue_calls)
#include "../../hpmvs_amd/csrc/cr_acos.hpp"
#include "../../hpmvs_amd/csrc/bobyqa3.hpp"

using namespace hpmvs;
typedef double (*obj_fn)(unsigned n, const double* x, double* grad, void* data);

extern "C" int bq3_minimize(obj_fn f, void* data, double* x, const double* lb, const double* ub, double xtol_rel,
                            int maxeval, double* minf, int* nevals, double* trace, int trace_cap) {
    Bobyqa3 B;
    double dx[3];
    bq_default_step(x, lb, ub, dx);
    int st = bq_start(B, x, lb, ub, dx, xtol_rel, maxeval);
    int nt = 0;
    while (st == BQ_CONTINUE) {
        const double xe[3] = {BQA(xeval, 0), BQA(xeval, 1), BQA(xeval, 2)};
        const double fv = f(3, xe, nullptr, data);
        if (trace && nt < trace_cap) { std::memcpy(trace + 4 * nt, xe, 24); trace[4 * nt + 3] = fv; nt++; }
        st = bq_step(B, fv);
    }
    for (int i = 0; i < 3; i++) x[i] = BQA(xfinal, i);
    *minf = BQS(minf);
    *nevals = BQN(nevals);
    return BQN(rc);
}

// bq_prelim_points against the sequence of points the state machine itself produces: returns the number of the six points
// (2 .. 7) that are bitwise equal, evaluating `f` like bq3_minimize does
extern "C" int bq3_prelim_points_equal(obj_fn f, void* data, const double* x0, const double* lb, const double* ub, double xtol_rel) {
    Bobyqa3 B;
    double dx[3];
    bq_default_step(x0, lb, ub, dx);
    if (bq_start(B, x0, lb, ub, dx, xtol_rel, 1000) != BQ_CONTINUE) return -1;
    double pts[6][3];
    bq_prelim_points(B, pts);
    int same = 0, st = BQ_CONTINUE;
    for (int k = 0; k < 6 && st == BQ_CONTINUE; k++) {
        const double xe[3] = {BQA(xeval, 0), BQA(xeval, 1), BQA(xeval, 2)};
        st = bq_step(B, f(3, xe, nullptr, data));
        if (st != BQ_CONTINUE) break;
        same += (std::memcmp(&BQA(xeval, 0), pts[k], 8) == 0 && std::memcmp(&BQA(xeval, 1), pts[k] + 1, 8) == 0 &&
                 std::memcmp(&BQA(xeval, 2), pts[k] + 2, 8) == 0) ? 1 : 0;
    }
    return same;
}

extern "C" int bq3_rescue_calls(void) { return g_rescue_calls; }

// host build of the device's round-to-nearest acos (hpmvs_amd/csrc/cr_acos.hpp)
extern "C" void cr_acos_host(const double* u, const double* y0, double* out, long n) {
    for (long i = 0; i < n; i++) out[i] = cra::acos_cr(u[i], y0[i]);
}
