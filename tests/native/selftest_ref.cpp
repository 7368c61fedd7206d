// Host side of the BOBYQA self-test: the analytic objectives of hpmvs_amd/csrc/selftest_obj.h as an NLopt-style C
// callback (so that the reference's genuine NLopt BOBYQA, oracle/_ref, can be run on them) with a trace of every
// evaluation.  Built with -ffp-contract=off like the device code: bit-identical objective values on both sides.
#include <cstring>

#include "../../hpmvs_amd/csrc/selftest_obj.h"

extern "C" {

struct selftest_problem {
    int kind;
    int n_trace, trace_cap;
    double p[ST_PARAMS];
    double* trace;  // trace_cap x 4: x0 x1 x2 f
};

double selftest_callback(unsigned n, const double* x, double* grad, void* data) {
    (void)n; (void)grad;
    selftest_problem* pr = (selftest_problem*)data;
    const double f = selftest_obj(pr->kind, pr->p, x);
    if (pr->trace && pr->n_trace < pr->trace_cap) {
        double* t = pr->trace + 4 * (size_t)pr->n_trace;
        t[0] = x[0]; t[1] = x[1]; t[2] = x[2]; t[3] = f;
    }
    pr->n_trace++;
    return f;
}

void* selftest_callback_address(void) { return (void*)&selftest_callback; }

}  // extern "C"
