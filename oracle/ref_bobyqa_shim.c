/*
 * ref_bobyqa_shim.c -- CPU ORACLE glue (test infrastructure, NOT product code).
 *
 * Compiled ONLY together with the reference's genuine, unmodified
 *   thirdLibs/nlopt-2.4.2/bobyqa/bobyqa.c, util/stop.c, util/rescale.c
 * (from where they lie under /root/reference) into oracle/_ref/libref_bobyqa.so.
 * It fills the nlopt_stopping record exactly as nlopt_optimize_ does
 * (thirdLibs/nlopt-2.4.2/api/optimize.c:398-408) and calls the genuine
 * bobyqa() (bobyqa/bobyqa.c:3073).  The two helpers below are link glue for
 * symbols that live in reference TUs which do not compile without the
 * generated config.h (api/general.c needs MAJOR_VERSION, util/timer.c needs
 * HAVE_GETTIMEOFDAY): they carry no optimiser arithmetic.
 */
#include <math.h>
#include <stdlib.h>
#include "bobyqa.h" /* reference header: nlopt_stopping, nlopt_func, bobyqa() */

/* api/general.c:30-36 (HAVE_ISINF undefined branch) */
int nlopt_isinf(double x) { return fabs(x) >= HUGE_VAL * 0.99; }
/* util/timer.c: only feeds maxtime, which hpmvs never sets (maxtime = 0) */
double nlopt_seconds(void) { return 0.0; }

int ref_bobyqa_run(int n, int npt, double *x, const double *lb, const double *ub, const double *dx,
                   double xtol_rel, int maxeval, double *minf, int *nevals, nlopt_func f, void *data) {
    nlopt_stopping stop;
    double *xtol_abs = (double *)calloc((size_t)n, sizeof(double));
    int force_stop = 0;
    stop.n = (unsigned)n;
    stop.minf_max = -HUGE_VAL; /* api/options.c: stopval default */
    stop.ftol_rel = 0.0;
    stop.ftol_abs = 0.0;
    stop.xtol_rel = xtol_rel;
    stop.xtol_abs = xtol_abs;
    stop.nevals = 0;
    stop.maxeval = maxeval;
    stop.maxtime = 0.0;
    stop.start = 0.0;
    stop.force_stop = &force_stop;
    nlopt_result r = bobyqa(n, npt, x, lb, ub, dx, &stop, minf, f, data);
    *nevals = stop.nevals;
    free(xtol_abs);
    return (int)r;
}
