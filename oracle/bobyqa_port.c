/*
 * bobyqa_port.c -- CPU ORACLE (test infrastructure, NOT product code).
 * Own C port of NLopt 2.4.2 BOBYQA (placeholder until the port lands: the
 * oracle currently runs the genuine library from oracle/_ref).
 */
#include "hpmvs_oracle.h"
int orc_port_bobyqa_available(void) { return 0; }
int orc_port_bobyqa(int n, int npt, double *x, const double *lb, const double *ub, const double *dx,
                    double xtol_rel, int maxeval, double *minf, int *nevals, orc_obj_fn f, void *data) {
    (void)n; (void)npt; (void)x; (void)lb; (void)ub; (void)dx; (void)xtol_rel; (void)maxeval;
    (void)minf; (void)nevals; (void)f; (void)data;
    return -1;
}
