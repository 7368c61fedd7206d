/*
 * bobyqa_port.cpp -- the optimiser that the oracle's ORC_OPT_PORT names (test infrastructure; built as tests/native/libbq_port.so,
 * NOT linked into oracle/liboracle.so: the oracle itself holds no product code and loads this at run time).
 *
 * Used only when the reference's genuine NLopt BOBYQA (oracle/_ref/libref_bobyqa.so, built from
 * /root/reference by oracle/Makefile) is not present on the box.  It is a host build of the BOBYQA
 * restatement in hpmvs_amd/csrc/bobyqa3.hpp (restates thirdLibs/nlopt-2.4.2/bobyqa/bobyqa.c:18-3268,
 * n = 3 only) driven through the same callback interface.  tests/test_cpu_oracle.py proves that this
 * code produces bit-identical evaluation points, minima, result codes and evaluation counts as the
 * genuine library, and tests/golden/ pins it on boxes where the genuine library is absent.  Because
 * it shares its source with the product's optimiser, results obtained with it are labelled
 * "oracle BOBYQA port" wherever they are reported.
 */
#include "../../oracle/hpmvs_oracle.h"

#include "../../hpmvs_amd/csrc/bobyqa3.hpp"

using namespace hpmvs;

extern "C" int orc_port_bobyqa_available(void) { return 1; }

extern "C" int orc_port_bobyqa(int n, int npt, double *x, const double *lb, const double *ub, const double *dx,
                               double xtol_rel, int maxeval, double *minf, int *nevals, orc_obj_fn f, void *data) {
    if (n != 3 || npt != 7) return -2;
    Bobyqa3 B;
    int st = bq_start(B, x, lb, ub, dx, xtol_rel, maxeval);
    while (st == BQ_CONTINUE) {
        const double xe[3] = {BQA(xeval, 0), BQA(xeval, 1), BQA(xeval, 2)};
        st = bq_step(B, f(3, xe, nullptr, data));
    }
    for (int i = 0; i < 3; i++) x[i] = BQA(xfinal, i);
    *minf = BQS(minf);
    *nevals = BQN(nevals);
    return BQN(rc);
}
