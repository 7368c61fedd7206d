// kernel_selftest.hip -- DIAGNOSTIC entry (hpmvs_selftest_bobyqa): the product's BOBYQA state machine
// (bobyqa3.hpp, the very code optimize_kernel runs) driven on the GPU by analytic objectives, 32 DIFFERENT problems
// per wavefront so that the lanes sit in different blocks of the optimiser exactly as in the product kernel
// (wave-aware label scheduler, divergent calls, LDS-resident [index][slot] state).  Its purpose is to execute on
// the device what the photometric objective never reaches: rescue_ (bobyqa.c:143-742), active bounds, the
// all-plateau case and ROUNDOFF_LIMITED.  tests/test_gpu_bobyqa_selftest.py compares every evaluation point with
// the reference's genuine NLopt BOBYQA run on the host on the same objectives (selftest_obj.h).
#include <hip/hip_runtime.h>

#define BQ_STATE_LDS 1
#define BQ_COLD_GLOBAL 1  // as in the product kernel's 64-slot build: bmat, zmat, vlag, wsuma in global memory
#define BQ_COLD_HI 145
#define BQ_P 32
#define BQ_SLOTS 32
namespace hpmvs { __shared__ int st_rescue_calls[BQ_SLOTS]; }
#define BQ_ON_RESCUE() (hpmvs::st_rescue_calls[B.slot]++)
#include "bobyqa3.hpp"
#include "selftest_obj.h"

namespace hpmvs {

__global__ void __launch_bounds__(64) selftest_kernel(int n, const int* kind, const double* params, const double* x0,
                                                      const double* lb, const double* ub, int maxeval, double* xfinal,
                                                      double* minf, int* rc, int* nevals, int* rescues, double* trace,
                                                      int trace_cap, double* cold) {
    const int lane = (int)__lane_id();
    const int prob = (int)blockIdx.x * BQ_P + lane;
    const bool active = lane < BQ_P && prob < n;
    Bobyqa3 B;
    B.slot = lane < BQ_P ? lane : 0;
    B.slot_i = B.slot;
    B.coff = (unsigned)B.slot * 8u;
    B.coldp = (char*)(cold + (size_t)blockIdx.x * BQ_NCOLD * BQ_P);
    B.cold = __builtin_amdgcn_make_buffer_rsrc(B.coldp, 0, BQ_NCOLD * BQ_CSTRIDE * 8, 0x00020000);
    if (lane < BQ_P) st_rescue_calls[lane] = 0;
    int st = BQ_DONE, k = 0, nt = 0;
    double p[ST_PARAMS];
    for (int i = 0; i < ST_PARAMS; i++) p[i] = 0.0;
    if (active) {
        k = kind[prob];
        for (int i = 0; i < ST_PARAMS; i++) p[i] = params[(size_t)prob * ST_PARAMS + i];
        const double xs[3] = {x0[3 * prob], x0[3 * prob + 1], x0[3 * prob + 2]};
        const double l[3] = {lb[3 * prob], lb[3 * prob + 1], lb[3 * prob + 2]};
        const double u[3] = {ub[3 * prob], ub[3 * prob + 1], ub[3 * prob + 2]};
        double dx[3];
        bq_default_step(xs, l, u, dx);
        st = bq_start(B, xs, l, u, dx, 1.e-7, maxeval);
    }
    while (__any(st == BQ_CONTINUE)) {
        if (st == BQ_CONTINUE) {  // divergent, like the product kernel's per-lane stage switch
            const double xe[3] = {BQA(xeval, 0), BQA(xeval, 1), BQA(xeval, 2)};
            const double f = selftest_obj(k, p, xe);
            if (trace && nt < trace_cap) {
                double* t = trace + ((size_t)prob * trace_cap + nt) * 4;
                t[0] = xe[0]; t[1] = xe[1]; t[2] = xe[2]; t[3] = f;
            }
            nt++;
            st = bq_step(B, f);
        }
    }
    if (active) {
        for (int i = 0; i < 3; i++) xfinal[3 * prob + i] = BQA(xfinal, i);
        minf[prob] = BQS(minf);
        rc[prob] = BQN(rc);
        nevals[prob] = BQN(nevals);
        rescues[prob] = st_rescue_calls[lane];
    }
}

void launch_selftest(int n, const int* kind, const double* params, const double* x0, const double* lb, const double* ub,
                     int maxeval, double* xfinal, double* minf, int* rc, int* nevals, int* rescues, double* trace,
                     int trace_cap, double* cold, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(selftest_kernel, dim3((n + BQ_P - 1) / BQ_P), dim3(64), 0, st, n, kind, params, x0, lb, ub, maxeval,
                       xfinal, minf, rc, nevals, rescues, trace, trace_cap, cold);
}
size_t selftest_cold_doubles(int n) { return (size_t)((n + BQ_P - 1) / BQ_P) * BQ_NCOLD * BQ_P; }

}  // namespace hpmvs
