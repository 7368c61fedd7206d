// bobyqa3.hpp -- resumable, fixed-size BOBYQA (n = 3 variables, npt = 2n+1 = 7 interpolation
// points) for the batched on-GPU refinement of patch depth + normal.
//
// The reference refines every patch with NLopt 2.4.2's LN_BOBYQA through a C callback
// (reference src/hpmvs/PatchOptimizer.cpp:348-365 -> thirdLibs/nlopt-2.4.2/api/optimize.c:669-681 ->
// bobyqa/bobyqa.c:3073).  A GPU lane cannot call back into a wave-cooperative objective, so the
// same method (M.J.D. Powell's bound-constrained quadratic-interpolation trust-region algorithm
// as NLopt configures it) is restated here as an inversion-of-control STATE MACHINE:
//
//     bq_start(B, ...)  -> B.xeval = first point to evaluate
//     bq_step(B, f)     -> consumes f(B.xeval); returns BQ_CONTINUE with the next B.xeval, or
//                          BQ_DONE with B.rc (nlopt_result code), B.minf, B.xfinal, B.nevals
//
// so that 64 lanes of a wavefront each advance their own patch's optimiser between
// wave-cooperative objective evaluations.  Everything is float64 like the reference; the code
// compiles for the host as well (tests pin it evaluation-by-evaluation against the genuine
// library built under oracle/_ref).
//
// Structure follows the published algorithm (Powell 2009, "The BOBYQA algorithm for bound
// constrained optimization without derivatives") as laid out in bobyqa.c; cited line ranges:
//   bq_start          <- bobyqa()   bobyqa/bobyqa.c:3073-3268 (+ util/rescale.c:29-82,
//                        rhoend from xtol_rel: bobyqa.c:3121)
//   prelim_*          <- prelim_    bobyqa.c:1713-1952
//   trsbox            <- trsbox_    bobyqa.c:1162-1711
//   altmov            <- altmov_    bobyqa.c:744-1160
//   update            <- update_    bobyqa.c:18-141
//   run / after_eval  <- bobyqb_    bobyqa.c:1954-3055
// rescue_ (bobyqa.c:143-742) is NOT restated: where the reference would call it the run stops
// with rc = BQ_NEED_RESCUE and the caller reports the patch as failed with that distinct code
// (measured frequency on the benchmark scenes: see DESIGN.md).
//
// Indexing: the accessor macros are 1-based like the mathematics; storage is dense 0-based.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BQ_HD __host__ __device__ inline
// the big routines are real calls on the GPU: their register demand stays out of the caller's
// wave-cooperative sampling loop (occupancy), and the code is shared by the call sites
#define BQ_HD_NOINLINE __host__ __device__ __attribute__((noinline))
#else
#define BQ_HD inline
#define BQ_HD_NOINLINE inline
#endif
#include <math.h>

namespace hpmvs {

enum { BQ_CONTINUE = 0, BQ_DONE = 1 };
// nlopt_result codes (thirdLibs/nlopt-2.4.2/api/nlopt.h) + one private code
enum {
    BQ_FAILURE = -1, BQ_INVALID_ARGS = -2, BQ_ROUNDOFF_LIMITED = -4,
    BQ_SUCCESS = 1, BQ_XTOL_REACHED = 4, BQ_MAXEVAL_REACHED = 5,
    BQ_NEED_RESCUE = -100
};

struct Bobyqa3 {
    static constexpr int N = 3, NPT = 7, NP = 4, NPTM = 3, NDIM = 10;
    // ---- outputs / interface
    double xeval[N];   // point to evaluate (original, unscaled variables)
    double xfinal[N];
    double minf;
    int rc;
    int nevals;
    // ---- problem
    double sc[N];          // variable scaling (dx_i / dx_0)
    double xl[N], xu[N];   // bounds in scaled variables
    double rhobeg, rhoend;
    int maxeval;
    // ---- model + geometry state (names as in Powell's paper / bobyqb_)
    double x[N];
    double xbase[N], xopt[N], gopt[N], sl[N], su[N], xnew[N], xalt[N], d[N], gnew[N];
    double xpt[NPT * N];
    double fval[NPT], pq[NPT];
    double hq[6];
    double bmat[NDIM * N];
    double zmat[NPT * NPTM];
    double vlag[NDIM];
    double wsuma[NPT];     // w(npt+k) of bobyqb_: (xpt_k . d), needed after the evaluation
    // ---- scalars that live across evaluations
    double rho, delta, diffa, diffb, diffc, dsq, crvmin, dnorm, xoptsq, ratio, adelt, alpha, cauchy, beta,
        denom, fsave, distsq, fbeg, stepa, stepb;
    int kopt, knew, kbase, ntrits, itest, nresc, nfsav, nf, phase;
};

#define BQ_XPT(k, i) B.xpt[((k) - 1) * 3 + ((i) - 1)]
#define BQ_BMAT(k, i) B.bmat[((k) - 1) * 3 + ((i) - 1)]
#define BQ_ZMAT(k, j) B.zmat[((k) - 1) * 3 + ((j) - 1)]
#define BQ_FVAL(k) B.fval[(k) - 1]
#define BQ_PQ(k) B.pq[(k) - 1]
#define BQ_HQ(ih) B.hq[(ih) - 1]
#define BQ_V(a, i) B.a[(i) - 1]

namespace bq {

BQ_HD double mn(double a, double b) { return a <= b ? a : b; }
BQ_HD double mx(double a, double b) { return a >= b ? a : b; }
BQ_HD bool is_inf(double v) { return fabs(v) >= HUGE_VAL * 0.99; }

enum { PH_PRELIM = 0, PH_MAIN = 1, PH_DONE = 2 };
enum { L20 = 0, L60, L90, L210, L230, L360, L650, L680, L720 };

// x (scaled) for the interpolation point / trial step `p` (relative to xbase), bounds respected
// exactly as bobyqb_ does before each calfun (bobyqa.c:2560-2575, 1871-1883)
BQ_HD void set_x_from(Bobyqa3& B, const double* p) {
    for (int i = 1; i <= 3; i++) {
        double v = mn(mx(BQ_V(xl, i), BQ_V(xbase, i) + p[i - 1]), BQ_V(xu, i));
        if (p[i - 1] == BQ_V(sl, i)) v = BQ_V(xl, i);
        if (p[i - 1] == BQ_V(su, i)) v = BQ_V(xu, i);
        BQ_V(x, i) = v;
    }
    for (int i = 0; i < 3; i++) B.xeval[i] = B.x[i] * B.sc[i];  // nlopt_unscale (util/rescale.c:55-60)
}

// ---------------------------------------------------------------- prelim_ (bobyqa.c:1713-1952)
// Choose interpolation point nf+1 (1-based after the increment) and put it in B.x / B.xeval.
BQ_HD void prelim_next_point(Bobyqa3& B) {
    const int nfm = B.nf;
    const int nfx = B.nf - 3;
    B.nf++;
    const int nf = B.nf;
    // with npt = 2n+1 only the first two families of points occur (nfm <= 2n)
    if (nfm >= 1 && nfm <= 3) {
        B.stepa = B.rhobeg;
        if (BQ_V(su, nfm) == 0.0) B.stepa = -B.stepa;
        BQ_XPT(nf, nfm) = B.stepa;
    } else if (nfm > 3) {
        B.stepa = BQ_XPT(nf - 3, nfx);
        B.stepb = -B.rhobeg;
        if (BQ_V(sl, nfx) == 0.0) B.stepb = mn(2.0 * B.rhobeg, BQ_V(su, nfx));
        if (BQ_V(su, nfx) == 0.0) B.stepb = mx(-2.0 * B.rhobeg, BQ_V(sl, nfx));
        BQ_XPT(nf, nfx) = B.stepb;
    }
    set_x_from(B, &BQ_XPT(nf, 1));
}

// Absorb f at point nf into the initial model.
BQ_HD void prelim_consume(Bobyqa3& B, double f) {
    const int nf = B.nf, nfm = nf - 1, nfx = nf - 1 - 3;
    const double rhosq = B.rhobeg * B.rhobeg;
    BQ_FVAL(nf) = f;
    if (nf == 1) {
        B.fbeg = f;
        B.kopt = 1;
    } else if (f < BQ_FVAL(B.kopt)) {
        B.kopt = nf;
    }
    if (nf >= 2 && nf <= 4) {
        BQ_V(gopt, nfm) = (f - B.fbeg) / B.stepa;
        // (npt < nf + n never holds for npt = 2n+1)
    } else if (nf >= 5) {
        const int ih = nfx * (nfx + 1) / 2;
        const double temp = (f - B.fbeg) / B.stepb;
        const double diff = B.stepb - B.stepa;
        BQ_HQ(ih) = 2.0 * (temp - BQ_V(gopt, nfx)) / diff;
        BQ_V(gopt, nfx) = (BQ_V(gopt, nfx) * B.stepb - temp * B.stepa) / diff;
        if (B.stepa * B.stepb < 0.0) {
            if (f < BQ_FVAL(nf - 3)) {
                BQ_FVAL(nf) = BQ_FVAL(nf - 3);
                BQ_FVAL(nf - 3) = f;
                if (B.kopt == nf) B.kopt = nf - 3;
                BQ_XPT(nf - 3, nfx) = B.stepb;
                BQ_XPT(nf, nfx) = B.stepa;
            }
        }
        BQ_BMAT(1, nfx) = -(B.stepa + B.stepb) / (B.stepa * B.stepb);
        BQ_BMAT(nf, nfx) = -0.5 / BQ_XPT(nf - 3, nfx);
        BQ_BMAT(nf - 3, nfx) = -BQ_BMAT(1, nfx) - BQ_BMAT(nf, nfx);
        BQ_ZMAT(1, nfx) = sqrt(2.0) / (B.stepa * B.stepb);
        BQ_ZMAT(nf, nfx) = sqrt(0.5) / rhosq;
        BQ_ZMAT(nf - 3, nfx) = -BQ_ZMAT(1, nfx) - BQ_ZMAT(nf, nfx);
    }
}

// ---------------------------------------------------------------- H times s (bobyqa.c:1676-1700)
BQ_HD void hess_mul(const Bobyqa3& B, const double* s, double* hs) {
    int ih = 0;
    for (int j = 1; j <= 3; j++) {
        hs[j - 1] = 0.0;
        for (int i = 1; i <= j; i++) {
            ++ih;
            if (i < j) hs[j - 1] += B.hq[ih - 1] * s[i - 1];
            hs[i - 1] += B.hq[ih - 1] * s[j - 1];
        }
    }
    for (int k = 1; k <= 7; k++) {
        if (B.pq[k - 1] != 0.0) {
            double temp = 0.0;
            for (int j = 1; j <= 3; j++) temp += B.xpt[(k - 1) * 3 + j - 1] * s[j - 1];
            temp *= B.pq[k - 1];
            for (int i = 1; i <= 3; i++) hs[i - 1] += temp * B.xpt[(k - 1) * 3 + i - 1];
        }
    }
}

// ---------------------------------------------------------------- trsbox_ (bobyqa.c:1162-1711)
// Approximately minimise the model within ||d|| <= delta and the bounds: truncated conjugate
// gradients with restarts on newly active bounds, then 2-D searches on the boundary.
// Sets B.xnew, B.d, B.gnew, B.dsq, B.crvmin.
BQ_HD_NOINLINE void trsbox(Bobyqa3& B) {
    double xbdi[3], s[3] = {0, 0, 0}, hs[3], hred[3] = {0, 0, 0};
    double* d = B.d;
    double* gnew = B.gnew;
    const double* xopt = B.xopt;
    const double* sl = B.sl;
    const double* su = B.su;
    int iterc = 0, nact = 0, itermax = 0, iact = 0, itcsav = 0, isav, iu;
    double delsq, qred = 0.0, crvmin = -1.0, beta = 0.0, stepsq, gredsq = 0.0, resid, ds, shs, temp, blen, stplen,
                  sdec, ggsav = 0.0, dredsq = 0.0, dredg = 0.0, sredg = 0.0, angbd = 0.0, xsav = 0.0, ssq, tempa,
                  tempb, dhs, dhd, redmax, redsav, angt = 0.0, sth, cth, rednew, rdprev = 0.0, rdnext = 0.0;
    for (int i = 0; i < 3; i++) {
        xbdi[i] = 0.0;
        if (xopt[i] <= sl[i]) {
            if (B.gopt[i] >= 0.0) xbdi[i] = -1.0;
        } else if (xopt[i] >= su[i]) {
            if (B.gopt[i] <= 0.0) xbdi[i] = 1.0;
        }
        if (xbdi[i] != 0.0) ++nact;
        d[i] = 0.0;
        gnew[i] = B.gopt[i];
    }
    delsq = B.delta * B.delta;

    enum { CG_RESTART, CG_NEXT, BOUNDARY, ALT_SETUP, ALT_ITER, FINISH };
    int st = CG_RESTART;
    for (;;) {
        if (st == CG_RESTART) { beta = 0.0; st = CG_NEXT; }
        if (st == CG_NEXT) {
            stepsq = 0.0;
            for (int i = 0; i < 3; i++) {
                if (xbdi[i] != 0.0) s[i] = 0.0;
                else if (beta == 0.0) s[i] = -gnew[i];
                else s[i] = beta * s[i] - gnew[i];
                stepsq += s[i] * s[i];
            }
            if (stepsq == 0.0) { st = FINISH; continue; }
            if (beta == 0.0) { gredsq = stepsq; itermax = iterc + 3 - nact; }
            if (gredsq * delsq <= qred * 1e-4 * qred) { st = FINISH; continue; }
            hess_mul(B, s, hs);
            resid = delsq; ds = 0.0; shs = 0.0;
            for (int i = 0; i < 3; i++)
                if (xbdi[i] == 0.0) { resid -= d[i] * d[i]; ds += s[i] * d[i]; shs += s[i] * hs[i]; }
            if (resid <= 0.0) { st = BOUNDARY; continue; }
            temp = sqrt(stepsq * resid + ds * ds);
            if (ds < 0.0) blen = (temp - ds) / stepsq;
            else blen = resid / (temp + ds);
            stplen = blen;
            if (shs > 0.0) stplen = mn(blen, gredsq / shs);
            iact = 0;
            for (int i = 0; i < 3; i++)
                if (s[i] != 0.0) {
                    const double xsum = xopt[i] + d[i];
                    if (s[i] > 0.0) temp = (su[i] - xsum) / s[i];
                    else temp = (sl[i] - xsum) / s[i];
                    if (temp < stplen) { stplen = temp; iact = i + 1; }
                }
            sdec = 0.0;
            if (stplen > 0.0) {
                ++iterc;
                temp = shs / stepsq;
                if (iact == 0 && temp > 0.0) {
                    crvmin = mn(crvmin, temp);
                    if (crvmin == -1.0) crvmin = temp;
                }
                ggsav = gredsq;
                gredsq = 0.0;
                for (int i = 0; i < 3; i++) {
                    gnew[i] += stplen * hs[i];
                    if (xbdi[i] == 0.0) gredsq += gnew[i] * gnew[i];
                    d[i] += stplen * s[i];
                }
                sdec = mx(stplen * (ggsav - 0.5 * stplen * shs), 0.0);
                qred += sdec;
            }
            if (iact > 0) {
                ++nact;
                xbdi[iact - 1] = 1.0;
                if (s[iact - 1] < 0.0) xbdi[iact - 1] = -1.0;
                delsq -= d[iact - 1] * d[iact - 1];
                if (delsq <= 0.0) { st = BOUNDARY; continue; }
                st = CG_RESTART;
                continue;
            }
            if (stplen < blen) {
                if (iterc == itermax) { st = FINISH; continue; }
                if (sdec <= qred * .01) { st = FINISH; continue; }
                beta = gredsq / ggsav;
                st = CG_NEXT;
                continue;
            }
            st = BOUNDARY;
        }
        if (st == BOUNDARY) { crvmin = 0.0; st = ALT_SETUP; }
        if (st == ALT_SETUP) {
            if (nact >= 2) { st = FINISH; continue; }
            dredsq = 0.0; dredg = 0.0; gredsq = 0.0;
            for (int i = 0; i < 3; i++) {
                if (xbdi[i] == 0.0) {
                    dredsq += d[i] * d[i];
                    dredg += d[i] * gnew[i];
                    gredsq += gnew[i] * gnew[i];
                    s[i] = d[i];
                } else s[i] = 0.0;
            }
            itcsav = iterc;
            hess_mul(B, s, hs);
            for (int i = 0; i < 3; i++) hred[i] = hs[i];
            st = ALT_ITER;
        }
        if (st == ALT_ITER) {
            ++iterc;
            temp = gredsq * dredsq - dredg * dredg;
            if (temp <= qred * 1e-4 * qred) { st = FINISH; continue; }
            temp = sqrt(temp);
            for (int i = 0; i < 3; i++) {
                if (xbdi[i] == 0.0) s[i] = (dredg * d[i] - dredsq * gnew[i]) / temp;
                else s[i] = 0.0;
            }
            sredg = -temp;
            angbd = 1.0;
            iact = 0;
            bool refix = false;
            for (int i = 0; i < 3; i++) {
                if (xbdi[i] == 0.0) {
                    tempa = xopt[i] + d[i] - sl[i];
                    tempb = su[i] - xopt[i] - d[i];
                    if (tempa <= 0.0) { ++nact; xbdi[i] = -1.0; refix = true; break; }
                    else if (tempb <= 0.0) { ++nact; xbdi[i] = 1.0; refix = true; break; }
                    ssq = d[i] * d[i] + s[i] * s[i];
                    temp = ssq - (xopt[i] - sl[i]) * (xopt[i] - sl[i]);
                    if (temp > 0.0) {
                        temp = sqrt(temp) - s[i];
                        if (angbd * temp > tempa) { angbd = tempa / temp; iact = i + 1; xsav = -1.0; }
                    }
                    temp = ssq - (su[i] - xopt[i]) * (su[i] - xopt[i]);
                    if (temp > 0.0) {
                        temp = sqrt(temp) + s[i];
                        if (angbd * temp > tempb) { angbd = tempb / temp; iact = i + 1; xsav = 1.0; }
                    }
                }
            }
            if (refix) { st = ALT_SETUP; continue; }
            hess_mul(B, s, hs);
            shs = 0.0; dhs = 0.0; dhd = 0.0;
            for (int i = 0; i < 3; i++)
                if (xbdi[i] == 0.0) { shs += s[i] * hs[i]; dhs += d[i] * hs[i]; dhd += d[i] * hred[i]; }
            redmax = 0.0; isav = 0; redsav = 0.0;
            iu = (int)(angbd * 17. + 3.1);
            for (int i = 1; i <= iu; i++) {
                angt = angbd * (double)i / (double)iu;
                sth = (angt + angt) / (1.0 + angt * angt);
                temp = shs + angt * (angt * dhd - dhs - dhs);
                rednew = sth * (angt * dredg - sredg - 0.5 * sth * temp);
                if (rednew > redmax) { redmax = rednew; isav = i; rdprev = redsav; }
                else if (i == isav + 1) rdnext = rednew;
                redsav = rednew;
            }
            if (isav == 0) { st = FINISH; continue; }
            if (isav < iu) {
                temp = (rdnext - rdprev) / (redmax + redmax - rdprev - rdnext);
                angt = angbd * ((double)isav + 0.5 * temp) / (double)iu;
            }
            cth = (1.0 - angt * angt) / (1.0 + angt * angt);
            sth = (angt + angt) / (1.0 + angt * angt);
            temp = shs + angt * (angt * dhd - dhs - dhs);
            sdec = sth * (angt * dredg - sredg - 0.5 * sth * temp);
            if (sdec <= 0.0) { st = FINISH; continue; }
            dredg = 0.0; gredsq = 0.0;
            for (int i = 0; i < 3; i++) {
                gnew[i] = gnew[i] + (cth - 1.0) * hred[i] + sth * hs[i];
                if (xbdi[i] == 0.0) {
                    d[i] = cth * d[i] + sth * s[i];
                    dredg += d[i] * gnew[i];
                    gredsq += gnew[i] * gnew[i];
                }
                hred[i] = cth * hred[i] + sth * hs[i];
            }
            qred += sdec;
            if (iact > 0 && isav == iu) { ++nact; xbdi[iact - 1] = xsav; st = ALT_SETUP; continue; }
            if (sdec > qred * .01) { st = ALT_ITER; continue; }
            st = FINISH;
        }
        if (st == FINISH) break;
    }
    B.dsq = 0.0;
    for (int i = 0; i < 3; i++) {
        B.xnew[i] = mx(mn(xopt[i] + d[i], su[i]), sl[i]);
        if (xbdi[i] == -1.0) B.xnew[i] = sl[i];
        if (xbdi[i] == 1.0) B.xnew[i] = su[i];
        d[i] = B.xnew[i] - xopt[i];
        B.dsq += d[i] * d[i];
    }
    B.crvmin = crvmin;
}

// ---------------------------------------------------------------- altmov_ (bobyqa.c:744-1160)
// Geometry-improving step for interpolation point knew: B.xnew (line through xopt and another
// point), B.xalt (constrained Cauchy step), B.alpha, B.cauchy.
BQ_HD_NOINLINE void altmov(Bobyqa3& B) {
    const double cst = 1.0 + sqrt(2.);
    double glag[3], hcol[7], w[3], wsave[3] = {0, 0, 0};
    const int knew = B.knew, kopt = B.kopt;
    const double adelt = B.adelt;
    for (int k = 1; k <= 7; k++) hcol[k - 1] = 0.0;
    for (int j = 1; j <= 3; j++) {
        const double temp = BQ_ZMAT(knew, j);
        for (int k = 1; k <= 7; k++) hcol[k - 1] += temp * BQ_ZMAT(k, j);
    }
    B.alpha = hcol[knew - 1];
    const double ha = 0.5 * B.alpha;
    for (int i = 1; i <= 3; i++) glag[i - 1] = BQ_BMAT(knew, i);
    for (int k = 1; k <= 7; k++) {
        double temp = 0.0;
        for (int j = 1; j <= 3; j++) temp += BQ_XPT(k, j) * BQ_V(xopt, j);
        temp = hcol[k - 1] * temp;
        for (int i = 1; i <= 3; i++) glag[i - 1] += temp * BQ_XPT(k, i);
    }
    double presav = 0.0, step = 0.0, stpsav = 0.0, vlag, temp, tempa, tempb, tempd, diff;
    int ksav = kopt, ibdsav = 0, isbd;
    for (int k = 1; k <= 7; k++) {
        if (k == kopt) continue;
        double dderiv = 0.0, distsq = 0.0;
        for (int i = 1; i <= 3; i++) {
            temp = BQ_XPT(k, i) - BQ_V(xopt, i);
            dderiv += glag[i - 1] * temp;
            distsq += temp * temp;
        }
        double subd = adelt / sqrt(distsq);
        double slbd = -subd;
        int ilbd = 0, iubd = 0;
        const double sumin = mn(1.0, subd);
        for (int i = 1; i <= 3; i++) {
            temp = BQ_XPT(k, i) - BQ_V(xopt, i);
            if (temp > 0.0) {
                if (slbd * temp < BQ_V(sl, i) - BQ_V(xopt, i)) { slbd = (BQ_V(sl, i) - BQ_V(xopt, i)) / temp; ilbd = -i; }
                if (subd * temp > BQ_V(su, i) - BQ_V(xopt, i)) { subd = mx(sumin, (BQ_V(su, i) - BQ_V(xopt, i)) / temp); iubd = i; }
            } else if (temp < 0.0) {
                if (slbd * temp > BQ_V(su, i) - BQ_V(xopt, i)) { slbd = (BQ_V(su, i) - BQ_V(xopt, i)) / temp; ilbd = i; }
                if (subd * temp < BQ_V(sl, i) - BQ_V(xopt, i)) { subd = mx(sumin, (BQ_V(sl, i) - BQ_V(xopt, i)) / temp); iubd = -i; }
            }
        }
        if (k == knew) {
            diff = dderiv - 1.0;
            step = slbd;
            vlag = slbd * (dderiv - slbd * diff);
            isbd = ilbd;
            temp = subd * (dderiv - subd * diff);
            if (fabs(temp) > fabs(vlag)) { step = subd; vlag = temp; isbd = iubd; }
            tempd = 0.5 * dderiv;
            tempa = tempd - diff * slbd;
            tempb = tempd - diff * subd;
            if (tempa * tempb < 0.0) {
                temp = tempd * tempd / diff;
                if (fabs(temp) > fabs(vlag)) { step = tempd / diff; vlag = temp; isbd = 0; }
            }
        } else {
            step = slbd;
            vlag = slbd * (1.0 - slbd);
            isbd = ilbd;
            temp = subd * (1.0 - subd);
            if (fabs(temp) > fabs(vlag)) { step = subd; vlag = temp; isbd = iubd; }
            if (subd > 0.5) {
                if (fabs(vlag) < .25) { step = 0.5; vlag = .25; isbd = 0; }
            }
            vlag *= dderiv;
        }
        temp = step * (1.0 - step) * distsq;
        const double predsq = vlag * vlag * (vlag * vlag + ha * temp * temp);
        if (predsq > presav) { presav = predsq; ksav = k; stpsav = step; ibdsav = isbd; }
    }
    for (int i = 1; i <= 3; i++) {
        temp = BQ_V(xopt, i) + stpsav * (BQ_XPT(ksav, i) - BQ_V(xopt, i));
        BQ_V(xnew, i) = mx(BQ_V(sl, i), mn(BQ_V(su, i), temp));
    }
    if (ibdsav < 0) BQ_V(xnew, -ibdsav) = BQ_V(sl, -ibdsav);
    if (ibdsav > 0) BQ_V(xnew, ibdsav) = BQ_V(su, ibdsav);

    const double bigstp = adelt + adelt;
    double csave = 0.0, wfixsq, ggfree, wsqsav, gw, curv, scale;
    for (int iflag = 0; iflag <= 1; iflag++) {
        wfixsq = 0.0; ggfree = 0.0;
        for (int i = 1; i <= 3; i++) {
            w[i - 1] = 0.0;
            tempa = mn(BQ_V(xopt, i) - BQ_V(sl, i), glag[i - 1]);
            tempb = mx(BQ_V(xopt, i) - BQ_V(su, i), glag[i - 1]);
            if (tempa > 0.0 || tempb < 0.0) { w[i - 1] = bigstp; ggfree += glag[i - 1] * glag[i - 1]; }
        }
        if (ggfree == 0.0) { B.cauchy = 0.0; return; }
        for (;;) {
            temp = adelt * adelt - wfixsq;
            if (!(temp > 0.0)) break;
            wsqsav = wfixsq;
            step = sqrt(temp / ggfree);
            ggfree = 0.0;
            for (int i = 1; i <= 3; i++) {
                if (w[i - 1] == bigstp) {
                    temp = BQ_V(xopt, i) - step * glag[i - 1];
                    if (temp <= BQ_V(sl, i)) { w[i - 1] = BQ_V(sl, i) - BQ_V(xopt, i); wfixsq += w[i - 1] * w[i - 1]; }
                    else if (temp >= BQ_V(su, i)) { w[i - 1] = BQ_V(su, i) - BQ_V(xopt, i); wfixsq += w[i - 1] * w[i - 1]; }
                    else ggfree += glag[i - 1] * glag[i - 1];
                }
            }
            if (!(wfixsq > wsqsav && ggfree > 0.0)) break;
        }
        gw = 0.0;
        for (int i = 1; i <= 3; i++) {
            if (w[i - 1] == bigstp) {
                w[i - 1] = -step * glag[i - 1];
                BQ_V(xalt, i) = mx(BQ_V(sl, i), mn(BQ_V(su, i), BQ_V(xopt, i) + w[i - 1]));
            } else if (w[i - 1] == 0.0) BQ_V(xalt, i) = BQ_V(xopt, i);
            else if (glag[i - 1] > 0.0) BQ_V(xalt, i) = BQ_V(sl, i);
            else BQ_V(xalt, i) = BQ_V(su, i);
            gw += glag[i - 1] * w[i - 1];
        }
        curv = 0.0;
        for (int k = 1; k <= 7; k++) {
            temp = 0.0;
            for (int j = 1; j <= 3; j++) temp += BQ_XPT(k, j) * w[j - 1];
            curv += hcol[k - 1] * temp * temp;
        }
        if (iflag == 1) curv = -curv;
        if (curv > -gw && curv < -cst * gw) {
            scale = -gw / curv;
            for (int i = 1; i <= 3; i++) {
                temp = BQ_V(xopt, i) + scale * w[i - 1];
                BQ_V(xalt, i) = mx(BQ_V(sl, i), mn(BQ_V(su, i), temp));
            }
            B.cauchy = (0.5 * gw * scale) * (0.5 * gw * scale);
        } else {
            B.cauchy = (gw + 0.5 * curv) * (gw + 0.5 * curv);
        }
        if (iflag == 0) {
            for (int i = 0; i < 3; i++) { glag[i] = -glag[i]; wsave[i] = B.xalt[i]; }
            csave = B.cauchy;
        }
    }
    if (csave > B.cauchy) {
        for (int i = 0; i < 3; i++) B.xalt[i] = wsave[i];
        B.cauchy = csave;
    }
}

// ---------------------------------------------------------------- update_ (bobyqa.c:18-141)
// Update BMAT and ZMAT for the move of interpolation point knew (uses vlag, beta, denom).
BQ_HD_NOINLINE void update(Bobyqa3& B) {
    const int knew = B.knew;
    double w[10];
    double ztest = 0.0;
    for (int k = 1; k <= 7; k++)
        for (int j = 1; j <= 3; j++) ztest = mx(ztest, fabs(BQ_ZMAT(k, j)));
    ztest *= 1e-20;
    for (int j = 2; j <= 3; j++) {
        if (fabs(BQ_ZMAT(knew, j)) > ztest) {
            double temp = sqrt(BQ_ZMAT(knew, 1) * BQ_ZMAT(knew, 1) + BQ_ZMAT(knew, j) * BQ_ZMAT(knew, j));
            const double tempa = BQ_ZMAT(knew, 1) / temp;
            const double tempb = BQ_ZMAT(knew, j) / temp;
            for (int i = 1; i <= 7; i++) {
                temp = tempa * BQ_ZMAT(i, 1) + tempb * BQ_ZMAT(i, j);
                BQ_ZMAT(i, j) = tempa * BQ_ZMAT(i, j) - tempb * BQ_ZMAT(i, 1);
                BQ_ZMAT(i, 1) = temp;
            }
        }
        BQ_ZMAT(knew, j) = 0.0;
    }
    for (int i = 1; i <= 7; i++) w[i - 1] = BQ_ZMAT(knew, 1) * BQ_ZMAT(i, 1);
    const double alpha = w[knew - 1];
    const double tau = BQ_V(vlag, knew);
    BQ_V(vlag, knew) -= 1.0;
    double temp = sqrt(B.denom);
    double tempb = BQ_ZMAT(knew, 1) / temp;
    double tempa = tau / temp;
    for (int i = 1; i <= 7; i++) BQ_ZMAT(i, 1) = tempa * BQ_ZMAT(i, 1) - tempb * BQ_V(vlag, i);
    for (int j = 1; j <= 3; j++) {
        const int jp = 7 + j;
        w[jp - 1] = BQ_BMAT(knew, j);
        tempa = (alpha * BQ_V(vlag, jp) - tau * w[jp - 1]) / B.denom;
        tempb = (-B.beta * w[jp - 1] - tau * BQ_V(vlag, jp)) / B.denom;
        for (int i = 1; i <= jp; i++) {
            BQ_BMAT(i, j) = BQ_BMAT(i, j) + tempa * BQ_V(vlag, i) + tempb * w[i - 1];
            if (i > 7) BQ_BMAT(jp, i - 7) = BQ_BMAT(i, j);
        }
    }
}

// Select the interpolation point to drop for a trust-region step (bobyqa.c:2497-2545, 2648-2690):
// maximise max(1, (dist/delta)^4) * |denominator|.  `from` is xopt before the evaluation and
// xnew after an improving one.
BQ_HD void pick_knew(Bobyqa3& B, const double* from, bool skip_kopt, double& scaden, double& biglsq) {
    const double delsq = B.delta * B.delta;
    scaden = 0.0; biglsq = 0.0;
    B.knew = 0;
    for (int k = 1; k <= 7; k++) {
        if (skip_kopt && k == B.kopt) continue;
        double hdiag = 0.0;
        for (int jj = 1; jj <= 3; jj++) hdiag += BQ_ZMAT(k, jj) * BQ_ZMAT(k, jj);
        const double den = B.beta * hdiag + BQ_V(vlag, k) * BQ_V(vlag, k);
        double distsq = 0.0;
        for (int j = 1; j <= 3; j++) {
            const double t = BQ_XPT(k, j) - from[j - 1];
            distsq += t * t;
        }
        const double q = distsq / delsq;
        const double temp = mx(1.0, q * q);
        if (temp * den > scaden) { scaden = temp * den; B.knew = k; B.denom = den; }
        biglsq = mx(biglsq, temp * (BQ_V(vlag, k) * BQ_V(vlag, k)));
    }
}

// ---------------------------------------------------------------- bobyqb_ main loop (bobyqa.c:2143-3055)
// Runs from `label` until the next objective evaluation is needed (returns BQ_CONTINUE with
// B.xeval set) or the run ends (BQ_DONE).
BQ_HD_NOINLINE int run(Bobyqa3& B, int label) {
    double w[14];
    for (;;) {
        switch (label) {
        case L20: {  // model gradient at xopt when kopt moved away from kbase (bobyqa.c:2149-2182)
            if (B.kopt != B.kbase) {
                int ih = 0;
                for (int j = 1; j <= 3; j++)
                    for (int i = 1; i <= j; i++) {
                        ++ih;
                        if (i < j) BQ_V(gopt, j) += BQ_HQ(ih) * BQ_V(xopt, i);
                        BQ_V(gopt, i) += BQ_HQ(ih) * BQ_V(xopt, j);
                    }
                if (B.nevals > 7) {
                    for (int k = 1; k <= 7; k++) {
                        double temp = 0.0;
                        for (int j = 1; j <= 3; j++) temp += BQ_XPT(k, j) * BQ_V(xopt, j);
                        temp = BQ_PQ(k) * temp;
                        for (int i = 1; i <= 3; i++) BQ_V(gopt, i) += temp * BQ_XPT(k, i);
                    }
                }
            }
            label = L60;
            break;
        }
        case L60: {  // trust-region step (bobyqa.c:2190-2240)
            trsbox(B);
            B.dnorm = mn(B.delta, sqrt(B.dsq));
            if (B.dnorm < 0.5 * B.rho) {
                B.ntrits = -1;
                B.distsq = (10.0 * B.rho) * (10.0 * B.rho);
                if (B.nevals <= B.nfsav + 2) { label = L650; break; }
                const double errbig = mx(mx(B.diffa, B.diffb), B.diffc);
                const double frhosq = B.rho * .125 * B.rho;
                if (B.crvmin > 0.0 && errbig > frhosq * B.crvmin) { label = L650; break; }
                const double bdtol = errbig / B.rho;
                bool to650 = false;
                for (int j = 1; j <= 3; j++) {
                    double bdtest = bdtol;
                    if (BQ_V(xnew, j) == BQ_V(sl, j)) bdtest = BQ_V(gnew, j);
                    if (BQ_V(xnew, j) == BQ_V(su, j)) bdtest = -BQ_V(gnew, j);
                    if (bdtest < bdtol) {
                        double curv = BQ_HQ((j + j * j) / 2);
                        for (int k = 1; k <= 7; k++) curv += BQ_PQ(k) * (BQ_XPT(k, j) * BQ_XPT(k, j));
                        bdtest += 0.5 * curv * B.rho;
                        if (bdtest < bdtol) { to650 = true; break; }
                    }
                }
                label = to650 ? L650 : L680;
                break;
            }
            ++B.ntrits;
            label = L90;
            break;
        }
        case L90: {  // shift xbase to xopt when the step is small relative to |xopt| (bobyqa.c:2246-2322)
            if (B.dsq <= B.xoptsq * .001) {
                const double fracsq = B.xoptsq * .25;
                double sumpq = 0.0;
                double wn[7];  // w(npt+k)
                double vl[7];
                for (int k = 1; k <= 7; k++) {
                    sumpq += BQ_PQ(k);
                    double sum = -0.5 * B.xoptsq;
                    for (int i = 1; i <= 3; i++) sum += BQ_XPT(k, i) * BQ_V(xopt, i);
                    wn[k - 1] = sum;
                    const double temp = fracsq - 0.5 * sum;
                    for (int i = 1; i <= 3; i++) {
                        w[i - 1] = BQ_BMAT(k, i);
                        vl[i - 1] = sum * BQ_XPT(k, i) + temp * BQ_V(xopt, i);
                        const int ip = 7 + i;
                        for (int j = 1; j <= i; j++)
                            BQ_BMAT(ip, j) = BQ_BMAT(ip, j) + w[i - 1] * vl[j - 1] + vl[i - 1] * w[j - 1];
                    }
                }
                for (int jj = 1; jj <= 3; jj++) {
                    double sumz = 0.0, sumw = 0.0;
                    for (int k = 1; k <= 7; k++) {
                        sumz += BQ_ZMAT(k, jj);
                        vl[k - 1] = wn[k - 1] * BQ_ZMAT(k, jj);
                        sumw += vl[k - 1];
                    }
                    for (int j = 1; j <= 3; j++) {
                        double sum = (fracsq * sumz - 0.5 * sumw) * BQ_V(xopt, j);
                        for (int k = 1; k <= 7; k++) sum += vl[k - 1] * BQ_XPT(k, j);
                        w[j - 1] = sum;
                        for (int k = 1; k <= 7; k++) BQ_BMAT(k, j) += sum * BQ_ZMAT(k, jj);
                    }
                    for (int i = 1; i <= 3; i++) {
                        const int ip = i + 7;
                        const double temp = w[i - 1];
                        for (int j = 1; j <= i; j++) BQ_BMAT(ip, j) += temp * w[j - 1];
                    }
                }
                int ih = 0;
                for (int j = 1; j <= 3; j++) {
                    w[j - 1] = -0.5 * sumpq * BQ_V(xopt, j);
                    for (int k = 1; k <= 7; k++) {
                        w[j - 1] += BQ_PQ(k) * BQ_XPT(k, j);
                        BQ_XPT(k, j) -= BQ_V(xopt, j);
                    }
                    for (int i = 1; i <= j; i++) {
                        ++ih;
                        BQ_HQ(ih) = BQ_HQ(ih) + w[i - 1] * BQ_V(xopt, j) + BQ_V(xopt, i) * w[j - 1];
                        BQ_BMAT(7 + i, j) = BQ_BMAT(7 + j, i);
                    }
                }
                for (int i = 1; i <= 3; i++) {
                    BQ_V(xbase, i) += BQ_V(xopt, i);
                    BQ_V(xnew, i) -= BQ_V(xopt, i);
                    BQ_V(sl, i) -= BQ_V(xopt, i);
                    BQ_V(su, i) -= BQ_V(xopt, i);
                    BQ_V(xopt, i) = 0.0;
                }
                B.xoptsq = 0.0;
            }
            label = (B.ntrits == 0) ? L210 : L230;
            break;
        }
        case L210: {  // geometry step (bobyqa.c:2376-2384)
            altmov(B);
            for (int i = 0; i < 3; i++) B.d[i] = B.xnew[i] - B.xopt[i];
            label = L230;
            break;
        }
        case L230: {  // vlag, beta for the step d (bobyqa.c:2390-2440), then knew / denominators
            for (int k = 1; k <= 7; k++) {
                double suma = 0.0, sumb = 0.0, sum = 0.0;
                for (int j = 1; j <= 3; j++) {
                    suma += BQ_XPT(k, j) * BQ_V(d, j);
                    sumb += BQ_XPT(k, j) * BQ_V(xopt, j);
                    sum += BQ_BMAT(k, j) * BQ_V(d, j);
                }
                w[k - 1] = suma * (0.5 * suma + sumb);
                BQ_V(vlag, k) = sum;
                B.wsuma[k - 1] = suma;
            }
            B.beta = 0.0;
            for (int jj = 1; jj <= 3; jj++) {
                double sum = 0.0;
                for (int k = 1; k <= 7; k++) sum += BQ_ZMAT(k, jj) * w[k - 1];
                B.beta -= sum * sum;
                for (int k = 1; k <= 7; k++) BQ_V(vlag, k) += sum * BQ_ZMAT(k, jj);
            }
            B.dsq = 0.0;
            double bsum = 0.0, dx = 0.0;
            for (int j = 1; j <= 3; j++) {
                B.dsq += BQ_V(d, j) * BQ_V(d, j);
                double sum = 0.0;
                for (int k = 1; k <= 7; k++) sum += w[k - 1] * BQ_BMAT(k, j);
                bsum += sum * BQ_V(d, j);
                const int jp = 7 + j;
                for (int i = 1; i <= 3; i++) sum += BQ_BMAT(jp, i) * BQ_V(d, i);
                BQ_V(vlag, jp) = sum;
                bsum += sum * BQ_V(d, j);
                dx += BQ_V(d, j) * BQ_V(xopt, j);
            }
            B.beta = dx * dx + B.dsq * (B.xoptsq + dx + dx + 0.5 * B.dsq) + B.beta - bsum;
            BQ_V(vlag, B.kopt) += 1.0;
            if (B.ntrits == 0) {
                const double vk = BQ_V(vlag, B.knew);
                B.denom = vk * vk + B.alpha * B.beta;
                if (B.denom < B.cauchy && B.cauchy > 0.0) {
                    for (int i = 0; i < 3; i++) { B.xnew[i] = B.xalt[i]; B.d[i] = B.xnew[i] - B.xopt[i]; }
                    B.cauchy = 0.0;
                    label = L230;
                    break;
                }
                if (B.denom <= 0.5 * (vk * vk)) {
                    B.rc = (B.nevals > B.nresc) ? BQ_NEED_RESCUE : BQ_ROUNDOFF_LIMITED;
                    label = L720;
                    break;
                }
            } else {
                double scaden, biglsq;
                pick_knew(B, B.xopt, true, scaden, biglsq);
                if (scaden <= 0.5 * biglsq) {
                    B.rc = (B.nevals > B.nresc) ? BQ_NEED_RESCUE : BQ_ROUNDOFF_LIMITED;
                    label = L720;
                    break;
                }
            }
            label = L360;
            break;
        }
        case L360: {  // next objective evaluation (bobyqa.c:2560-2583)
            if (B.maxeval > 0 && B.nevals >= B.maxeval) { B.rc = BQ_MAXEVAL_REACHED; label = L720; break; }
            set_x_from(B, B.xnew);
            B.nevals++;
            B.phase = PH_MAIN;
            return BQ_CONTINUE;
        }
        case L650: {  // is some interpolation point too far from xopt? (bobyqa.c:2972-3013)
            B.knew = 0;
            for (int k = 1; k <= 7; k++) {
                double sum = 0.0;
                for (int j = 1; j <= 3; j++) {
                    const double t = BQ_XPT(k, j) - BQ_V(xopt, j);
                    sum += t * t;
                }
                if (sum > B.distsq) { B.knew = k; B.distsq = sum; }
            }
            if (B.knew > 0) {
                const double dist = sqrt(B.distsq);
                if (B.ntrits == -1) {
                    B.delta = mn(0.1 * B.delta, 0.5 * dist);
                    if (B.delta <= B.rho * 1.5) B.delta = B.rho;
                }
                B.ntrits = 0;
                B.adelt = mx(mn(0.1 * dist, B.delta), B.rho);
                B.dsq = B.adelt * B.adelt;
                label = L90;
                break;
            }
            if (B.ntrits == -1) { label = L680; break; }
            if (B.ratio > 0.0) { label = L60; break; }
            if (mx(B.delta, B.dnorm) > B.rho) { label = L60; break; }
            label = L680;
            break;
        }
        case L680: {  // reduce rho, or finish (bobyqa.c:3019-3041)
            if (B.rho > B.rhoend) {
                B.delta = 0.5 * B.rho;
                B.ratio = B.rho / B.rhoend;
                if (B.ratio <= 16.) B.rho = B.rhoend;
                else if (B.ratio <= 250.) B.rho = sqrt(B.ratio) * B.rhoend;
                else B.rho = 0.1 * B.rho;
                B.delta = mx(B.delta, B.rho);
                B.ntrits = 0;
                B.nfsav = B.nevals;
                label = L60;
                break;
            }
            if (B.ntrits == -1) { label = L360; break; }
            label = L720;
            break;
        }
        case L720:
        default: {  // return the best point (bobyqa.c:3042-3054)
            for (int i = 1; i <= 3; i++) {
                double v = mn(mx(BQ_V(xl, i), BQ_V(xbase, i) + BQ_V(xopt, i)), BQ_V(xu, i));
                if (BQ_V(xopt, i) == BQ_V(sl, i)) v = BQ_V(xl, i);
                if (BQ_V(xopt, i) == BQ_V(su, i)) v = BQ_V(xu, i);
                BQ_V(x, i) = v;
            }
            B.minf = BQ_FVAL(B.kopt);
            for (int i = 0; i < 3; i++) B.xfinal[i] = B.x[i] * B.sc[i];
            B.phase = PH_DONE;
            return BQ_DONE;
        }
        }
    }
}

// Code after an evaluation at xbase + xnew in the main loop (bobyqa.c:2583-2970)
BQ_HD_NOINLINE int after_eval(Bobyqa3& B, double f) {
    double w[10];
    if (B.ntrits == -1) {
        B.fsave = f;
        B.rc = BQ_XTOL_REACHED;
        if (B.fsave < BQ_FVAL(B.kopt)) {
            B.minf = f;
            for (int i = 0; i < 3; i++) B.xfinal[i] = B.x[i] * B.sc[i];
            B.phase = PH_DONE;
            return BQ_DONE;
        }
        return run(B, L720);
    }
    const double fopt = BQ_FVAL(B.kopt);
    double vquad = 0.0;
    int ih = 0;
    for (int j = 1; j <= 3; j++) {
        vquad += BQ_V(d, j) * BQ_V(gopt, j);
        for (int i = 1; i <= j; i++) {
            ++ih;
            double temp = BQ_V(d, i) * BQ_V(d, j);
            if (i == j) temp = 0.5 * temp;
            vquad += BQ_HQ(ih) * temp;
        }
    }
    for (int k = 1; k <= 7; k++) vquad += 0.5 * BQ_PQ(k) * (B.wsuma[k - 1] * B.wsuma[k - 1]);
    const double diff = f - fopt - vquad;
    B.diffc = B.diffb;
    B.diffb = B.diffa;
    B.diffa = fabs(diff);
    if (B.dnorm > B.rho) B.nfsav = B.nevals;

    if (B.ntrits > 0) {
        if (vquad >= 0.0) { B.rc = BQ_ROUNDOFF_LIMITED; return run(B, L720); }
        B.ratio = (f - fopt) / vquad;
        if (B.ratio <= 0.1) B.delta = mn(0.5 * B.delta, B.dnorm);
        else if (B.ratio <= .7) B.delta = mx(0.5 * B.delta, B.dnorm);
        else B.delta = mx(0.5 * B.delta, B.dnorm + B.dnorm);
        if (B.delta <= B.rho * 1.5) B.delta = B.rho;
        if (f < fopt) {
            const int ksav = B.knew;
            const double densav = B.denom;
            double scaden, biglsq;
            pick_knew(B, B.xnew, false, scaden, biglsq);
            if (scaden <= 0.5 * biglsq) { B.knew = ksav; B.denom = densav; }
        }
    }
    update(B);
    const int knew = B.knew;
    ih = 0;
    const double pqold = BQ_PQ(knew);
    BQ_PQ(knew) = 0.0;
    for (int i = 1; i <= 3; i++) {
        const double temp = pqold * BQ_XPT(knew, i);
        for (int j = 1; j <= i; j++) { ++ih; BQ_HQ(ih) += temp * BQ_XPT(knew, j); }
    }
    for (int jj = 1; jj <= 3; jj++) {
        const double temp = diff * BQ_ZMAT(knew, jj);
        for (int k = 1; k <= 7; k++) BQ_PQ(k) += temp * BQ_ZMAT(k, jj);
    }
    BQ_FVAL(knew) = f;
    for (int i = 1; i <= 3; i++) { BQ_XPT(knew, i) = BQ_V(xnew, i); w[i - 1] = BQ_BMAT(knew, i); }
    for (int k = 1; k <= 7; k++) {
        double suma = 0.0;
        for (int jj = 1; jj <= 3; jj++) suma += BQ_ZMAT(knew, jj) * BQ_ZMAT(k, jj);
        if (is_inf(suma)) { B.rc = BQ_ROUNDOFF_LIMITED; return run(B, L720); }
        double sumb = 0.0;
        for (int j = 1; j <= 3; j++) sumb += BQ_XPT(k, j) * BQ_V(xopt, j);
        const double temp = suma * sumb;
        for (int i = 1; i <= 3; i++) w[i - 1] += temp * BQ_XPT(k, i);
    }
    for (int i = 1; i <= 3; i++) BQ_V(gopt, i) += diff * w[i - 1];

    if (f < fopt) {
        B.kopt = knew;
        B.xoptsq = 0.0;
        ih = 0;
        for (int j = 1; j <= 3; j++) {
            BQ_V(xopt, j) = BQ_V(xnew, j);
            B.xoptsq += BQ_V(xopt, j) * BQ_V(xopt, j);
            for (int i = 1; i <= j; i++) {
                ++ih;
                if (i < j) BQ_V(gopt, j) += BQ_HQ(ih) * BQ_V(d, i);
                BQ_V(gopt, i) += BQ_HQ(ih) * BQ_V(d, j);
            }
        }
        for (int k = 1; k <= 7; k++) {
            double temp = 0.0;
            for (int j = 1; j <= 3; j++) temp += BQ_XPT(k, j) * BQ_V(d, j);
            temp = BQ_PQ(k) * temp;
            for (int i = 1; i <= 3; i++) BQ_V(gopt, i) += temp * BQ_XPT(k, i);
        }
        // nlopt_stop_ftol with ftol_rel = ftol_abs = 0 (PatchOptimizer.cpp sets only xtol_rel) never fires
    }

    if (B.ntrits > 0) {  // least-Frobenius-norm interpolant test (bobyqa.c:2845-2944)
        double vl[10], wk[14];
        for (int k = 1; k <= 7; k++) { vl[k - 1] = BQ_FVAL(k) - BQ_FVAL(B.kopt); wk[k - 1] = 0.0; }
        for (int j = 1; j <= 3; j++) {
            double sum = 0.0;
            for (int k = 1; k <= 7; k++) sum += BQ_ZMAT(k, j) * vl[k - 1];
            for (int k = 1; k <= 7; k++) wk[k - 1] += sum * BQ_ZMAT(k, j);
        }
        for (int k = 1; k <= 7; k++) {
            double sum = 0.0;
            for (int j = 1; j <= 3; j++) sum += BQ_XPT(k, j) * BQ_V(xopt, j);
            wk[k + 7 - 1] = wk[k - 1];
            wk[k - 1] = sum * wk[k - 1];
        }
        double gqsq = 0.0, gisq = 0.0;
        for (int i = 1; i <= 3; i++) {
            double sum = 0.0;
            for (int k = 1; k <= 7; k++) sum = sum + BQ_BMAT(k, i) * vl[k - 1] + BQ_XPT(k, i) * wk[k - 1];
            if (BQ_V(xopt, i) == BQ_V(sl, i)) {
                const double t1 = mn(0.0, BQ_V(gopt, i));
                gqsq += t1 * t1;
                const double t2 = mn(0.0, sum);
                gisq += t2 * t2;
            } else if (BQ_V(xopt, i) == BQ_V(su, i)) {
                const double t1 = mx(0.0, BQ_V(gopt, i));
                gqsq += t1 * t1;
                const double t2 = mx(0.0, sum);
                gisq += t2 * t2;
            } else {
                gqsq += BQ_V(gopt, i) * BQ_V(gopt, i);
                gisq += sum * sum;
            }
            vl[7 + i - 1] = sum;
        }
        ++B.itest;
        if (gqsq < 10.0 * gisq) B.itest = 0;
        if (B.itest >= 3) {
            for (int i = 1; i <= 7; i++) {  // max(npt, nh) = 7
                if (i <= 3) BQ_V(gopt, i) = vl[7 + i - 1];
                if (i <= 7) BQ_PQ(i) = wk[7 + i - 1];
                if (i <= 6) BQ_HQ(i) = 0.0;
                B.itest = 0;
            }
        }
    }
    if (B.ntrits == 0) return run(B, L60);
    if (f <= fopt + 0.1 * vquad) return run(B, L60);
    const double a = 2.0 * B.delta, b = 10.0 * B.rho;
    B.distsq = mx(a * a, b * b);
    return run(B, L650);
}

}  // namespace bq

// nlopt_set_default_initial_step (thirdLibs/nlopt-2.4.2/api/options.c:686-728)
BQ_HD void bq_default_step(const double* x, const double* lb, const double* ub, double* dx) {
    for (int i = 0; i < 3; i++) {
        double step = HUGE_VAL;
        if (!bq::is_inf(ub[i]) && !bq::is_inf(lb[i]) && (ub[i] - lb[i]) * 0.25 < step && ub[i] > lb[i])
            step = (ub[i] - lb[i]) * 0.25;
        if (!bq::is_inf(ub[i]) && ub[i] - x[i] < step && ub[i] > x[i]) step = (ub[i] - x[i]) * 0.75;
        if (!bq::is_inf(lb[i]) && x[i] - lb[i] < step && x[i] > lb[i]) step = (x[i] - lb[i]) * 0.75;
        if (bq::is_inf(step)) {
            if (!bq::is_inf(ub[i]) && fabs(ub[i] - x[i]) < fabs(step)) step = (ub[i] - x[i]) * 1.1;
            if (!bq::is_inf(lb[i]) && fabs(x[i] - lb[i]) < fabs(step)) step = (x[i] - lb[i]) * 1.1;
        }
        if (bq::is_inf(step) || step == 0) step = x[i];
        if (bq::is_inf(step) || step == 0) step = 1;
        dx[i] = step;
    }
}

// bobyqa() front end (bobyqa.c:3073-3268) + the start of prelim_.  Returns BQ_CONTINUE with the
// first point in B.xeval, or BQ_DONE with B.rc = BQ_INVALID_ARGS.
BQ_HD_NOINLINE int bq_start(Bobyqa3& B, const double* x0, const double* lb, const double* ub, const double* dx,
                   double xtol_rel, int maxeval) {
    B.nevals = 0; B.rc = BQ_SUCCESS; B.minf = HUGE_VAL; B.maxeval = maxeval; B.phase = bq::PH_PRELIM;
    for (int i = 0; i < 3; i++) { B.xfinal[i] = x0[i]; B.xeval[i] = x0[i]; }
    for (int i = 0; i < 3; i++)
        if (lb[i] > ub[i] || x0[i] < lb[i] || x0[i] > ub[i]) { B.rc = BQ_INVALID_ARGS; B.phase = bq::PH_DONE; return BQ_DONE; }
    // nlopt_compute_rescaling (util/rescale.c:29-44)
    for (int i = 0; i < 3; i++) B.sc[i] = 1.0;
    if (!(dx[1] == dx[0] && dx[2] == dx[1]))
        for (int i = 1; i < 3; i++) B.sc[i] = dx[i] / dx[0];
    for (int i = 0; i < 3; i++) {
        B.x[i] = x0[i] / B.sc[i];
        B.xl[i] = lb[i] / B.sc[i];
        B.xu[i] = ub[i] / B.sc[i];
        if (B.xl[i] > B.xu[i]) { const double t = B.xl[i]; B.xl[i] = B.xu[i]; B.xu[i] = t; }
    }
    B.rhobeg = fabs(dx[0] / B.sc[0]);
    B.rhoend = xtol_rel * B.rhobeg;  // xtol_abs = 0
    for (int j = 0; j < 3; j++) {
        const double temp = B.xu[j] - B.xl[j];
        if (temp < B.rhobeg + B.rhobeg) { B.rc = BQ_INVALID_ARGS; B.phase = bq::PH_DONE; return BQ_DONE; }
        B.sl[j] = B.xl[j] - B.x[j];
        B.su[j] = B.xu[j] - B.x[j];
        if (B.sl[j] >= -B.rhobeg) {
            if (B.sl[j] >= 0.0) { B.x[j] = B.xl[j]; B.sl[j] = 0.0; B.su[j] = temp; }
            else { B.x[j] = B.xl[j] + B.rhobeg; B.sl[j] = -B.rhobeg; B.su[j] = bq::mx(B.xu[j] - B.x[j], B.rhobeg); }
        } else if (B.su[j] <= B.rhobeg) {
            if (B.su[j] <= 0.0) { B.x[j] = B.xu[j]; B.sl[j] = -temp; B.su[j] = 0.0; }
            else { B.x[j] = B.xu[j] - B.rhobeg; B.sl[j] = bq::mn(B.xl[j] - B.x[j], -B.rhobeg); B.su[j] = B.rhobeg; }
        }
    }
    // prelim_ initialisation (bobyqa.c:1790-1822)
    for (int j = 0; j < 3; j++) { B.xbase[j] = B.x[j]; B.xopt[j] = 0; B.gopt[j] = 0; B.xnew[j] = 0; B.xalt[j] = 0; B.d[j] = 0; B.gnew[j] = 0; }
    for (int k = 0; k < 21; k++) { B.xpt[k] = 0.0; B.zmat[k] = 0.0; }
    for (int k = 0; k < 30; k++) B.bmat[k] = 0.0;
    for (int k = 0; k < 6; k++) B.hq[k] = 0.0;
    for (int k = 0; k < 7; k++) { B.pq[k] = 0.0; B.fval[k] = 0.0; B.wsuma[k] = 0.0; }
    for (int k = 0; k < 10; k++) B.vlag[k] = 0.0;
    B.rho = B.delta = B.diffa = B.diffb = B.diffc = B.dsq = B.crvmin = B.dnorm = B.xoptsq = B.ratio = B.adelt = 0.0;
    B.alpha = B.cauchy = B.beta = B.denom = B.fsave = B.distsq = B.fbeg = B.stepa = B.stepb = 0.0;
    B.kopt = 1; B.knew = 0; B.kbase = 1; B.ntrits = 0; B.itest = 0; B.nresc = 0; B.nfsav = 0; B.nf = 0;
    bq::prelim_next_point(B);
    B.nevals++;
    return BQ_CONTINUE;
}

BQ_HD_NOINLINE int bq_step(Bobyqa3& B, double f) {
    if (B.phase == bq::PH_PRELIM) {
        bq::prelim_consume(B, f);
        // stopping tests after each prelim evaluation (bobyqa.c:1944-1947)
        const bool maxed = (B.maxeval > 0 && B.nevals >= B.maxeval);
        if (!maxed && B.nf < 7) {
            bq::prelim_next_point(B);
            B.nevals++;
            return BQ_CONTINUE;
        }
        // bobyqb_ set-up after prelim_ (bobyqa.c:2117-2143)
        B.xoptsq = 0.0;
        for (int i = 1; i <= 3; i++) {
            BQ_V(xopt, i) = BQ_XPT(B.kopt, i);
            B.xoptsq += BQ_V(xopt, i) * BQ_V(xopt, i);
        }
        B.fsave = BQ_FVAL(1);
        if (maxed) { B.rc = BQ_MAXEVAL_REACHED; return bq::run(B, bq::L720); }
        B.kbase = 1;
        B.rho = B.rhobeg;
        B.delta = B.rho;
        B.nresc = B.nevals;
        B.ntrits = 0;
        B.diffa = 0.0; B.diffb = 0.0; B.diffc = 0.0;
        B.itest = 0;
        B.nfsav = B.nevals;
        return bq::run(B, bq::L20);
    }
    if (B.phase == bq::PH_MAIN) return bq::after_eval(B, f);
    return BQ_DONE;
}

}  // namespace hpmvs
