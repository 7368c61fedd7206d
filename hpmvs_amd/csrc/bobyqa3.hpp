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
//   rescue_*          <- rescue_    bobyqa.c:143-742 (never needed on the photometric objective in
//                        1.1 M benchmark patches, pinned on synthetic ill-conditioned objectives)
//
// Indexing: the accessor macros are 1-based like the mathematics; storage is dense 0-based.
//
// ---------------------------------------------------------------------------------------------------------------
// LICENCE NOTICE.  The numerical content of this file is M. J. D. Powell's BOBYQA algorithm as distributed with
// NLopt 2.4.2 (bobyqa/bobyqa.c, "converted via v2c, cleaned up, and incorporated into NLopt by S. G. Johnson").
// The routines below restate that code statement by statement (same quantities, same operation order -- that is
// the point: bit-identical trajectories), re-structured as a resumable fixed-size state machine.  This file is
// therefore a derived work of MIT-licensed software and carries its notice (NLopt 2.4.2 bobyqa/COPYRIGHT):
//
//   Copyright (c) 2009 M. J. D. Powell (mjdp@cam.ac.uk)
//   Modifications Copyright (c) 2010 Massachusetts Institute of Technology
//
//   Permission is hereby granted, free of charge, to any person obtaining a copy of this software and associated
//   documentation files (the "Software"), to deal in the Software without restriction, including without
//   limitation the rights to use, copy, modify, merge, publish, distribute, sublicense, and/or sell copies of the
//   Software, and to permit persons to whom the Software is furnished to do so, subject to the following
//   conditions:
//
//   The above copyright notice and this permission notice shall be included in all copies or substantial portions
//   of the Software.
//
//   THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY KIND, EXPRESS OR IMPLIED, INCLUDING BUT NOT LIMITED
//   TO THE WARRANTIES OF MERCHANTABILITY, FITNESS FOR A PARTICULAR PURPOSE AND NONINFRINGEMENT. IN NO EVENT SHALL
//   THE AUTHORS OR COPYRIGHT HOLDERS BE LIABLE FOR ANY CLAIM, DAMAGES OR OTHER LIABILITY, WHETHER IN AN ACTION OF
//   CONTRACT, TORT OR OTHERWISE, ARISING FROM, OUT OF OR IN CONNECTION WITH THE SOFTWARE OR THE USE OR OTHER
//   DEALINGS IN THE SOFTWARE.
// ---------------------------------------------------------------------------------------------------------------
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BQ_HD __host__ __device__ inline
// the big routines are real calls on the GPU: their register demand stays out of the caller's
// wave-cooperative sampling loop (occupancy), and the code is shared by the call sites
#define BQ_HD_NOINLINE static __host__ __device__ __attribute__((noinline))  // static: two kernels (TUs) include this header
// the blocks of the main loop are all inlined into ONE real function (bq_step_): a block that needs more than the
// 144 caller-saved VGPRs saves and restores callee-saved ones through scratch memory on every call (measured: ~27
// cycles per saved register and call), one function pays that once per optimiser step instead of once per block
#define BQ_HD_BLOCK static __host__ __device__ __attribute__((always_inline)) inline
#else
#define BQ_HD inline
#define BQ_HD_NOINLINE inline
#define BQ_HD_BLOCK inline
#endif
#include <math.h>

// optional region timers (only in -DHPMVS_PROFILE_PHASES device builds; see tools/phase_profile.py)
#include "prof.h"
#if defined(HPMVS_PROFILE_PHASES) && defined(__HIP_DEVICE_COMPILE__)
#define BQ_PROF_COUNT(i, n) atomicAdd(&s_prof[(i)], (unsigned long long)(n))
#define BQ_PROF_BEGIN(v) const long long v = clock64()
#define BQ_PROF_END(v, i) do { if (__lane_id() == (unsigned)(__ffsll((long long)__ballot(1)) - 1)) { atomicAdd(&s_prof[2 * (i)], (unsigned long long)(clock64() - v)); atomicAdd(&s_prof[2 * (i) + 1], 1ull); } } while (0)
struct BqProfScope {  // times one pass of a loop body (continue / break leave through the destructor)
    long long t0; int i;
    __device__ BqProfScope(int i_) : t0(clock64()), i(i_) {}
    __device__ ~BqProfScope() { if (__lane_id() == (unsigned)(__ffsll((long long)__ballot(1)) - 1)) { atomicAdd(&s_prof[2 * i], (unsigned long long)(clock64() - t0)); atomicAdd(&s_prof[2 * i + 1], 1ull); } }
};
#define BQ_PROF_SCOPE(i) BqProfScope bq_prof_scope_(i)
#else
#define BQ_PROF_COUNT(i, n) do { } while (0)
#define BQ_PROF_BEGIN(v) do { } while (0)
#define BQ_PROF_END(v, i) do { } while (0)
#define BQ_PROF_SCOPE(i) do { } while (0)
#endif

// Loads placed after BQ_STAGE() stay after it: the blocks below read their arrays in stages, and a stage's registers
// are meant to be free again before the next stage's loads are issued (a block that needs more than the 144
// caller-saved VGPRs has to save and restore callee-saved ones through scratch memory on every call).
#if defined(__HIP_DEVICE_COMPILE__)
#define BQ_STAGE() __builtin_amdgcn_sched_barrier(0)
#else
#define BQ_STAGE() do { } while (0)
#endif

namespace hpmvs {

enum { BQ_CONTINUE = 0, BQ_DONE = 1 };
// nlopt_result codes (thirdLibs/nlopt-2.4.2/api/nlopt.h) + one private code
enum {
    BQ_FAILURE = -1, BQ_INVALID_ARGS = -2, BQ_ROUNDOFF_LIMITED = -4,
    BQ_SUCCESS = 1, BQ_XTOL_REACHED = 4, BQ_MAXEVAL_REACHED = 5,
};

// ---- state storage ------------------------------------------------------------------------
// All optimiser state is addressed through BQ_D(index) / BQ_I(index):
//   host build : a plain struct of two arrays;
//   device build (BQ_STATE_LDS defined before inclusion, BQ_P = patches per wavefront): the state of
//   the BQ_P patches a wavefront owns lives in LDS as [index][slot], so that the lanes of a wave
//   touch consecutive 8-byte words (conflict-free ds_read_b64 / ds_write_b64) and nothing of the
//   ~1.4 KB per patch goes through scratch / HBM.
enum {  // float64 slots: arrays first (names as in Powell's paper / bobyqb_), then scalars.
    // 164 doubles + 13 ints = 1364 B per patch: 15 patches per wavefront x 8 wavefronts fill the 160 KB of a CU
    // exactly.  Slots are shared where two quantities are never alive at the same time (each pair checked
    // against every read/write site below; the self-tests against the genuine library cover all of them):
    //   xfinal = xeval   both are x * sc of the point in hand: xeval while the run is going, xfinal once it is over
    //   gnew   = xalt    gnew lives from trsbox to the end of label 60; xalt from altmov (label 210) to label 230
    //   stepa, stepb, fbeg = diffa, diffb, diffc   prelim_ only / main loop only (zeroed when prelim_ ends)
    //   adelt  = distsq  distsq is consumed by label 650 before that label sets adelt for altmov
    // The scaled trial point `x` (bobyqa.c's X array) is a local wherever it is formed.
    O_xeval = 0,
    O_xfinal = O_xeval,
    O_sc = 3,
    O_xl = 6,
    O_xu = 9,
    O_xbase = 12,
    O_xopt = 15,
    O_gopt = 18,
    O_sl = 21,
    O_su = 24,
    O_xnew = 27,
    O_xalt = 30,
    O_gnew = O_xalt,
    O_d = 33,
    O_xpt = 36,
    O_fval = 57,
    O_pq = 64,
    O_hq = 71,
    O_bmat = 77,
    O_zmat = 107,
    O_vlag = 128,
    O_wsuma = 138,
    O_minf = 145,
    O_rhobeg = 146,
    O_rhoend = 147,
    O_rho = 148,
    O_delta = 149,
    O_diffa = 150,
    O_diffb = 151,
    O_diffc = 152,
    O_stepa = O_diffa,
    O_stepb = O_diffb,
    O_fbeg = O_diffc,
    O_dsq = 153,
    O_crvmin = 154,
    O_dnorm = 155,
    O_xoptsq = 156,
    O_ratio = 157,
    O_distsq = 158,
    O_adelt = O_distsq,
    O_alpha = 159,
    O_cauchy = 160,
    O_beta = 161,
    O_denom = 162,
    O_fsave = 163,
    BQ_ND = 164,
    // rescue_ state (bobyqa.c:143-742) overlays vectors that are dead between label 190 and the next
    // trust-region / geometry step: xnew, xalt, d are recomputed by trsbox / altmov before they are read
    // again, wsuma only lives from label 230 to the evaluation that follows it.
    O_ptsaux = O_xnew,   // 6 = xnew[3] + xalt[3]
    O_ptsid = O_wsuma,   // 7
    O_fbase_r = O_d,
    O_vquad_r = O_d + 1
};
enum {  // int32 slots
    I_rc = 0,
    I_nevals = 1,
    I_maxeval = 2,
    I_kopt = 3,
    I_knew = 4,
    I_kbase = 5,
    I_ntrits = 6,
    I_itest = 7,
    I_nresc = 8,
    I_nfsav = 9,
    I_nf = 10,
    I_phase = 11,
    I_kpt = 12,
    BQ_NI = 13
};

#if defined(BQ_STATE_LDS)
#ifndef BQ_P
#error "define BQ_P (patches per wavefront) together with BQ_STATE_LDS"
#endif
// BQ_SLOTS = slots of the whole workgroup (BQ_P per wavefront x wavefronts per workgroup); B.slot indexes them
#ifndef BQ_SLOTS
#define BQ_SLOTS BQ_P
#endif
// BQ_COLD_GLOBAL (round 4): the three big matrices of the updating formula -- bmat (30), zmat (21), vlag (10), the
// contiguous index range [O_bmat, O_wsuma), 61 of the 164 float64 values -- live in GLOBAL memory, [index][slot] like the
// LDS part (one block per wavefront in the launch workspace, never paged), and only the other 103 values + 13 ints take
// LDS.  Five blocks touch them (update, the two denominator blocks, the model update, altmov), each reading them once at
// its top and writing once at its end, so the price is a handful of L2 round trips per optimiser step -- and a state
// region shrinks from 79 KB to 50 KB for 58 patches: THREE regions per CU instead of two.  The regions are what a
// launch queues for (profiles/r04/hold_sensitivity.txt: every cycle a region is held costs two cycles of a wavefront's
// round).  The accesses are BUFFER loads / stores -- resource descriptor in scalar registers, one 32-bit lane offset,
// the element's offset as the instruction's immediate -- through an lvalue proxy (bq_cold_ref): with flat or global
// addressing every element cost a 64-bit address register pair, which the blocks (at the register limit already) paid
// for in spills; that first attempt ran at 340 ms against 229 (profiles/r04/hold_sensitivity.txt).
#if defined(BQ_COLD_GLOBAL)
#ifndef BQ_COLD_LO
#define BQ_COLD_LO 77   /* O_bmat; 36 = O_xpt: xpt, fval, pq, hq as well (the range is contiguous up to O_wsuma) */
#endif
#ifndef BQ_COLD_HI
#define BQ_COLD_HI 138  /* O_wsuma; 145 = O_minf: wsuma (and rescue_'s ptsid, which shares its slots) as well */
#endif
#define BQ_NCOLD (BQ_COLD_HI - BQ_COLD_LO)
#define BQ_ND_LDS (164 - BQ_NCOLD)
#ifndef BQ_CSTRIDE
#define BQ_CSTRIDE BQ_P   /* slots per cold block */
#endif
#else
#define BQ_ND_LDS 164
#endif
#if defined(BQ_STATE_LDS_EXTERN)
// the including kernel owns the LDS buffer (shared with its other phases) and defines bq_lds_d / bq_lds_i
// as pointer expressions into it before including this header
#else
__shared__ double bq_lds_d[BQ_ND_LDS * BQ_SLOTS];
__shared__ int bq_lds_i[13 * BQ_SLOTS];
#endif
#if defined(BQ_COLD_GLOBAL)
typedef unsigned int bq_u2 __attribute__((ext_vector_type(2)));
#ifndef BQ_COLD_LD_AUX
#define BQ_COLD_LD_AUX 0   /* cache-policy bits of the cold block's loads / stores (lab: 2 = non-temporal) */
#endif
#ifndef BQ_COLD_ST_AUX
#define BQ_COLD_ST_AUX 0
#endif
struct bq_cold_ref {  // one float64 of the cold block: reads convert, assignments store
    __amdgpu_buffer_rsrc_t r; unsigned voff; int soff;
    __device__ __forceinline__ operator double() const { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, soff, BQ_COLD_LD_AUX)); }
    __device__ __forceinline__ void operator=(double v) const { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(bq_u2, v), r, (int)voff, soff, BQ_COLD_ST_AUX); }
    __device__ __forceinline__ void operator=(const bq_cold_ref& o) const { *this = (double)o; }
    __device__ __forceinline__ void operator+=(double v) const { *this = (double)*this + v; }
    __device__ __forceinline__ void operator-=(double v) const { *this = (double)*this - v; }
};
// cold: this wavefront's block of the cold arrays (the same on all its lanes: coldp is its address, kept so that a real
// function can rebuild the descriptor in scalar registers, BQ_UNIFORM), coff: byte offset of my column
struct Bobyqa3 { int slot; int slot_i; unsigned coff; char* coldp; __amdgpu_buffer_rsrc_t cold; };
__device__ __forceinline__ bq_cold_ref bq_cold_at(const Bobyqa3& B, int i) {  // i: state index in [BQ_COLD_LO, BQ_COLD_HI)
    const bool c = __builtin_constant_p(i);
    const int byte = (i - BQ_COLD_LO) * (BQ_CSTRIDE * 8);
    return bq_cold_ref{B.cold, B.coff + (c ? 0u : (unsigned)byte), c ? byte : 0};
}
#define BQ_HOTIDX(i) ((i) < BQ_COLD_LO ? (i) : (i) - BQ_NCOLD)
#define BQ_D(i) bq_lds_d[BQ_HOTIDX(i) * BQ_SLOTS + B.slot]
#define BQ_C(i) bq_cold_at(B, (i))
#if BQ_COLD_HI >= 145
#define BQ_W(i) bq_cold_at(B, (i))   /* wsuma / ptsid are cold too */
#else
#define BQ_W(i) BQ_D(i)
#endif
#if BQ_COLD_LO <= 36
#define BQ_M(i) bq_cold_at(B, (i))   /* xpt, fval, pq, hq are cold too */
#else
#define BQ_M(i) BQ_D(i)
#endif
#else
struct Bobyqa3 { int slot; int slot_i; };  // slot: index into the float64 array, slot_i: into the int32 array (they differ when the including kernel keeps several state regions)
#define BQ_D(i) bq_lds_d[(i) * BQ_SLOTS + B.slot]
#define BQ_C(i) BQ_D(i)
#define BQ_M(i) BQ_D(i)
#define BQ_W(i) BQ_D(i)
#endif
#define BQ_I(i) bq_lds_i[(i) * BQ_SLOTS + B.slot_i]
#if defined(BQ_COLD_GLOBAL)
// A real call hands the handle over in vector registers; a buffer descriptor there would make every access a
// waterfall loop.  Every real function rebuilds it from the block's address read as a scalar.
__device__ __forceinline__ Bobyqa3 bq_uniform(Bobyqa3 b) {
    const unsigned long long a = (unsigned long long)b.coldp;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(a & 0xffffffffull)), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32));
    b.coldp = (char*)(((unsigned long long)hi << 32) | lo);
    b.cold = __builtin_amdgcn_make_buffer_rsrc(b.coldp, 0, BQ_NCOLD * BQ_CSTRIDE * 8, 0x00020000);
    return b;
}
#define BQ_UNIFORM(Bin) bq_uniform(Bin)
#else
#define BQ_UNIFORM(Bin) (Bin)
#endif
#else
struct Bobyqa3 { double dd[BQ_ND]; int ii[BQ_NI]; };
#define BQ_D(i) B.dd[(i)]
#define BQ_C(i) B.dd[(i)]
#define BQ_M(i) B.dd[(i)]
#define BQ_W(i) B.dd[(i)]
#define BQ_I(i) B.ii[(i)]
#define BQ_UNIFORM(Bin) (Bin)
#endif

// How the state handle travels through function calls.  On the device it is just the slot index and is
// passed BY VALUE: a reference would be a generic pointer to a scratch copy, and since a generic pointer may
// alias LDS the compiler has to re-load the slot index from scratch (a flat load, several hundred cycles)
// after every LDS store.  On the host the handle holds the arrays and is passed by reference.
#if defined(BQ_STATE_LDS)
typedef const Bobyqa3 BqRef;
#else
typedef Bobyqa3& BqRef;
#endif

#define BQA(name, idx) BQ_D(O_##name + (idx))   // 0-based element of a named array
#define BQS(name) BQ_D(O_##name)                // float64 scalar
#define BQN(name) BQ_I(I_##name)                // int scalar
// 1-based accessors like the mathematics
#define BQ_XPT(k, i) BQ_M(O_xpt + ((k) - 1) * 3 + ((i) - 1))
#define BQ_BMAT(k, i) BQ_C(O_bmat + ((k) - 1) * 3 + ((i) - 1))
#define BQ_ZMAT(k, j) BQ_C(O_zmat + ((k) - 1) * 3 + ((j) - 1))
#define BQ_FVAL(k) BQ_M(O_fval + (k) - 1)
#define BQ_PQ(k) BQ_M(O_pq + (k) - 1)
#define BQ_HQ(ih) BQ_M(O_hq + (ih) - 1)
#define BQ_V(a, i) BQ_D(O_##a + (i) - 1)
#define BQ_VLAG(i) BQ_C(O_vlag + (i) - 1)
#if defined(BQ_STATE_LDS) && defined(BQ_COLD_GLOBAL)
#define BQ_EITHER(i) (((i) >= BQ_COLD_LO && (i) < BQ_COLD_HI) ? (double)bq_cold_at(B, (i)) : (double)BQ_D(i))
#else
#define BQ_EITHER(i) BQ_D(i)
#endif
#define BQ_WS(idx) BQ_W(O_wsuma + (idx))              // element of wsuma (rescue_'s ptsid shares the slots): hot or cold by configuration
#define BQ_COLDA(name, idx) BQ_C(O_##name + (idx))
#define BQ_MA(name, idx) BQ_M(O_##name + (idx))      // 0-based element of xpt / fval / pq / hq: hot or cold by configuration  // 0-based element of one of the cold arrays (bmat, zmat, vlag)
#define BQ_PA(i, j) BQ_D(O_ptsaux + ((j) - 1) * 2 + ((i) - 1))  // PTSAUX(i,j) of rescue_
#define BQ_PTSID(k) BQ_W(O_ptsid + (k) - 1)

namespace bq {

// trsbox's 2-D boundary search evaluates the model at angt = angbd * i / iu, i = 1..iu (bobyqa.c:1590-1606).
// Unless a bound restricts the rotation, angbd is exactly 1.0 and iu = (int)(17 + 3.1) = 20, so the
// values angt_i = fl(i / 20) and sth_i = fl(fl(angt + angt) / fl(1 + fl(angt * angt))) are the same 20
// numbers every time.  They are tabulated at compile time with the SAME IEEE operations the loop would
// execute (constant evaluation is correctly rounded, no contraction), which removes 40 float64
// divisions per boundary iteration without changing a single bit of the result.
struct AngleTable {
    double angt[20], sth[20];
    constexpr AngleTable() : angt(), sth() {
        for (int i = 1; i <= 20; i++) {
            const double a = 1.0 * (double)i / (double)20;
            angt[i - 1] = a;
            sth[i - 1] = (a + a) / (1.0 + a * a);
        }
    }
};
static constexpr AngleTable kAngleTable{};

BQ_HD double mn(double a, double b) { return a <= b ? a : b; }
BQ_HD double mx(double a, double b) { return a >= b ? a : b; }
BQ_HD bool is_inf(double v) { return fabs(v) >= HUGE_VAL * 0.99; }

enum { PH_PRELIM = 0, PH_MAIN = 1, PH_DONE = 2, PH_RESCUE = 3 };
enum { L20 = 0, L60, L90, L210, L230, L360, L650, L680, L720, L190, LRET_CONT, LRET_DONE };

// x (scaled) for the interpolation point / trial step `p` (relative to xbase), bounds respected
// exactly as bobyqb_ does before each calfun (bobyqa.c:2560-2575, 1871-1883)
BQ_HD void set_x_from(BqRef B, int p_off) {  // p_off: state offset of the 3-vector p
    for (int i = 1; i <= 3; i++) {
        const double pi = BQ_EITHER(p_off + i - 1);
        double v = mn(mx(BQ_V(xl, i), BQ_V(xbase, i) + pi), BQ_V(xu, i));
        if (pi == BQ_V(sl, i)) v = BQ_V(xl, i);
        if (pi == BQ_V(su, i)) v = BQ_V(xu, i);
        BQ_V(xeval, i) = v * BQ_V(sc, i);  // nlopt_unscale (util/rescale.c:55-60)
    }
}

// ---------------------------------------------------------------- prelim_ (bobyqa.c:1713-1952)
// Choose interpolation point nf+1 (1-based after the increment) and put it in B.x / B.xeval.
BQ_HD void prelim_next_point(BqRef B) {
    const int nfm = BQN(nf);
    const int nfx = BQN(nf) - 3;
    BQN(nf)++;
    const int nf = BQN(nf);
    // with npt = 2n+1 only the first two families of points occur (nfm <= 2n)
    if (nfm >= 1 && nfm <= 3) {
        BQS(stepa) = BQS(rhobeg);
        if (BQ_V(su, nfm) == 0.0) BQS(stepa) = -BQS(stepa);
        BQ_XPT(nf, nfm) = BQS(stepa);
    } else if (nfm > 3) {
        BQS(stepa) = BQ_XPT(nf - 3, nfx);
        BQS(stepb) = -BQS(rhobeg);
        if (BQ_V(sl, nfx) == 0.0) BQS(stepb) = mn(2.0 * BQS(rhobeg), BQ_V(su, nfx));
        if (BQ_V(su, nfx) == 0.0) BQS(stepb) = mx(-2.0 * BQS(rhobeg), BQ_V(sl, nfx));
        BQ_XPT(nf, nfx) = BQS(stepb);
    }
    set_x_from(B, O_xpt + (nf - 1) * 3);
}

// Absorb f at point nf into the initial model.
BQ_HD void prelim_consume(BqRef B, double f) {
    const int nf = BQN(nf), nfm = nf - 1, nfx = nf - 1 - 3;
    const double rhosq = BQS(rhobeg) * BQS(rhobeg);
    BQ_FVAL(nf) = f;
    if (nf == 1) {
        BQS(fbeg) = f;
        BQN(kopt) = 1;
    } else if (f < BQ_FVAL(BQN(kopt))) {
        BQN(kopt) = nf;
    }
    if (nf >= 2 && nf <= 4) {
        BQ_V(gopt, nfm) = (f - BQS(fbeg)) / BQS(stepa);
        // (npt < nf + n never holds for npt = 2n+1)
    } else if (nf >= 5) {
        const int ih = nfx * (nfx + 1) / 2;
        const double temp = (f - BQS(fbeg)) / BQS(stepb);
        const double diff = BQS(stepb) - BQS(stepa);
        BQ_HQ(ih) = 2.0 * (temp - BQ_V(gopt, nfx)) / diff;
        BQ_V(gopt, nfx) = (BQ_V(gopt, nfx) * BQS(stepb) - temp * BQS(stepa)) / diff;
        if (BQS(stepa) * BQS(stepb) < 0.0) {
            if (f < BQ_FVAL(nf - 3)) {
                BQ_FVAL(nf) = BQ_FVAL(nf - 3);
                BQ_FVAL(nf - 3) = f;
                if (BQN(kopt) == nf) BQN(kopt) = nf - 3;
                BQ_XPT(nf - 3, nfx) = BQS(stepb);
                BQ_XPT(nf, nfx) = BQS(stepa);
            }
        }
        BQ_BMAT(1, nfx) = -(BQS(stepa) + BQS(stepb)) / (BQS(stepa) * BQS(stepb));
        BQ_BMAT(nf, nfx) = -0.5 / BQ_XPT(nf - 3, nfx);
        BQ_BMAT(nf - 3, nfx) = -BQ_BMAT(1, nfx) - BQ_BMAT(nf, nfx);
        BQ_ZMAT(1, nfx) = sqrt(2.0) / (BQS(stepa) * BQS(stepb));
        BQ_ZMAT(nf, nfx) = sqrt(0.5) / rhosq;
        BQ_ZMAT(nf - 3, nfx) = -BQ_ZMAT(1, nfx) - BQ_ZMAT(nf, nfx);
    }
}

// ---------------------------------------------------------------- H times s (bobyqa.c:1676-1700)
BQ_HD void hess_mul(BqRef B, const double* s, double* hs) {
    int ih = 0;
    for (int j = 1; j <= 3; j++) {
        hs[j - 1] = 0.0;
        for (int i = 1; i <= j; i++) {
            ++ih;
            if (i < j) hs[j - 1] += BQ_MA(hq, ih - 1) * s[i - 1];
            hs[i - 1] += BQ_MA(hq, ih - 1) * s[j - 1];
        }
    }
    for (int k = 1; k <= 7; k++) {
        if (BQ_MA(pq, k - 1) != 0.0) {
            double temp = 0.0;
            for (int j = 1; j <= 3; j++) temp += BQ_MA(xpt, (k - 1) * 3 + j - 1) * s[j - 1];
            temp *= BQ_MA(pq, k - 1);
            for (int i = 1; i <= 3; i++) hs[i - 1] += temp * BQ_MA(xpt, (k - 1) * 3 + i - 1);
        }
    }
}

// same product with the model held in registers by the caller
BQ_HD void hess_mul_local(const double* hq, const double* pq, const double* xpt, const double* s, double* hs) {
    int ih = 0;
    for (int j = 1; j <= 3; j++) {
        hs[j - 1] = 0.0;
        for (int i = 1; i <= j; i++) {
            ++ih;
            if (i < j) hs[j - 1] += hq[ih - 1] * s[i - 1];
            hs[i - 1] += hq[ih - 1] * s[j - 1];
        }
    }
    for (int k = 1; k <= 7; k++) {
        if (pq[k - 1] != 0.0) {
            double temp = 0.0;
            for (int j = 1; j <= 3; j++) temp += xpt[(k - 1) * 3 + j - 1] * s[j - 1];
            temp *= pq[k - 1];
            for (int i = 1; i <= 3; i++) hs[i - 1] += temp * xpt[(k - 1) * 3 + i - 1];
        }
    }
}

// ---------------------------------------------------------------- trsbox_ (bobyqa.c:1162-1711)
// Approximately minimise the model within ||d|| <= delta and the bounds: truncated conjugate
// gradients with restarts on newly active bounds, then 2-D searches on the boundary.
// Sets B.xnew, B.d, B.gnew, BQS(dsq), BQS(crvmin).
// three float64 values held as named scalars: with a per-lane (divergent) index the compiler would turn a
// local array into scratch memory; the accessors compile to selects, or to nothing for constant indices
struct Vec3Regs {
    double v0, v1, v2;
    BQ_HD double get(int i) const { return i == 0 ? v0 : (i == 1 ? v1 : v2); }
    BQ_HD void set(int i, double x) { if (i == 0) v0 = x; else if (i == 1) v1 = x; else v2 = x; }
};

BQ_HD_BLOCK void trsbox(BqRef B) {
    Vec3Regs xbdi;  // +-1 for variables fixed at a bound, else 0
    double s[3] = {0, 0, 0}, hs[3], hred[3] = {0, 0, 0};
    // read-only model data and the working vectors are held in registers for the whole call
    // (one batch of LDS reads at entry, one write-back of d / gnew / xnew at exit)
    double xopt_[3], sl_[3], su_[3], d_[3], gnew_[3], hq_[6], pq_[7], xpt_[21];
    for (int i = 0; i < 3; i++) { xopt_[i] = BQA(xopt, i); sl_[i] = BQA(sl, i); su_[i] = BQA(su, i); }
    for (int i = 0; i < 6; i++) hq_[i] = BQ_MA(hq, i);
    for (int i = 0; i < 7; i++) pq_[i] = BQ_MA(pq, i);
    for (int i = 0; i < 21; i++) xpt_[i] = BQ_MA(xpt, i);
    int iterc = 0, nact = 0, itermax = 0, iact = 0, isav, iu;
    double delsq, qred = 0.0, crvmin = -1.0, beta = 0.0, stepsq, gredsq = 0.0, resid, ds, shs, temp, blen, stplen,
                  sdec, ggsav = 0.0, dredsq = 0.0, dredg = 0.0, sredg = 0.0, angbd = 0.0, xsav = 0.0, ssq, tempa,
                  tempb, dhs, dhd, redmax, redsav, angt = 0.0, sth, cth, rednew, rdprev = 0.0, rdnext = 0.0;
    for (int i = 0; i < 3; i++) {
        double xb = 0.0;
        if (xopt_[i] <= sl_[i]) {
            if (BQA(gopt, i) >= 0.0) xb = -1.0;
        } else if (xopt_[i] >= su_[i]) {
            if (BQA(gopt, i) <= 0.0) xb = 1.0;
        }
        xbdi.set(i, xb);
        if (xb != 0.0) ++nact;
        d_[i] = 0.0;
        gnew_[i] = BQA(gopt, i);
    }
    delsq = BQS(delta) * BQS(delta);

    enum { CG_RESTART, CG_NEXT, BOUNDARY, ALT_SETUP, ALT_ITER, FINISH };
    int st = CG_RESTART;
    // The lanes of a wavefront that are in here run different patches and need different numbers of conjugate-
    // gradient and boundary iterations.  One loop over "whatever state each lane is in" executes the CG block and
    // the (3x longer) boundary block in every pass while ANY lane still needs either; the state graph has no edge
    // from the boundary search back to CG, so the two phases run as two loops: lanes that finished CG wait for the
    // others, then all boundary searches advance together.  Per lane the operations and their order are unchanged.
#if defined(__HIP_DEVICE_COMPILE__)
#define BQ_ANY(c) __any(c)
#else
#define BQ_ANY(c) (c)
#endif
    while (BQ_ANY(st == CG_RESTART || st == CG_NEXT)) {
        BQ_PROF_SCOPE(18);
        if (st == CG_RESTART) { beta = 0.0; st = CG_NEXT; }
        if (st == CG_NEXT) {
            stepsq = 0.0;
            for (int i = 0; i < 3; i++) {
                if (xbdi.get(i) != 0.0) s[i] = 0.0;
                else if (beta == 0.0) s[i] = -gnew_[i];
                else s[i] = beta * s[i] - gnew_[i];
                stepsq += s[i] * s[i];
            }
            if (stepsq == 0.0) { st = FINISH; continue; }
            if (beta == 0.0) { gredsq = stepsq; itermax = iterc + 3 - nact; }
            if (gredsq * delsq <= qred * 1e-4 * qred) { st = FINISH; continue; }
            hess_mul_local(hq_, pq_, xpt_, s, hs);
            BQ_PROF_COUNT(56, 1);
            resid = delsq; ds = 0.0; shs = 0.0;
            for (int i = 0; i < 3; i++)
                if (xbdi.get(i) == 0.0) { resid -= d_[i] * d_[i]; ds += s[i] * d_[i]; shs += s[i] * hs[i]; }
            if (resid <= 0.0) { st = BOUNDARY; continue; }
            temp = sqrt(stepsq * resid + ds * ds);
            if (ds < 0.0) blen = (temp - ds) / stepsq;
            else blen = resid / (temp + ds);
            stplen = blen;
            if (shs > 0.0) stplen = mn(blen, gredsq / shs);
            iact = 0;
            for (int i = 0; i < 3; i++)
                if (s[i] != 0.0) {
                    const double xsum = xopt_[i] + d_[i];
                    if (s[i] > 0.0) temp = (su_[i] - xsum) / s[i];
                    else temp = (sl_[i] - xsum) / s[i];
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
                    gnew_[i] += stplen * hs[i];
                    if (xbdi.get(i) == 0.0) gredsq += gnew_[i] * gnew_[i];
                    d_[i] += stplen * s[i];
                }
                sdec = mx(stplen * (ggsav - 0.5 * stplen * shs), 0.0);
                qred += sdec;
            }
            if (iact > 0) {
                ++nact;
                for (int i = 0; i < 3; i++)  // index iact-1 differs per lane: selects keep the arrays in registers
                    if (i == iact - 1) {
                        xbdi.set(i, (s[i] < 0.0) ? -1.0 : 1.0);
                        delsq -= d_[i] * d_[i];
                    }
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
    }
    while (BQ_ANY(st == BOUNDARY || st == ALT_SETUP || st == ALT_ITER)) {
        BQ_PROF_SCOPE(19);
        if (st == BOUNDARY) { crvmin = 0.0; st = ALT_SETUP; }
        if (st == ALT_SETUP) {
            if (nact >= 2) { st = FINISH; continue; }
            dredsq = 0.0; dredg = 0.0; gredsq = 0.0;
            for (int i = 0; i < 3; i++) {
                if (xbdi.get(i) == 0.0) {
                    dredsq += d_[i] * d_[i];
                    dredg += d_[i] * gnew_[i];
                    gredsq += gnew_[i] * gnew_[i];
                    s[i] = d_[i];
                } else s[i] = 0.0;
            }
            hess_mul_local(hq_, pq_, xpt_, s, hs);
            for (int i = 0; i < 3; i++) hred[i] = hs[i];
            st = ALT_ITER;
        }
        if (st == ALT_ITER) {
            ++iterc;
            temp = gredsq * dredsq - dredg * dredg;
            if (temp <= qred * 1e-4 * qred) { st = FINISH; continue; }
            temp = sqrt(temp);
            for (int i = 0; i < 3; i++) {
                if (xbdi.get(i) == 0.0) s[i] = (dredg * d_[i] - dredsq * gnew_[i]) / temp;
                else s[i] = 0.0;
            }
            sredg = -temp;
            angbd = 1.0;
            iact = 0;
            bool refix = false;
            for (int i = 0; i < 3; i++) {
                if (xbdi.get(i) == 0.0) {
                    tempa = xopt_[i] + d_[i] - sl_[i];
                    tempb = su_[i] - xopt_[i] - d_[i];
                    if (tempa <= 0.0) { ++nact; xbdi.set(i, -1.0); refix = true; break; }
                    else if (tempb <= 0.0) { ++nact; xbdi.set(i, 1.0); refix = true; break; }
                    ssq = d_[i] * d_[i] + s[i] * s[i];
                    temp = ssq - (xopt_[i] - sl_[i]) * (xopt_[i] - sl_[i]);
                    if (temp > 0.0) {
                        temp = sqrt(temp) - s[i];
                        if (angbd * temp > tempa) { angbd = tempa / temp; iact = i + 1; xsav = -1.0; }
                    }
                    temp = ssq - (su_[i] - xopt_[i]) * (su_[i] - xopt_[i]);
                    if (temp > 0.0) {
                        temp = sqrt(temp) + s[i];
                        if (angbd * temp > tempb) { angbd = tempb / temp; iact = i + 1; xsav = 1.0; }
                    }
                }
            }
            if (refix) { st = ALT_SETUP; continue; }
            hess_mul_local(hq_, pq_, xpt_, s, hs);
            shs = 0.0; dhs = 0.0; dhd = 0.0;
            for (int i = 0; i < 3; i++)
                if (xbdi.get(i) == 0.0) { shs += s[i] * hs[i]; dhs += d_[i] * hs[i]; dhd += d_[i] * hred[i]; }
            redmax = 0.0; isav = 0; redsav = 0.0;
            iu = (int)(angbd * 17. + 3.1);
            BQ_PROF_COUNT(57, 1);
            BQ_PROF_COUNT(58, iu);
#define BQ_ANGLE_STEP(i_, angt_, sth_)                                                        \
    {                                                                                         \
        temp = shs + (angt_) * ((angt_) * dhd - dhs - dhs);                                   \
        rednew = (sth_) * ((angt_) * dredg - sredg - 0.5 * (sth_) * temp);                    \
        if (rednew > redmax) { redmax = rednew; isav = (i_); rdprev = redsav; }              \
        else if ((i_) == isav + 1) rdnext = rednew;                                           \
        redsav = rednew;                                                                      \
    }
            if (angbd == 1.0) {  // iu == 20: tabulated angles, unrolled so that they are literals
#pragma unroll
                for (int i = 1; i <= 20; i++) BQ_ANGLE_STEP(i, kAngleTable.angt[i - 1], kAngleTable.sth[i - 1])
                angt = kAngleTable.angt[19];
            } else {
                for (int i = 1; i <= iu; i++) {
                    angt = angbd * (double)i / (double)iu;
                    sth = (angt + angt) / (1.0 + angt * angt);
                    BQ_ANGLE_STEP(i, angt, sth)
                }
            }
#undef BQ_ANGLE_STEP
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
                gnew_[i] = gnew_[i] + (cth - 1.0) * hred[i] + sth * hs[i];
                if (xbdi.get(i) == 0.0) {
                    d_[i] = cth * d_[i] + sth * s[i];
                    dredg += d_[i] * gnew_[i];
                    gredsq += gnew_[i] * gnew_[i];
                }
                hred[i] = cth * hred[i] + sth * hs[i];
            }
            qred += sdec;
            if (iact > 0 && isav == iu) {
                ++nact;
                for (int i = 0; i < 3; i++) if (i == iact - 1) xbdi.set(i, xsav);
                st = ALT_SETUP;
                continue;
            }
            if (sdec > qred * .01) { st = ALT_ITER; continue; }
            st = FINISH;
        }
    }
#undef BQ_ANY
    BQ_PROF_COUNT(59, 1);
    double dsq = 0.0;
    for (int i = 0; i < 3; i++) {
        double xn = mx(mn(xopt_[i] + d_[i], su_[i]), sl_[i]);
        if (xbdi.get(i) == -1.0) xn = sl_[i];
        if (xbdi.get(i) == 1.0) xn = su_[i];
        d_[i] = xn - xopt_[i];
        dsq += d_[i] * d_[i];
        BQA(xnew, i) = xn; BQA(d, i) = d_[i]; BQA(gnew, i) = gnew_[i];
    }
    BQS(dsq) = dsq;
    BQS(crvmin) = crvmin;
}

// ---------------------------------------------------------------- altmov_ (bobyqa.c:744-1160)
// Geometry-improving step for interpolation point knew: B.xnew (line through xopt and another
// point), B.xalt (constrained Cauchy step), BQS(alpha), BQS(cauchy).
BQ_HD_BLOCK void altmov(BqRef B) {
    const double cst = 1.0 + sqrt(2.);
    double glag[3], hcol[7], w[3], wsave[3] = {0, 0, 0};
    const int knew = BQN(knew), kopt = BQN(kopt);
    const double adelt = BQS(adelt);
    // the model data this block reads, in registers for the whole call (nothing of it is written here)
    double xp_[7][3], xo_[3], sl_[3], su_[3];
    double ha;
    {
        double z[7][3], zk[3];
        for (int k = 0; k < 7; k++)
            for (int j = 0; j < 3; j++) { z[k][j] = BQ_ZMAT(k + 1, j + 1); xp_[k][j] = BQ_XPT(k + 1, j + 1); }
        for (int j = 0; j < 3; j++) { zk[j] = BQ_ZMAT(knew, j + 1); glag[j] = BQ_BMAT(knew, j + 1); }
        for (int j = 0; j < 3; j++) { xo_[j] = BQ_V(xopt, j + 1); sl_[j] = BQ_V(sl, j + 1); su_[j] = BQ_V(su, j + 1); }
        for (int k = 1; k <= 7; k++) hcol[k - 1] = 0.0;
        for (int j = 1; j <= 3; j++) {
            const double temp = zk[j - 1];
            for (int k = 1; k <= 7; k++) hcol[k - 1] += temp * z[k - 1][j - 1];
        }
        // = hcol[knew-1] (same operations in the same order), without a per-lane index into a local array
        double al = 0.0;
        for (int j = 1; j <= 3; j++) al += zk[j - 1] * zk[j - 1];
        BQS(alpha) = al;
        ha = 0.5 * al;
    }
    for (int k = 1; k <= 7; k++) {
        double temp = 0.0;
        for (int j = 1; j <= 3; j++) temp += xp_[k - 1][j - 1] * xo_[j - 1];
        temp = hcol[k - 1] * temp;
        for (int i = 1; i <= 3; i++) glag[i - 1] += temp * xp_[k - 1][i - 1];
    }
    double presav = 0.0, step = 0.0, stpsav = 0.0, vlag, temp, tempa, tempb, tempd, diff;
    int ksav = kopt, ibdsav = 0, isbd;
    for (int k = 1; k <= 7; k++) {
        if (k == kopt) continue;
        double dderiv = 0.0, distsq = 0.0;
        for (int i = 1; i <= 3; i++) {
            temp = xp_[k - 1][i - 1] - xo_[i - 1];
            dderiv += glag[i - 1] * temp;
            distsq += temp * temp;
        }
        double subd = adelt / sqrt(distsq);
        double slbd = -subd;
        int ilbd = 0, iubd = 0;
        const double sumin = mn(1.0, subd);
        for (int i = 1; i <= 3; i++) {
            temp = xp_[k - 1][i - 1] - xo_[i - 1];
            if (temp > 0.0) {
                if (slbd * temp < sl_[i - 1] - xo_[i - 1]) { slbd = (sl_[i - 1] - xo_[i - 1]) / temp; ilbd = -i; }
                if (subd * temp > su_[i - 1] - xo_[i - 1]) { subd = mx(sumin, (su_[i - 1] - xo_[i - 1]) / temp); iubd = i; }
            } else if (temp < 0.0) {
                if (slbd * temp > su_[i - 1] - xo_[i - 1]) { slbd = (su_[i - 1] - xo_[i - 1]) / temp; ilbd = i; }
                if (subd * temp < sl_[i - 1] - xo_[i - 1]) { subd = mx(sumin, (sl_[i - 1] - xo_[i - 1]) / temp); iubd = -i; }
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
        temp = xo_[i - 1] + stpsav * (BQ_XPT(ksav, i) - xo_[i - 1]);
        BQ_V(xnew, i) = mx(sl_[i - 1], mn(su_[i - 1], temp));
    }
    if (ibdsav < 0) BQ_V(xnew, -ibdsav) = BQ_V(sl, -ibdsav);
    if (ibdsav > 0) BQ_V(xnew, ibdsav) = BQ_V(su, ibdsav);

    const double bigstp = adelt + adelt;
    double csave = 0.0, wfixsq, ggfree, wsqsav, gw, curv, scale;
    double xalt_[3] = {0, 0, 0}, cauchy = 0.0;  // written to the state once, at the end
    for (int iflag = 0; iflag <= 1; iflag++) {
        wfixsq = 0.0; ggfree = 0.0;
        for (int i = 1; i <= 3; i++) {
            w[i - 1] = 0.0;
            tempa = mn(xo_[i - 1] - sl_[i - 1], glag[i - 1]);
            tempb = mx(xo_[i - 1] - su_[i - 1], glag[i - 1]);
            if (tempa > 0.0 || tempb < 0.0) { w[i - 1] = bigstp; ggfree += glag[i - 1] * glag[i - 1]; }
        }
        if (ggfree == 0.0) {
            // (bobyqa.c returns here with xalt as the first pass left it, or untouched)
            if (iflag == 1) for (int i = 0; i < 3; i++) BQA(xalt, i) = xalt_[i];
            BQS(cauchy) = 0.0;
            return;
        }
        for (;;) {
            temp = adelt * adelt - wfixsq;
            if (!(temp > 0.0)) break;
            wsqsav = wfixsq;
            step = sqrt(temp / ggfree);
            ggfree = 0.0;
            for (int i = 1; i <= 3; i++) {
                if (w[i - 1] == bigstp) {
                    temp = xo_[i - 1] - step * glag[i - 1];
                    if (temp <= sl_[i - 1]) { w[i - 1] = sl_[i - 1] - xo_[i - 1]; wfixsq += w[i - 1] * w[i - 1]; }
                    else if (temp >= su_[i - 1]) { w[i - 1] = su_[i - 1] - xo_[i - 1]; wfixsq += w[i - 1] * w[i - 1]; }
                    else ggfree += glag[i - 1] * glag[i - 1];
                }
            }
            if (!(wfixsq > wsqsav && ggfree > 0.0)) break;
        }
        gw = 0.0;
        for (int i = 1; i <= 3; i++) {
            if (w[i - 1] == bigstp) {
                w[i - 1] = -step * glag[i - 1];
                xalt_[i - 1] = mx(sl_[i - 1], mn(su_[i - 1], xo_[i - 1] + w[i - 1]));
            } else if (w[i - 1] == 0.0) xalt_[i - 1] = xo_[i - 1];
            else if (glag[i - 1] > 0.0) xalt_[i - 1] = sl_[i - 1];
            else xalt_[i - 1] = su_[i - 1];
            gw += glag[i - 1] * w[i - 1];
        }
        curv = 0.0;
        for (int k = 1; k <= 7; k++) {
            temp = 0.0;
            for (int j = 1; j <= 3; j++) temp += xp_[k - 1][j - 1] * w[j - 1];
            curv += hcol[k - 1] * temp * temp;
        }
        if (iflag == 1) curv = -curv;
        if (curv > -gw && curv < -cst * gw) {
            scale = -gw / curv;
            for (int i = 1; i <= 3; i++) {
                temp = xo_[i - 1] + scale * w[i - 1];
                xalt_[i - 1] = mx(sl_[i - 1], mn(su_[i - 1], temp));
            }
            cauchy = (0.5 * gw * scale) * (0.5 * gw * scale);
        } else {
            cauchy = (gw + 0.5 * curv) * (gw + 0.5 * curv);
        }
        if (iflag == 0) {
            for (int i = 0; i < 3; i++) { glag[i] = -glag[i]; wsave[i] = xalt_[i]; }
            csave = cauchy;
        }
    }
    if (csave > cauchy) {
        for (int i = 0; i < 3; i++) xalt_[i] = wsave[i];
        cauchy = csave;
    }
    for (int i = 0; i < 3; i++) BQA(xalt, i) = xalt_[i];
    BQS(cauchy) = cauchy;
}

// ---------------------------------------------------------------- update_ (bobyqa.c:18-141)
// Update BMAT and ZMAT for the move of interpolation point knew (uses vlag, beta, denom).
BQ_HD_BLOCK void update(BqRef B) {
    // Memory traffic (here and in the other blocks of the main loop): every array the block needs is read from the
    // state ONCE, in one batch at the top, worked on in registers and written back once at the end.  The state lives
    // in LDS; read-modify-write through it costs a ~100-cycle round trip per dependent access and a wavefront in
    // here has nothing else to issue.  Rows selected by a per-lane index (knew) are read separately and carried
    // along: the operations applied to the copy are the ones the loop applies to that row.  The arithmetic and its
    // order are bobyqa.c's.
    const int knew = BQN(knew);
    double z[7][3], vl[10], bm[10][3], zk[3], bk[3];
    for (int k = 0; k < 7; k++)
        for (int j = 0; j < 3; j++) z[k][j] = BQ_ZMAT(k + 1, j + 1);
    for (int j = 0; j < 3; j++) { zk[j] = BQ_ZMAT(knew, j + 1); bk[j] = BQ_BMAT(knew, j + 1); }
    for (int i = 0; i < 10; i++) vl[i] = BQ_VLAG(i + 1);
    for (int i = 0; i < 10; i++)
        for (int j = 0; j < 3; j++) bm[i][j] = BQ_BMAT(i + 1, j + 1);
    const double tau = BQ_VLAG(knew);
    const double denom = BQS(denom), beta = BQS(beta);
    double w[10];
    double ztest = 0.0;
    for (int k = 1; k <= 7; k++)
        for (int j = 1; j <= 3; j++) ztest = mx(ztest, fabs(z[k - 1][j - 1]));
    ztest *= 1e-20;
    for (int j = 2; j <= 3; j++) {
        if (fabs(zk[j - 1]) > ztest) {
            double temp = sqrt(zk[0] * zk[0] + zk[j - 1] * zk[j - 1]);
            const double tempa = zk[0] / temp;
            const double tempb = zk[j - 1] / temp;
            for (int i = 1; i <= 7; i++) {
                temp = tempa * z[i - 1][0] + tempb * z[i - 1][j - 1];
                z[i - 1][j - 1] = tempa * z[i - 1][j - 1] - tempb * z[i - 1][0];
                z[i - 1][0] = temp;
            }
            temp = tempa * zk[0] + tempb * zk[j - 1];  // row knew of the loop above
            zk[0] = temp;
        }
        zk[j - 1] = 0.0;
        for (int i = 1; i <= 7; i++) if (i == knew) z[i - 1][j - 1] = 0.0;
    }
    for (int i = 1; i <= 7; i++) w[i - 1] = zk[0] * z[i - 1][0];
    const double alpha = zk[0] * zk[0];  // = w[knew-1], without a per-lane index into a local array
    for (int i = 1; i <= 7; i++) if (i == knew) vl[i - 1] = tau - 1.0;
    double temp = sqrt(denom);
    double tempb = zk[0] / temp;
    double tempa = tau / temp;
    for (int i = 1; i <= 7; i++) z[i - 1][0] = tempa * z[i - 1][0] - tempb * vl[i - 1];
    for (int j = 1; j <= 3; j++) {
        const int jp = 7 + j;
        w[jp - 1] = bk[j - 1];
        tempa = (alpha * vl[jp - 1] - tau * w[jp - 1]) / denom;
        tempb = (-beta * w[jp - 1] - tau * vl[jp - 1]) / denom;
        for (int i = 1; i <= jp; i++) {
            bm[i - 1][j - 1] = bm[i - 1][j - 1] + tempa * vl[i - 1] + tempb * w[i - 1];
            if (i > 7) bm[jp - 1][i - 7 - 1] = bm[i - 1][j - 1];
        }
    }
    for (int k = 0; k < 7; k++)
        for (int j = 0; j < 3; j++) BQ_ZMAT(k + 1, j + 1) = z[k][j];
    for (int i = 0; i < 10; i++)
        for (int j = 0; j < 3; j++) BQ_BMAT(i + 1, j + 1) = bm[i][j];
    BQ_VLAG(knew) = tau - 1.0;
}

// pick_knew on per-point quantities the caller formed from its register copies: hdiag[k] = sum_j zmat(k,j)^2 and
// distsq[k] = |xpt(k) - from|^2 (each accumulated in j order, as the loop of bobyqa.c does); knew and denom are
// written to the state once, at the end
BQ_HD void pick_knew_core(BqRef B, const double* hdiag, const double* distsq, const double* vl, double beta,
                          bool skip_kopt, double& scaden, double& biglsq) {
    const double delta = BQS(delta);
    const double delsq = delta * delta;
    const int kopt = BQN(kopt);
    scaden = 0.0; biglsq = 0.0;
    int knew = 0;
    double denom = 0.0;
    for (int k = 1; k <= 7; k++) {
        if (skip_kopt && k == kopt) continue;
        const double den = beta * hdiag[k - 1] + vl[k - 1] * vl[k - 1];
        const double q = distsq[k - 1] / delsq;
        const double temp = mx(1.0, q * q);
        if (temp * den > scaden) { scaden = temp * den; knew = k; denom = den; }
        biglsq = mx(biglsq, temp * (vl[k - 1] * vl[k - 1]));
    }
    BQN(knew) = knew;
    if (knew > 0) BQS(denom) = denom;
}

// Select the interpolation point to drop for a trust-region step (bobyqa.c:2497-2545, 2648-2690):
// maximise max(1, (dist/delta)^4) * |denominator|.  `from` is xopt before the evaluation and
// xnew after an improving one.
BQ_HD void pick_knew(BqRef B, int from_off, bool skip_kopt, double& scaden, double& biglsq) {
    double hdiag[7], distsq[7], vl[7];
    {
        double z[7][3], xp[7][3], from[3];
        for (int k = 0; k < 7; k++)
            for (int j = 0; j < 3; j++) { z[k][j] = BQ_ZMAT(k + 1, j + 1); xp[k][j] = BQ_XPT(k + 1, j + 1); }
        for (int k = 0; k < 7; k++) vl[k] = BQ_VLAG(k + 1);
        for (int j = 0; j < 3; j++) from[j] = BQ_EITHER(from_off + j);
        for (int k = 0; k < 7; k++) {
            double h = 0.0, ds = 0.0;
            for (int j = 0; j < 3; j++) h += z[k][j] * z[k][j];
            for (int j = 0; j < 3; j++) { const double t = xp[k][j] - from[j]; ds += t * t; }
            hdiag[k] = h; distsq[k] = ds;
        }
    }
    pick_knew_core(B, hdiag, distsq, vl, BQS(beta), skip_kopt, scaden, biglsq);
}

// ---------------------------------------------------------------- rescue_ (bobyqa.c:143-742)
// Called (label L190) when rounding errors have damaged the denominator of the updating formula:
// xbase moves to xopt, BMAT/ZMAT are rebuilt from provisional points along the coordinate directions,
// original interpolation points are reinstated while the denominators stay healthy, and the remaining
// provisional points are evaluated (one objective evaluation each, hence the resumable loop).

BQ_HD int rescue_finish(BqRef B, int rc2) {  // bobyqa.c:2340-2365
    BQS(xoptsq) = 0.0;
    if (BQN(kopt) != BQN(kbase)) {
        for (int i = 1; i <= 3; i++) {
            BQ_V(xopt, i) = BQ_XPT(BQN(kopt), i);
            BQS(xoptsq) += BQ_V(xopt, i) * BQ_V(xopt, i);
        }
    }
    if (rc2 != BQ_SUCCESS) { BQN(rc) = rc2; return L720; }
    BQN(nresc) = BQN(nevals);
    if (BQN(nfsav) < BQN(nevals)) { BQN(nfsav) = BQN(nevals); return L20; }
    if (BQN(ntrits) > 0) return L60;
    return L210;
}

// the L260 loop of rescue_ (bobyqa.c:560-651): next provisional point that still needs a value
BQ_HD_NOINLINE int rescue_next(BqRef B_in) {
    BqRef B = BQ_UNIFORM(B_in);
    for (; BQN(kpt) <= 7; BQN(kpt)++) {
        const int kpt = BQN(kpt);
        if (BQ_PTSID(kpt) == 0.0) continue;
        if (BQN(maxeval) > 0 && BQN(nevals) >= BQN(maxeval)) return rescue_finish(B, BQ_MAXEVAL_REACHED);
        double w[3];
        int ih = 0;
        for (int j = 1; j <= 3; j++) {
            w[j - 1] = BQ_XPT(kpt, j);
            BQ_XPT(kpt, j) = 0.0;
            const double temp = BQ_PQ(kpt) * w[j - 1];
            for (int i = 1; i <= j; i++) { ++ih; BQ_HQ(ih) += temp * w[i - 1]; }
        }
        BQ_PQ(kpt) = 0.0;
        const int ip = (int)BQ_PTSID(kpt);
        const int iq = (int)(4.0 * BQ_PTSID(kpt) - (double)(ip * 4));
        double xp = 0.0, xq = 0.0;
        if (ip > 0) { xp = BQ_PA(1, ip); BQ_XPT(kpt, ip) = xp; }
        if (iq > 0) {
            xq = BQ_PA(1, iq);
            if (ip == 0) xq = BQ_PA(2, iq);
            BQ_XPT(kpt, iq) = xq;
        }
        // value of the current model at the new point
        double vquad = BQS(fbase_r);
        int ihp = 0, ihq = 0;
        if (ip > 0) {
            ihp = (ip + ip * ip) / 2;
            vquad += xp * (BQ_V(gopt, ip) + 0.5 * xp * BQ_HQ(ihp));
        }
        if (iq > 0) {
            ihq = (iq + iq * iq) / 2;
            vquad += xq * (BQ_V(gopt, iq) + 0.5 * xq * BQ_HQ(ihq));
            if (ip > 0) {
                const int dd = ip - iq;
                const int iw = (ihp >= ihq ? ihp : ihq) - (dd < 0 ? -dd : dd);
                vquad += xp * xq * BQ_HQ(iw);
            }
        }
        for (int k = 1; k <= 7; k++) {
            double temp = 0.0;
            if (ip > 0) temp += xp * BQ_XPT(k, ip);
            if (iq > 0) temp += xq * BQ_XPT(k, iq);
            vquad += 0.5 * BQ_PQ(k) * temp * temp;
        }
        BQS(vquad_r) = vquad;
        set_x_from(B, O_xpt + (kpt - 1) * 3);
        BQN(nevals)++;
        BQN(phase) = PH_RESCUE;
        return LRET_CONT;
    }
    return rescue_finish(B, BQ_SUCCESS);
}

// after the evaluation of provisional point kpt (bobyqa.c:652-700)
BQ_HD_NOINLINE int rescue_consume(BqRef B_in, double f) {
    BqRef B = BQ_UNIFORM(B_in);
    const int kpt = BQN(kpt);
    BQ_FVAL(kpt) = f;
    if (f < BQ_FVAL(BQN(kopt))) BQN(kopt) = kpt;
    if (BQN(maxeval) > 0 && BQN(nevals) >= BQN(maxeval)) return rescue_finish(B, BQ_MAXEVAL_REACHED);
    const double diff = f - BQS(vquad_r);
    for (int i = 1; i <= 3; i++) BQ_V(gopt, i) += diff * BQ_BMAT(kpt, i);
    for (int k = 1; k <= 7; k++) {
        double sum = 0.0;
        for (int j = 1; j <= 3; j++) sum += BQ_ZMAT(k, j) * BQ_ZMAT(kpt, j);
        const double temp = diff * sum;
        if (BQ_PTSID(k) == 0.0) {
            BQ_PQ(k) += temp;
        } else {
            const int ip = (int)BQ_PTSID(k);
            const int iq = (int)(4.0 * BQ_PTSID(k) - (double)(ip * 4));
            const int ihq = (iq * iq + iq) / 2;
            if (ip == 0) {
                BQ_HQ(ihq) += temp * (BQ_PA(2, iq) * BQ_PA(2, iq));
            } else {
                const int ihp = (ip * ip + ip) / 2;
                BQ_HQ(ihp) += temp * (BQ_PA(1, ip) * BQ_PA(1, ip));
                if (iq > 0) {
                    BQ_HQ(ihq) += temp * (BQ_PA(1, iq) * BQ_PA(1, iq));
                    const int dd = iq - ip;
                    const int iw = (ihp >= ihq ? ihp : ihq) - (dd < 0 ? -dd : dd);
                    BQ_HQ(iw) += temp * BQ_PA(1, ip) * BQ_PA(1, iq);
                }
            }
        }
    }
    BQ_PTSID(kpt) = 0.0;
    BQN(kpt) = kpt + 1;
    return rescue_next(B);
}

// everything of rescue_ before its evaluation loop (bobyqa.c:240-558)
#ifndef BQ_ON_RESCUE
#define BQ_ON_RESCUE() do { } while (0)  // test hook (tests/native/bq_host.cpp counts the calls)
#endif
BQ_HD_NOINLINE int rescue_begin(BqRef B_in) {
    BqRef B = BQ_UNIFORM(B_in);
    BQ_ON_RESCUE();
    const double sfrac = 0.5 / 4.0;
    BQN(nfsav) = BQN(nevals);
    BQN(kbase) = BQN(kopt);
    const int kopt = BQN(kopt);
    const int knew_main = BQN(knew);  // bobyqb_'s KNEW survives the call (rescue_ has its own)
    double wd[8], w[4], wv[4], ww[8];
    double sumpq = 0.0, winc = 0.0;
    for (int k = 1; k <= 7; k++) {
        double distsq = 0.0;
        for (int j = 1; j <= 3; j++) {
            BQ_XPT(k, j) -= BQ_V(xopt, j);
            distsq += BQ_XPT(k, j) * BQ_XPT(k, j);
        }
        sumpq += BQ_PQ(k);
        wd[k] = distsq;
        winc = mx(winc, distsq);
        for (int j = 1; j <= 3; j++) BQ_ZMAT(k, j) = 0.0;
    }
    int ih = 0;
    for (int j = 1; j <= 3; j++) {
        w[j] = 0.5 * sumpq * BQ_V(xopt, j);
        for (int k = 1; k <= 7; k++) w[j] += BQ_PQ(k) * BQ_XPT(k, j);
        for (int i = 1; i <= j; i++) {
            ++ih;
            BQ_HQ(ih) = BQ_HQ(ih) + w[i] * BQ_V(xopt, j) + w[j] * BQ_V(xopt, i);
        }
    }
    for (int j = 1; j <= 3; j++) {
        BQ_V(xbase, j) += BQ_V(xopt, j);
        BQ_V(sl, j) -= BQ_V(xopt, j);
        BQ_V(su, j) -= BQ_V(xopt, j);
        BQ_V(xopt, j) = 0.0;
        BQ_PA(1, j) = mn(BQS(delta), BQ_V(su, j));
        BQ_PA(2, j) = mx(-BQS(delta), BQ_V(sl, j));
        if (BQ_PA(1, j) + BQ_PA(2, j) < 0.0) {
            const double temp = BQ_PA(1, j);
            BQ_PA(1, j) = BQ_PA(2, j);
            BQ_PA(2, j) = temp;
        }
        if (fabs(BQ_PA(2, j)) < 0.5 * fabs(BQ_PA(1, j))) BQ_PA(2, j) = 0.5 * BQ_PA(1, j);
        for (int i = 1; i <= 10; i++) BQ_BMAT(i, j) = 0.0;
    }
    BQS(fbase_r) = BQ_FVAL(kopt);
    BQ_PTSID(1) = sfrac;
    for (int j = 1; j <= 3; j++) {
        const int jp = j + 1, jpn = jp + 3;  // jpn <= npt always holds for npt = 2n+1
        BQ_PTSID(jp) = (double)j + sfrac;
        BQ_PTSID(jpn) = (double)j / 4.0 + sfrac;
        const double temp = 1.0 / (BQ_PA(1, j) - BQ_PA(2, j));
        BQ_BMAT(jp, j) = -temp + 1.0 / BQ_PA(1, j);
        BQ_BMAT(jpn, j) = temp + 1.0 / BQ_PA(2, j);
        BQ_BMAT(1, j) = -BQ_BMAT(jp, j) - BQ_BMAT(jpn, j);
        BQ_ZMAT(1, j) = sqrt(2.) / fabs(BQ_PA(1, j) * BQ_PA(2, j));
        BQ_ZMAT(jp, j) = BQ_ZMAT(1, j) * BQ_PA(2, j) * temp;
        BQ_ZMAT(jpn, j) = -BQ_ZMAT(1, j) * BQ_PA(1, j) * temp;
    }
    // (npt = 2n+1: no further provisional points with two non-zero coordinates)
    int nrem = 7, kold = 1, knew = kopt;
    bool reinstate = true;  // enter at label 80
    for (;;) {
        if (reinstate) {  // label 80: exchange PTSID(KOLD) with PTSID(KNEW)
            for (int j = 1; j <= 3; j++) {
                double temp = BQ_BMAT(kold, j); BQ_BMAT(kold, j) = BQ_BMAT(knew, j); BQ_BMAT(knew, j) = temp;
                temp = BQ_ZMAT(kold, j); BQ_ZMAT(kold, j) = BQ_ZMAT(knew, j); BQ_ZMAT(knew, j) = temp;
            }
            BQ_PTSID(kold) = BQ_PTSID(knew);
            BQ_PTSID(knew) = 0.0;
            wd[knew] = 0.0;
            --nrem;
            if (knew != kopt) {
                const double temp = BQ_VLAG(kold);
                BQ_VLAG(kold) = BQ_VLAG(knew);
                BQ_VLAG(knew) = temp;
                BQN(knew) = knew;  // update() reads knew / beta / denom from the state
                update(B);
                if (nrem == 0) break;  // label 350
                for (int k = 1; k <= 7; k++) wd[k] = fabs(wd[k]);
            }
        }
        // label 120: nearest original point not yet reinstated
        double dsqmin = 0.0;
        for (int k = 1; k <= 7; k++)
            if (wd[k] > 0.0)
                if (dsqmin == 0.0 || wd[k] < dsqmin) { knew = k; dsqmin = wd[k]; }
        if (dsqmin == 0.0) {  // label 260
            BQN(knew) = knew_main;
            BQN(kpt) = 1;
            return rescue_next(B);
        }
        for (int j = 1; j <= 3; j++) wv[j] = BQ_XPT(knew, j);
        for (int k = 1; k <= 7; k++) {
            double sum = 0.0;
            if (k == kopt) {
            } else if (BQ_PTSID(k) == 0.0) {
                for (int j = 1; j <= 3; j++) sum += wv[j] * BQ_XPT(k, j);
            } else {
                const int ip = (int)BQ_PTSID(k);
                if (ip > 0) sum = wv[ip] * BQ_PA(1, ip);
                const int iq = (int)(4.0 * BQ_PTSID(k) - (double)(ip * 4));
                if (iq > 0) {
                    const int iw = (ip == 0) ? 2 : 1;
                    sum += wv[iq] * BQ_PA(iw, iq);
                }
            }
            ww[k] = 0.5 * sum * sum;
        }
        for (int k = 1; k <= 7; k++) {
            double sum = 0.0;
            for (int j = 1; j <= 3; j++) sum += BQ_BMAT(k, j) * wv[j];
            BQ_VLAG(k) = sum;
        }
        double beta = 0.0;
        for (int j = 1; j <= 3; j++) {
            double sum = 0.0;
            for (int k = 1; k <= 7; k++) sum += BQ_ZMAT(k, j) * ww[k];
            beta -= sum * sum;
            for (int k = 1; k <= 7; k++) BQ_VLAG(k) += sum * BQ_ZMAT(k, j);
        }
        double bsum = 0.0, distsq = 0.0;
        for (int j = 1; j <= 3; j++) {
            double sum = 0.0;
            for (int k = 1; k <= 7; k++) sum += BQ_BMAT(k, j) * ww[k];
            const int jp = j + 7;
            bsum += sum * wv[j];
            for (int ip = 8; ip <= 10; ip++) sum += BQ_BMAT(ip, j) * wv[ip - 7];
            bsum += sum * wv[j];
            BQ_VLAG(jp) = sum;
            distsq += BQ_XPT(knew, j) * BQ_XPT(knew, j);
        }
        beta = 0.5 * distsq * distsq + beta - bsum;
        BQ_VLAG(kopt) += 1.0;
        double denom = 0.0, vlmxsq = 0.0;
        for (int k = 1; k <= 7; k++) {
            if (BQ_PTSID(k) != 0.0) {
                double hdiag = 0.0;
                for (int j = 1; j <= 3; j++) hdiag += BQ_ZMAT(k, j) * BQ_ZMAT(k, j);
                const double den = beta * hdiag + BQ_VLAG(k) * BQ_VLAG(k);
                if (den > denom) { kold = k; denom = den; }
            }
            vlmxsq = mx(vlmxsq, BQ_VLAG(k) * BQ_VLAG(k));
        }
        if (denom <= vlmxsq * .01) {
            wd[knew] = -wd[knew] - winc;
            reinstate = false;  // back to label 120
            continue;
        }
        BQS(beta) = beta;
        BQS(denom) = denom;
        reinstate = true;  // label 80
    }
    // label 350: every original point was reinstated, nothing to evaluate
    BQN(knew) = knew_main;
    return rescue_finish(B, BQ_SUCCESS);
}

// label 90: shift xbase to xopt when the step is small relative to |xopt| (bobyqa.c:2246-2322)
BQ_HD_BLOCK int label90(BqRef B) {
    const double xoptsq = BQS(xoptsq);
    if (BQS(dsq) <= xoptsq * .001) {
        const double fracsq = xoptsq * .25;
        double sumpq = 0.0;
        double w[3], wn[7], vl[7], xo[3], pq[7], xp[7][3], bm[10][3];
        for (int j = 0; j < 3; j++) xo[j] = BQ_V(xopt, j + 1);
        for (int k = 0; k < 7; k++) {
            pq[k] = BQ_PQ(k + 1);
            for (int j = 0; j < 3; j++) xp[k][j] = BQ_XPT(k + 1, j + 1);
        }
        for (int i = 0; i < 10; i++)
            for (int j = 0; j < 3; j++) bm[i][j] = BQ_BMAT(i + 1, j + 1);
        for (int k = 1; k <= 7; k++) {
            sumpq += pq[k - 1];
            double sum = -0.5 * xoptsq;
            for (int i = 1; i <= 3; i++) sum += xp[k - 1][i - 1] * xo[i - 1];
            wn[k - 1] = sum;
            const double temp = fracsq - 0.5 * sum;
            for (int i = 1; i <= 3; i++) {
                w[i - 1] = bm[k - 1][i - 1];
                vl[i - 1] = sum * xp[k - 1][i - 1] + temp * xo[i - 1];
                const int ip = 7 + i;
                for (int j = 1; j <= i; j++)
                    bm[ip - 1][j - 1] = bm[ip - 1][j - 1] + w[i - 1] * vl[j - 1] + vl[i - 1] * w[j - 1];
            }
        }
        BQ_STAGE();
        {
            double z[7][3];
            for (int k = 0; k < 7; k++)
                for (int j = 0; j < 3; j++) z[k][j] = BQ_ZMAT(k + 1, j + 1);
            for (int jj = 1; jj <= 3; jj++) {
                double sumz = 0.0, sumw = 0.0;
                for (int k = 1; k <= 7; k++) {
                    sumz += z[k - 1][jj - 1];
                    vl[k - 1] = wn[k - 1] * z[k - 1][jj - 1];
                    sumw += vl[k - 1];
                }
                for (int j = 1; j <= 3; j++) {
                    double sum = (fracsq * sumz - 0.5 * sumw) * xo[j - 1];
                    for (int k = 1; k <= 7; k++) sum += vl[k - 1] * xp[k - 1][j - 1];
                    w[j - 1] = sum;
                    for (int k = 1; k <= 7; k++) bm[k - 1][j - 1] += sum * z[k - 1][jj - 1];
                }
                for (int i = 1; i <= 3; i++) {
                    const int ip = i + 7;
                    const double temp = w[i - 1];
                    for (int j = 1; j <= i; j++) bm[ip - 1][j - 1] += temp * w[j - 1];
                }
            }
        }
        BQ_STAGE();
        {
            double hq[6];
            for (int i = 0; i < 6; i++) hq[i] = BQ_HQ(i + 1);
            int ih = 0;
            for (int j = 1; j <= 3; j++) {
                w[j - 1] = -0.5 * sumpq * xo[j - 1];
                for (int k = 1; k <= 7; k++) {
                    w[j - 1] += pq[k - 1] * xp[k - 1][j - 1];
                    xp[k - 1][j - 1] -= xo[j - 1];
                }
                for (int i = 1; i <= j; i++) {
                    ++ih;
                    hq[ih - 1] = hq[ih - 1] + w[i - 1] * xo[j - 1] + xo[i - 1] * w[j - 1];
                    bm[7 + i - 1][j - 1] = bm[7 + j - 1][i - 1];
                }
            }
            for (int i = 0; i < 6; i++) BQ_HQ(i + 1) = hq[i];
        }
        for (int k = 0; k < 7; k++)
            for (int j = 0; j < 3; j++) BQ_XPT(k + 1, j + 1) = xp[k][j];
        for (int i = 0; i < 10; i++)
            for (int j = 0; j < 3; j++) BQ_BMAT(i + 1, j + 1) = bm[i][j];
        for (int i = 1; i <= 3; i++) {
            BQ_V(xbase, i) += xo[i - 1];
            BQ_V(xnew, i) -= xo[i - 1];
            BQ_V(sl, i) -= xo[i - 1];
            BQ_V(su, i) -= xo[i - 1];
            BQ_V(xopt, i) = 0.0;
        }
        BQS(xoptsq) = 0.0;
    }
    return (BQN(ntrits) == 0) ? L210 : L230;
}

// label 230: vlag, beta for the step d (bobyqa.c:2390-2440), then knew / denominators (bobyqa.c:2441-2556)
BQ_HD_BLOCK int label230(BqRef B) {
    double w[7], vl[10], dd[3], xo[3], hdiag[7], distsq[7];
    for (int j = 0; j < 3; j++) { dd[j] = BQ_V(d, j + 1); xo[j] = BQ_V(xopt, j + 1); }
    const double xoptsq = BQS(xoptsq);
    const int ntrits = BQN(ntrits), kopt = BQN(kopt), knew0 = BQN(knew);
    const double alpha = BQS(alpha), cauchy = BQS(cauchy);
    double bm[10][3];
    for (int i = 0; i < 10; i++)
        for (int j = 0; j < 3; j++) bm[i][j] = BQ_BMAT(i + 1, j + 1);
    {
        double xp[7][3];
        for (int k = 0; k < 7; k++)
            for (int j = 0; j < 3; j++) xp[k][j] = BQ_XPT(k + 1, j + 1);
        for (int k = 1; k <= 7; k++) {
            double suma = 0.0, sumb = 0.0, sum = 0.0;
            for (int j = 1; j <= 3; j++) {
                suma += xp[k - 1][j - 1] * dd[j - 1];
                sumb += xp[k - 1][j - 1] * xo[j - 1];
                sum += bm[k - 1][j - 1] * dd[j - 1];
            }
            w[k - 1] = suma * (0.5 * suma + sumb);
            vl[k - 1] = sum;
            BQ_WS(k - 1) = suma;
            double ds = 0.0;  // pick_knew's distance of point k from xopt
            for (int j = 1; j <= 3; j++) { const double t = xp[k - 1][j - 1] - xo[j - 1]; ds += t * t; }
            distsq[k - 1] = ds;
        }
    }
    BQ_STAGE();
    double beta = 0.0;
    {
        double z[7][3];
        for (int k = 0; k < 7; k++)
            for (int j = 0; j < 3; j++) z[k][j] = BQ_ZMAT(k + 1, j + 1);
        for (int jj = 1; jj <= 3; jj++) {
            double sum = 0.0;
            for (int k = 1; k <= 7; k++) sum += z[k - 1][jj - 1] * w[k - 1];
            beta -= sum * sum;
            for (int k = 1; k <= 7; k++) vl[k - 1] += sum * z[k - 1][jj - 1];
        }
        for (int k = 0; k < 7; k++) {
            double h = 0.0;
            for (int j = 0; j < 3; j++) h += z[k][j] * z[k][j];
            hdiag[k] = h;
        }
    }
    double dsq = 0.0;
    double bsum = 0.0, dx = 0.0;
    for (int j = 1; j <= 3; j++) {
        dsq += dd[j - 1] * dd[j - 1];
        double sum = 0.0;
        for (int k = 1; k <= 7; k++) sum += w[k - 1] * bm[k - 1][j - 1];
        bsum += sum * dd[j - 1];
        const int jp = 7 + j;
        for (int i = 1; i <= 3; i++) sum += bm[jp - 1][i - 1] * dd[i - 1];
        vl[jp - 1] = sum;
        bsum += sum * dd[j - 1];
        dx += dd[j - 1] * xo[j - 1];
    }
    beta = dx * dx + dsq * (xoptsq + dx + dx + 0.5 * dsq) + beta - bsum;
    for (int k = 1; k <= 7; k++) if (k == kopt) vl[k - 1] += 1.0;
    for (int i = 0; i < 10; i++) BQ_VLAG(i + 1) = vl[i];
    BQS(beta) = beta;
    BQS(dsq) = dsq;
    if (ntrits == 0) {
        double vk = 0.0;
        for (int k = 1; k <= 7; k++) if (k == knew0) vk = vl[k - 1];
        const double denom = vk * vk + alpha * beta;
        BQS(denom) = denom;
        if (denom < cauchy && cauchy > 0.0) {
            for (int i = 0; i < 3; i++) { const double xa = BQA(xalt, i); BQA(xnew, i) = xa; BQA(d, i) = xa - xo[i]; }
            BQS(cauchy) = 0.0;
            return L230;
        }
        if (denom <= 0.5 * (vk * vk)) {
            if (BQN(nevals) > BQN(nresc)) return L190;
            BQN(rc) = BQ_ROUNDOFF_LIMITED;
            return L720;
        }
    } else {
        double scaden, biglsq;
        pick_knew_core(B, hdiag, distsq, vl, beta, true, scaden, biglsq);
        if (scaden <= 0.5 * biglsq) {
            if (BQN(nevals) > BQN(nresc)) return L190;
            BQN(rc) = BQ_ROUNDOFF_LIMITED;
            return L720;
        }
    }
    return L360;
}

// ---------------------------------------------------------------- bobyqb_ main loop (bobyqa.c:2143-3055)
// Runs from `label` until the next objective evaluation is needed (returns BQ_CONTINUE with
// B.xeval set) or the run ends (BQ_DONE).
BQ_HD_BLOCK int step_label(BqRef B, int label) {
    {
        switch (label) {
        case L20: {  // model gradient at xopt when kopt moved away from kbase (bobyqa.c:2149-2182)
            if (BQN(kopt) != BQN(kbase)) {
                double go[3], xo[3], hq[6];
                const int nevals = BQN(nevals);
                for (int j = 0; j < 3; j++) { go[j] = BQ_V(gopt, j + 1); xo[j] = BQ_V(xopt, j + 1); }
                for (int i = 0; i < 6; i++) hq[i] = BQ_HQ(i + 1);
                int ih = 0;
                for (int j = 1; j <= 3; j++)
                    for (int i = 1; i <= j; i++) {
                        ++ih;
                        if (i < j) go[j - 1] += hq[ih - 1] * xo[i - 1];
                        go[i - 1] += hq[ih - 1] * xo[j - 1];
                    }
                if (nevals > 7) {
                    double pq[7], xp[7][3];
                    for (int k = 0; k < 7; k++) {
                        pq[k] = BQ_PQ(k + 1);
                        for (int j = 0; j < 3; j++) xp[k][j] = BQ_XPT(k + 1, j + 1);
                    }
                    for (int k = 1; k <= 7; k++) {
                        double temp = 0.0;
                        for (int j = 1; j <= 3; j++) temp += xp[k - 1][j - 1] * xo[j - 1];
                        temp = pq[k - 1] * temp;
                        for (int i = 1; i <= 3; i++) go[i - 1] += temp * xp[k - 1][i - 1];
                    }
                }
                for (int j = 0; j < 3; j++) BQ_V(gopt, j + 1) = go[j];
            }
            return L60;
        }
        case L60: {  // trust-region step (bobyqa.c:2190-2240)
            BQ_PROF_BEGIN(t_);
            trsbox(B);
            BQ_PROF_END(t_, 0);
            const double delta = BQS(delta), rho = BQS(rho);
            const double dnorm = mn(delta, sqrt(BQS(dsq)));
            BQS(dnorm) = dnorm;
            if (dnorm < 0.5 * rho) {
                const int nevals = BQN(nevals), nfsav = BQN(nfsav);
                const double diffa = BQS(diffa), diffb = BQS(diffb), diffc = BQS(diffc), crvmin = BQS(crvmin);
                BQN(ntrits) = -1;
                BQS(distsq) = (10.0 * rho) * (10.0 * rho);
                if (nevals <= nfsav + 2) { return L650; }
                const double errbig = mx(mx(diffa, diffb), diffc);
                const double frhosq = rho * .125 * rho;
                if (crvmin > 0.0 && errbig > frhosq * crvmin) { return L650; }
                const double bdtol = errbig / rho;
                bool to650 = false;
                for (int j = 1; j <= 3; j++) {
                    double bdtest = bdtol;
                    if (BQ_V(xnew, j) == BQ_V(sl, j)) bdtest = BQ_V(gnew, j);
                    if (BQ_V(xnew, j) == BQ_V(su, j)) bdtest = -BQ_V(gnew, j);
                    if (bdtest < bdtol) {
                        double curv = BQ_HQ((j + j * j) / 2);
                        for (int k = 1; k <= 7; k++) curv += BQ_PQ(k) * (BQ_XPT(k, j) * BQ_XPT(k, j));
                        bdtest += 0.5 * curv * rho;
                        if (bdtest < bdtol) { to650 = true; break; }
                    }
                }
                return to650 ? L650 : L680;
            }
            ++BQN(ntrits);
            return L90;
        }
        case L90: return label90(B);
        case L210: {  // geometry step (bobyqa.c:2376-2384)
            BQ_PROF_BEGIN(t_);
            altmov(B);
            BQ_PROF_END(t_, 1);
            for (int i = 0; i < 3; i++) BQA(d, i) = BQA(xnew, i) - BQA(xopt, i);
            return L230;
        }
        case L230: return label230(B);
        case L360: {  // next objective evaluation (bobyqa.c:2560-2583)
            if (BQN(maxeval) > 0 && BQN(nevals) >= BQN(maxeval)) { BQN(rc) = BQ_MAXEVAL_REACHED; return L720; }
            set_x_from(B, O_xnew);
            BQN(nevals)++;
            BQN(phase) = PH_MAIN;
            return LRET_CONT;
        }
        case L650: {  // is some interpolation point too far from xopt? (bobyqa.c:2972-3013)
            double xp[7][3], xo[3];
            for (int k = 0; k < 7; k++)
                for (int j = 0; j < 3; j++) xp[k][j] = BQ_XPT(k + 1, j + 1);
            for (int j = 0; j < 3; j++) xo[j] = BQ_V(xopt, j + 1);
            double distsq = BQS(distsq);
            const int ntrits = BQN(ntrits);
            const double delta0 = BQS(delta), rho = BQS(rho), ratio = BQS(ratio), dnorm = BQS(dnorm);
            int knew = 0;
            for (int k = 1; k <= 7; k++) {
                double sum = 0.0;
                for (int j = 1; j <= 3; j++) {
                    const double t = xp[k - 1][j - 1] - xo[j - 1];
                    sum += t * t;
                }
                if (sum > distsq) { knew = k; distsq = sum; }
            }
            BQN(knew) = knew;
            if (knew > 0) {
                BQS(distsq) = distsq;
                const double dist = sqrt(distsq);
                double delta = delta0;
                if (ntrits == -1) {
                    delta = mn(0.1 * delta, 0.5 * dist);
                    if (delta <= rho * 1.5) delta = rho;
                    BQS(delta) = delta;
                }
                BQN(ntrits) = 0;
                const double adelt = mx(mn(0.1 * dist, delta), rho);
                BQS(adelt) = adelt;
                BQS(dsq) = adelt * adelt;
                return L90;
            }
            if (ntrits == -1) { return L680; }
            if (ratio > 0.0) { return L60; }
            if (mx(delta0, dnorm) > rho) { return L60; }
            return L680;
        }
        case L680: {  // reduce rho, or finish (bobyqa.c:3019-3041)
            if (BQS(rho) > BQS(rhoend)) {
                BQS(delta) = 0.5 * BQS(rho);
                BQS(ratio) = BQS(rho) / BQS(rhoend);
                if (BQS(ratio) <= 16.) BQS(rho) = BQS(rhoend);
                else if (BQS(ratio) <= 250.) BQS(rho) = sqrt(BQS(ratio)) * BQS(rhoend);
                else BQS(rho) = 0.1 * BQS(rho);
                BQS(delta) = mx(BQS(delta), BQS(rho));
                BQN(ntrits) = 0;
                BQN(nfsav) = BQN(nevals);
                return L60;
            }
            if (BQN(ntrits) == -1) { return L360; }
            return L720;
        }
        case L190:  // rescue_ (bobyqa.c:2331-2365)
            return rescue_begin(B);
        case L720:
        default: {  // return the best point (bobyqa.c:3042-3054)
            for (int i = 1; i <= 3; i++) {
                double v = mn(mx(BQ_V(xl, i), BQ_V(xbase, i) + BQ_V(xopt, i)), BQ_V(xu, i));
                if (BQ_V(xopt, i) == BQ_V(sl, i)) v = BQ_V(xl, i);
                if (BQ_V(xopt, i) == BQ_V(su, i)) v = BQ_V(xu, i);
                BQ_V(xfinal, i) = v * BQ_V(sc, i);
            }
            BQS(minf) = BQ_FVAL(BQN(kopt));
            BQN(phase) = PH_DONE;
            return LRET_DONE;
        }
        }
    }
}

// Drive the labels until the next evaluation (BQ_CONTINUE) or the end of the run (BQ_DONE).
// On the GPU the lanes of a wavefront that are inside this function run DIFFERENT patches; a plain
// per-lane loop lets them drift out of phase, and SIMT execution then repeats the expensive blocks
// (trsbox, altmov) once per straggler.  The wave-aware scheduler below executes ONE label per
// iteration for all lanes waiting on it, cheap transition labels first and the expensive ones only
// when nobody can advance otherwise, so each expensive block runs ~once per round for all lanes.
BQ_HD int run(BqRef B, int label) {
#if defined(__HIP_DEVICE_COMPILE__)
    for (;;) {
        const bool live = (label != LRET_CONT && label != LRET_DONE);
        if (!__any(live)) break;
        // Priority = a topological order of the label graph (20 -> 60; 650 -> 680 | 60 | 90; 680 -> 60 | 360;
        // 60 -> 650 | 90; 90 -> 210 | 230; 210 -> 230; 230 -> 360): a label runs only when no lane can still
        // arrive at it from an earlier one, so each block executes about once per round for all lanes.
        int pick = LRET_DONE;
        if (__any(live && label == L20)) pick = L20;
        else if (__any(live && label == L650)) pick = L650;
        else if (__any(live && label == L680)) pick = L680;
        else if (__any(live && label == L720)) pick = L720;
        else if (__any(live && label == L190)) pick = L190;
        else if (__any(live && label == L60)) pick = L60;
        else if (__any(live && label == L90)) pick = L90;
        else if (__any(live && label == L210)) pick = L210;
        else if (__any(live && label == L230)) pick = L230;
        else if (__any(live && label == L360)) pick = L360;
        if (live && label == pick) {
            BQ_PROF_BEGIN(tl_);
            const int was = label;
            label = step_label(B, label);
            (void)was;
            BQ_PROF_END(tl_, 4 + was);  // per-label timers (profiling builds only)
        }
    }
#else
    while (label != LRET_CONT && label != LRET_DONE) label = step_label(B, label);
#endif
    return label == LRET_CONT ? BQ_CONTINUE : BQ_DONE;
}


// The same scheduler restricted to a SET of labels (bit (1 << label) of `allowed`): runs until no lane waits on
// an allowed label.  The stream kernel (kernel_stream.hip) executes the expensive blocks (L60 trsbox, L90 shift,
// L210 altmov, L230, L190 rescue) only for wavefronts whose lanes ALL wait on that block (patches queue up per
// block in global memory); the cheap transition labels run inline wherever a patch happens to be.
BQ_HD int run_set(BqRef B, int label, unsigned allowed) {
#if defined(__HIP_DEVICE_COMPILE__)
    for (;;) {
        const bool live = (label != LRET_CONT && label != LRET_DONE) && ((allowed >> label) & 1u);
        if (!__any(live)) break;
        int pick = LRET_DONE;
        if (__any(live && label == L20)) pick = L20;
        else if (__any(live && label == L650)) pick = L650;
        else if (__any(live && label == L680)) pick = L680;
        else if (__any(live && label == L720)) pick = L720;
        else if (__any(live && label == L190)) pick = L190;
        else if (__any(live && label == L60)) pick = L60;
        else if (__any(live && label == L90)) pick = L90;
        else if (__any(live && label == L210)) pick = L210;
        else if (__any(live && label == L230)) pick = L230;
        else if (__any(live && label == L360)) pick = L360;
        if (live && label == pick) label = step_label(B, label);
    }
#else
    while (label != LRET_CONT && label != LRET_DONE && ((allowed >> label) & 1u)) label = step_label(B, label);
#endif
    return label;
}

// L90 (bobyqa.c:2246-2322) does nothing unless the step is small relative to |xopt|: the cheap outcome
BQ_HD bool l90_is_noop(BqRef B) { return !(BQS(dsq) <= BQS(xoptsq) * .001); }
BQ_HD int l90_next(BqRef B) { return (BQN(ntrits) == 0) ? L210 : L230; }

// Code after an evaluation at xbase + xnew in the main loop (bobyqa.c:2583-2970)
// returns the label to continue at (or LRET_DONE)
BQ_HD_BLOCK int after_eval(BqRef B, double f) {
    double w[10];
    if (BQN(ntrits) == -1) {
        BQS(fsave) = f;
        BQN(rc) = BQ_XTOL_REACHED;
        if (BQS(fsave) < BQ_FVAL(BQN(kopt))) {
            BQS(minf) = f;
            // xfinal = x * sc of the point just evaluated: that is what the shared xeval / xfinal slot holds
            BQN(phase) = PH_DONE;
            return LRET_DONE;
        }
        return L720;
    }
    const int ntrits = BQN(ntrits);
    const double fopt = BQ_FVAL(BQN(kopt));
    double vquad = 0.0;
    {
        double dd[3], go[3], hq[6], pq[7], ws[7];
        for (int j = 0; j < 3; j++) { dd[j] = BQ_V(d, j + 1); go[j] = BQ_V(gopt, j + 1); }
        for (int i = 0; i < 6; i++) hq[i] = BQ_HQ(i + 1);
        for (int k = 0; k < 7; k++) { pq[k] = BQ_PQ(k + 1); ws[k] = BQ_WS(k); }
        int ih = 0;
        for (int j = 1; j <= 3; j++) {
            vquad += dd[j - 1] * go[j - 1];
            for (int i = 1; i <= j; i++) {
                ++ih;
                double temp = dd[i - 1] * dd[j - 1];
                if (i == j) temp = 0.5 * temp;
                vquad += hq[ih - 1] * temp;
            }
        }
        for (int k = 1; k <= 7; k++) vquad += 0.5 * pq[k - 1] * (ws[k - 1] * ws[k - 1]);
    }
    const double diff = f - fopt - vquad;
    {
        const double diffa = BQS(diffa), diffb = BQS(diffb);
        BQS(diffc) = diffb;
        BQS(diffb) = diffa;
        BQS(diffa) = fabs(diff);
    }
    const double dnorm = BQS(dnorm), rho = BQS(rho);
    if (dnorm > rho) BQN(nfsav) = BQN(nevals);

    if (ntrits > 0) {
        if (vquad >= 0.0) { BQN(rc) = BQ_ROUNDOFF_LIMITED; return L720; }
        const double ratio = (f - fopt) / vquad;
        BQS(ratio) = ratio;
        double delta = BQS(delta);
        if (ratio <= 0.1) delta = mn(0.5 * delta, dnorm);
        else if (ratio <= .7) delta = mx(0.5 * delta, dnorm);
        else delta = mx(0.5 * delta, dnorm + dnorm);
        if (delta <= rho * 1.5) delta = rho;
        BQS(delta) = delta;
        if (f < fopt) {
            const int ksav = BQN(knew);
            const double densav = BQS(denom);
            double scaden, biglsq;
            pick_knew(B, O_xnew, false, scaden, biglsq);
            if (scaden <= 0.5 * biglsq) { BQN(knew) = ksav; BQS(denom) = densav; }
        }
    }
    { BQ_PROF_BEGIN(t_); update(B); BQ_PROF_END(t_, 2); }
    const int knew = BQN(knew);
    double z[7][3], xp[7][3], xn[3], xo[3], go[3];
    {   // model update for the replaced point: hq with the OLD xpt(knew), pq through zmat (bobyqa.c:2720-2760)
        double pq[7], hq[6], zk[3], xk[3];
        for (int k = 0; k < 7; k++) {
            pq[k] = BQ_PQ(k + 1);
            for (int j = 0; j < 3; j++) { z[k][j] = BQ_ZMAT(k + 1, j + 1); xp[k][j] = BQ_XPT(k + 1, j + 1); }
        }
        for (int i = 0; i < 6; i++) hq[i] = BQ_HQ(i + 1);
        for (int j = 0; j < 3; j++) { zk[j] = BQ_ZMAT(knew, j + 1); xk[j] = BQ_XPT(knew, j + 1); w[j] = BQ_BMAT(knew, j + 1); }
        for (int j = 0; j < 3; j++) { xn[j] = BQ_V(xnew, j + 1); xo[j] = BQ_V(xopt, j + 1); go[j] = BQ_V(gopt, j + 1); }
        const double pqold = BQ_PQ(knew);
        for (int k = 1; k <= 7; k++) if (k == knew) pq[k - 1] = 0.0;
        int ih = 0;
        for (int i = 1; i <= 3; i++) {
            const double temp = pqold * xk[i - 1];
            for (int j = 1; j <= i; j++) { ++ih; hq[ih - 1] += temp * xk[j - 1]; }
        }
        for (int jj = 1; jj <= 3; jj++) {
            const double temp = diff * zk[jj - 1];
            for (int k = 1; k <= 7; k++) pq[k - 1] += temp * z[k - 1][jj - 1];
        }
        for (int k = 0; k < 7; k++) BQ_PQ(k + 1) = pq[k];
        for (int i = 0; i < 6; i++) BQ_HQ(i + 1) = hq[i];
        BQ_FVAL(knew) = f;
        for (int i = 1; i <= 3; i++) {
            BQ_XPT(knew, i) = xn[i - 1];
            for (int k = 1; k <= 7; k++) if (k == knew) xp[k - 1][i - 1] = xn[i - 1];
        }
        for (int k = 1; k <= 7; k++) {
            double suma = 0.0;
            for (int jj = 1; jj <= 3; jj++) suma += zk[jj - 1] * z[k - 1][jj - 1];
            if (is_inf(suma)) { BQN(rc) = BQ_ROUNDOFF_LIMITED; return L720; }
            double sumb = 0.0;
            for (int j = 1; j <= 3; j++) sumb += xp[k - 1][j - 1] * xo[j - 1];
            const double temp = suma * sumb;
            for (int i = 1; i <= 3; i++) w[i - 1] += temp * xp[k - 1][i - 1];
        }
        for (int i = 1; i <= 3; i++) go[i - 1] += diff * w[i - 1];

        if (f < fopt) {
            double dd[3];
            for (int j = 0; j < 3; j++) dd[j] = BQ_V(d, j + 1);
            BQN(kopt) = knew;
            double xoptsq = 0.0;
            ih = 0;
            for (int j = 1; j <= 3; j++) {
                xo[j - 1] = xn[j - 1];
                xoptsq += xo[j - 1] * xo[j - 1];
                for (int i = 1; i <= j; i++) {
                    ++ih;
                    if (i < j) go[j - 1] += hq[ih - 1] * dd[i - 1];
                    go[i - 1] += hq[ih - 1] * dd[j - 1];
                }
            }
            for (int k = 1; k <= 7; k++) {
                double temp = 0.0;
                for (int j = 1; j <= 3; j++) temp += xp[k - 1][j - 1] * dd[j - 1];
                temp = pq[k - 1] * temp;
                for (int i = 1; i <= 3; i++) go[i - 1] += temp * xp[k - 1][i - 1];
            }
            for (int j = 0; j < 3; j++) BQ_V(xopt, j + 1) = xo[j];
            BQS(xoptsq) = xoptsq;
            // nlopt_stop_ftol with ftol_rel = ftol_abs = 0 (PatchOptimizer.cpp sets only xtol_rel) never fires
        }
        for (int j = 0; j < 3; j++) BQ_V(gopt, j + 1) = go[j];
    }

    if (ntrits > 0) {  // least-Frobenius-norm interpolant test (bobyqa.c:2845-2944)
        double vl[10], wk[14], fv[7], bmr[7][3], sl_[3], su_[3];
        for (int k = 0; k < 7; k++) {
            fv[k] = BQ_FVAL(k + 1);
            for (int j = 0; j < 3; j++) bmr[k][j] = BQ_BMAT(k + 1, j + 1);
        }
        for (int j = 0; j < 3; j++) { sl_[j] = BQ_V(sl, j + 1); su_[j] = BQ_V(su, j + 1); }
        const double fko = BQ_FVAL(BQN(kopt));
        for (int k = 1; k <= 7; k++) { vl[k - 1] = fv[k - 1] - fko; wk[k - 1] = 0.0; }
        for (int j = 1; j <= 3; j++) {
            double sum = 0.0;
            for (int k = 1; k <= 7; k++) sum += z[k - 1][j - 1] * vl[k - 1];
            for (int k = 1; k <= 7; k++) wk[k - 1] += sum * z[k - 1][j - 1];
        }
        for (int k = 1; k <= 7; k++) {
            double sum = 0.0;
            for (int j = 1; j <= 3; j++) sum += xp[k - 1][j - 1] * xo[j - 1];
            wk[k + 7 - 1] = wk[k - 1];
            wk[k - 1] = sum * wk[k - 1];
        }
        double gqsq = 0.0, gisq = 0.0;
        for (int i = 1; i <= 3; i++) {
            double sum = 0.0;
            for (int k = 1; k <= 7; k++) sum = sum + bmr[k - 1][i - 1] * vl[k - 1] + xp[k - 1][i - 1] * wk[k - 1];
            if (xo[i - 1] == sl_[i - 1]) {
                const double t1 = mn(0.0, go[i - 1]);
                gqsq += t1 * t1;
                const double t2 = mn(0.0, sum);
                gisq += t2 * t2;
            } else if (xo[i - 1] == su_[i - 1]) {
                const double t1 = mx(0.0, go[i - 1]);
                gqsq += t1 * t1;
                const double t2 = mx(0.0, sum);
                gisq += t2 * t2;
            } else {
                gqsq += go[i - 1] * go[i - 1];
                gisq += sum * sum;
            }
            vl[7 + i - 1] = sum;
        }
        int itest = BQN(itest) + 1;
        if (gqsq < 10.0 * gisq) itest = 0;
        if (itest >= 3) {
            for (int i = 1; i <= 7; i++) {  // max(npt, nh) = 7
                if (i <= 3) BQ_V(gopt, i) = vl[7 + i - 1];
                if (i <= 7) BQ_PQ(i) = wk[7 + i - 1];
                if (i <= 6) BQ_HQ(i) = 0.0;
            }
            itest = 0;
        }
        BQN(itest) = itest;
    }
    if (ntrits == 0) return L60;
    if (f <= fopt + 0.1 * vquad) return L60;
    const double a = 2.0 * BQS(delta), b = 10.0 * rho;
    BQS(distsq) = mx(a * a, b * b);
    return L650;
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
// first point in B.xeval, or BQ_DONE with BQN(rc) = BQ_INVALID_ARGS.
BQ_HD void bq_prelim_points(BqRef B, double (*out)[3]);
// prelim_out (optional, 18 values): the start points 2..7 of prelim_, see bq_prelim_points
BQ_HD_NOINLINE int bq_start(BqRef B_in, const double* x0, const double* lb, const double* ub, const double* dx,
                   double xtol_rel, int maxeval, double* prelim_out = nullptr) {
    BqRef B = BQ_UNIFORM(B_in);
    BQN(nevals) = 0; BQN(rc) = BQ_SUCCESS; BQS(minf) = HUGE_VAL; BQN(maxeval) = maxeval; BQN(phase) = bq::PH_PRELIM;
    for (int i = 0; i < 3; i++) { BQA(xfinal, i) = x0[i]; BQA(xeval, i) = x0[i]; }
    for (int i = 0; i < 3; i++)
        if (lb[i] > ub[i] || x0[i] < lb[i] || x0[i] > ub[i]) { BQN(rc) = BQ_INVALID_ARGS; BQN(phase) = bq::PH_DONE; return BQ_DONE; }
    // nlopt_compute_rescaling (util/rescale.c:29-44)
    for (int i = 0; i < 3; i++) BQA(sc, i) = 1.0;
    if (!(dx[1] == dx[0] && dx[2] == dx[1]))
        for (int i = 1; i < 3; i++) BQA(sc, i) = dx[i] / dx[0];
    double xs[3];  // the scaled starting point (bobyqa.c's X)
    for (int i = 0; i < 3; i++) {
        xs[i] = x0[i] / BQA(sc, i);
        BQA(xl, i) = lb[i] / BQA(sc, i);
        BQA(xu, i) = ub[i] / BQA(sc, i);
        if (BQA(xl, i) > BQA(xu, i)) { const double t = BQA(xl, i); BQA(xl, i) = BQA(xu, i); BQA(xu, i) = t; }
    }
    BQS(rhobeg) = fabs(dx[0] / BQA(sc, 0));
    BQS(rhoend) = xtol_rel * BQS(rhobeg);  // xtol_abs = 0
    for (int j = 0; j < 3; j++) {
        const double temp = BQA(xu, j) - BQA(xl, j);
        if (temp < BQS(rhobeg) + BQS(rhobeg)) { BQN(rc) = BQ_INVALID_ARGS; BQN(phase) = bq::PH_DONE; return BQ_DONE; }
        BQA(sl, j) = BQA(xl, j) - xs[j];
        BQA(su, j) = BQA(xu, j) - xs[j];
        if (BQA(sl, j) >= -BQS(rhobeg)) {
            if (BQA(sl, j) >= 0.0) { xs[j] = BQA(xl, j); BQA(sl, j) = 0.0; BQA(su, j) = temp; }
            else { xs[j] = BQA(xl, j) + BQS(rhobeg); BQA(sl, j) = -BQS(rhobeg); BQA(su, j) = bq::mx(BQA(xu, j) - xs[j], BQS(rhobeg)); }
        } else if (BQA(su, j) <= BQS(rhobeg)) {
            if (BQA(su, j) <= 0.0) { xs[j] = BQA(xu, j); BQA(sl, j) = -temp; BQA(su, j) = 0.0; }
            else { xs[j] = BQA(xu, j) - BQS(rhobeg); BQA(sl, j) = bq::mn(BQA(xl, j) - xs[j], -BQS(rhobeg)); BQA(su, j) = BQS(rhobeg); }
        }
    }
    // prelim_ initialisation (bobyqa.c:1790-1822)
    for (int j = 0; j < 3; j++) { BQA(xbase, j) = xs[j]; BQA(xopt, j) = 0; BQA(gopt, j) = 0; BQA(xnew, j) = 0; BQA(xalt, j) = 0; BQA(d, j) = 0; BQA(gnew, j) = 0; }
    for (int k = 0; k < 21; k++) { BQ_MA(xpt, k) = 0.0; BQ_COLDA(zmat, k) = 0.0; }
    for (int k = 0; k < 30; k++) BQ_COLDA(bmat, k) = 0.0;
    for (int k = 0; k < 6; k++) BQ_MA(hq, k) = 0.0;
    for (int k = 0; k < 7; k++) { BQ_MA(pq, k) = 0.0; BQ_MA(fval, k) = 0.0; BQ_WS(k) = 0.0; }
    for (int k = 0; k < 10; k++) BQ_COLDA(vlag, k) = 0.0;
    BQS(rho) = BQS(delta) = BQS(diffa) = BQS(diffb) = BQS(diffc) = BQS(dsq) = BQS(crvmin) = BQS(dnorm) = BQS(xoptsq) = BQS(ratio) = BQS(adelt) = 0.0;
    BQS(alpha) = BQS(cauchy) = BQS(beta) = BQS(denom) = BQS(fsave) = BQS(distsq) = BQS(fbeg) = BQS(stepa) = BQS(stepb) = 0.0;
    BQN(kopt) = 1; BQN(knew) = 0; BQN(kbase) = 1; BQN(ntrits) = 0; BQN(itest) = 0; BQN(nresc) = 0; BQN(nfsav) = 0; BQN(nf) = 0;
    bq::prelim_next_point(B);
    BQN(nevals)++;
    if (prelim_out) {
        double pts[6][3];
        bq_prelim_points(B, pts);
        for (int k = 0; k < 6; k++) for (int i = 0; i < 3; i++) prelim_out[3 * k + i] = pts[k][i];
    }
    return BQ_CONTINUE;
}

// The evaluation points 2 .. 7 of prelim_ (bobyqa.c:1822-1883), all at once.  They are functions of the start point, the
// bounds and rhobeg only -- point nf is xbase + (one coordinate step), the step of the second family (nf >= 5) reads the
// first family's step of the same coordinate, and the exchange prelim_ makes after an evaluation (bobyqa.c:1916-1926)
// touches one column of XPT that no later point reads -- so a caller may evaluate them side by side and feed the values
// to bq_step in order (fewer sequential rounds per patch).  Pure: the state is not changed; out[k] is
// bit for bit the xeval that bq_step leaves after consuming point k + 1 (tests/test_cpu_oracle.py pins that on the host
// build).  Valid after a bq_start that returned BQ_CONTINUE.
BQ_HD void bq_prelim_points(BqRef B, double (*out)[3]) {
    const double rhobeg = BQS(rhobeg);
    for (int nf = 2; nf <= 7; nf++) {
        double row[3] = {0.0, 0.0, 0.0};
        if (nf <= 4) {
            const int nfm = nf - 1;
            double stepa = rhobeg;
            if (BQ_V(su, nfm) == 0.0) stepa = -stepa;
            row[nfm - 1] = stepa;
        } else {
            const int nfx = nf - 4;
            double stepb = -rhobeg;
            if (BQ_V(sl, nfx) == 0.0) stepb = bq::mn(2.0 * rhobeg, BQ_V(su, nfx));
            if (BQ_V(su, nfx) == 0.0) stepb = bq::mx(-2.0 * rhobeg, BQ_V(sl, nfx));
            row[nfx - 1] = stepb;
        }
        for (int i = 1; i <= 3; i++) {  // set_x_from on that row
            const double pi = row[i - 1];
            double v = bq::mn(bq::mx(BQ_V(xl, i), BQ_V(xbase, i) + pi), BQ_V(xu, i));
            if (pi == BQ_V(sl, i)) v = BQ_V(xl, i);
            if (pi == BQ_V(su, i)) v = BQ_V(xu, i);
            out[nf - 2][i - 1] = v * BQ_V(sc, i);
        }
    }
}

BQ_HD_NOINLINE int bq_step_(BqRef B, double f);
BQ_HD int bq_step(BqRef B, double f) {
    BQ_PROF_BEGIN(t_);
    const int r = bq_step_(B, f);
    BQ_PROF_END(t_, 3);
    return r;
}
// everything bq_step does before the label loop: consumes f, returns the label to continue at
BQ_HD_BLOCK int bq_step_entry(BqRef B, double f);
BQ_HD_NOINLINE int bq_step_(BqRef B_in, double f) {
    BqRef B = BQ_UNIFORM(B_in);
    return bq::run(B, bq_step_entry(B, f));
}
BQ_HD_BLOCK int bq_step_entry(BqRef B, double f) {
    int label = bq::LRET_DONE;
    if (BQN(phase) == bq::PH_PRELIM) {
        bq::prelim_consume(B, f);
        // stopping tests after each prelim evaluation (bobyqa.c:1944-1947)
        const bool maxed = (BQN(maxeval) > 0 && BQN(nevals) >= BQN(maxeval));
        if (!maxed && BQN(nf) < 7) {
            bq::prelim_next_point(B);
            BQN(nevals)++;
            label = bq::LRET_CONT;
        } else {
            // bobyqb_ set-up after prelim_ (bobyqa.c:2117-2143)
            BQS(xoptsq) = 0.0;
            for (int i = 1; i <= 3; i++) {
                BQ_V(xopt, i) = BQ_XPT(BQN(kopt), i);
                BQS(xoptsq) += BQ_V(xopt, i) * BQ_V(xopt, i);
            }
            BQS(fsave) = BQ_FVAL(1);
            if (maxed) { BQN(rc) = BQ_MAXEVAL_REACHED; label = bq::L720; }
            else {
                BQN(kbase) = 1;
                BQS(rho) = BQS(rhobeg);
                BQS(delta) = BQS(rho);
                BQN(nresc) = BQN(nevals);
                BQN(ntrits) = 0;
                BQS(diffa) = 0.0; BQS(diffb) = 0.0; BQS(diffc) = 0.0;
                BQN(itest) = 0;
                BQN(nfsav) = BQN(nevals);
                label = bq::L20;
            }
        }
    } else if (BQN(phase) == bq::PH_MAIN) {
        { BQ_PROF_BEGIN(tae_); label = bq::after_eval(B, f); BQ_PROF_END(tae_, 23); }
    } else if (BQN(phase) == bq::PH_RESCUE) {
        label = bq::rescue_consume(B, f);
    }
    return label;
}

}  // namespace hpmvs
