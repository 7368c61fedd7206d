"""Problem set of the BOBYQA device self-test (hpmvs_selftest_bobyqa) + the host reference run.

The objectives are those of hpmvs_amd/csrc/selftest_obj.h.  `reference_run` minimises them with the reference's
GENUINE NLopt 2.4.2 BOBYQA (oracle/_ref) through the C callback of tests/native/libselftest_ref.so and returns
(result, x, minf, nevals, trace) per problem; `host_state_machine_run` does the same with the host build of the
product's bobyqa3.hpp and also reports how often rescue_ was entered."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
INF = float("inf")
LB = [-INF, -23.99999, -23.99999]   # reference src/hpmvs/PatchOptimizer.cpp:326-336
UB = [INF, 23.99999, 23.99999]
TRACE_CAP = 1000


class Problem(C.Structure):
    _fields_ = [("kind", C.c_int), ("n_trace", C.c_int), ("trace_cap", C.c_int), ("p", C.c_double * 8),
                ("trace", C.POINTER(C.c_double))]


def _libs():
    d = os.path.join(HERE, "native")
    subprocess.run(["make", "-C", d, "-s", "libselftest_ref.so", "libbq_host.so"], check=True, capture_output=True)
    S = C.CDLL(os.path.join(d, "libselftest_ref.so"))
    S.selftest_callback_address.restype = C.c_void_p
    H = C.CDLL(os.path.join(d, "libbq_host.so"))
    H.bq3_rescue_calls.restype = C.c_int
    return S, H


def problems():
    """(kind, params[8], x0[3], lb[3], ub[3]) -- deterministic."""
    out = []
    P = lambda *a: list(a) + [0.0] * (8 - len(a))
    # rescue_ cases found by tools/find_rescue_cases.py (genuine NLopt enters rescue_ on each)
    for kind, p, x0 in RESCUE:
        out.append((kind, P(*p), x0, LB, UB))
    # Rosenbrock chains, different conditioning
    for k, c in enumerate((1.0, 10.0, 100.0, 1000.0)):
        out.append((1, P(0, 0, 0, c), [-1.2 + 0.1 * k, 1.0, 0.5], LB, UB))
    # the plateau: every evaluation returns 2.0
    out.append((2, P(), [0.0, 0.0, 0.0], LB, UB))
    out.append((2, P(), [0.0, 23.0, -23.99999], LB, UB))
    # minimum outside the angle bounds: active bounds / ROUNDOFF_LIMITED
    out.append((5, P(1.0, 30.0, -30.0), [0.0, 0.0, 0.0], LB, UB))
    out.append((5, P(-3.0, 23.99999, 40.0), [0.0, 23.0, 23.0], LB, UB))
    out.append((5, P(0.5, -100.0, 5.0), [0.0, -23.99999, 0.0], LB, UB))
    # non-smooth valleys, bound-active starts
    out.append((4, P(0.37, 3.0, -5.0, 0.3, 0.01, 0.02, 0.05), [0.0, 0.1, -0.2], LB, UB))
    out.append((4, P(0.37, 3.0, -5.0, 0.3, 0.01, 0.02, 0.05), [0.0, 23.0, -23.99999], LB, UB))
    out.append((4, P(-2.0, 40.0, -40.0, 1.0, 0.5, 0.25, 0.0), [0.0, 10.0, -10.0], LB, UB))
    # a finite box in all three variables (start on and near the bounds)
    out.append((1, P(0, 0, 0, 100.0), [0.9, 0.9, 0.9], [0.9, 0.0, 0.9], [1.2, 2.0, 1.2]))
    out.append((0, P(1.0, 10.0, 1.0, 3.0, 0.5, 7.0, 0.0), [0.95, 9.5, 1.1], [0.9, 9.0, 0.9], [1.2, 11.2, 1.2]))
    # random well- and ill-conditioned quadratics (some with sawtooth noise), random starts
    rng = np.random.default_rng(0x5E1F)
    for _ in range(64 - len(out)):
        a = rng.uniform(-3, 3, 3)
        s = 10.0 ** rng.uniform(-4, 4, 3)
        noise = 0.0 if rng.random() < 0.5 else 10.0 ** rng.uniform(-14, -6)
        x0 = [0.0, float(rng.uniform(-20, 20)), float(rng.uniform(-20, 20))]
        out.append((0, P(*a, *s, noise), x0, LB, UB))
    return out


def _run(fn_ptr, minimizer, probs, trace_cap):
    res = []
    for kind, p, x0, lb, ub in probs:
        pr = Problem()
        pr.kind = kind; pr.n_trace = 0; pr.trace_cap = trace_cap
        for i in range(8):
            pr.p[i] = p[i]
        tr = np.zeros((trace_cap, 4))
        pr.trace = tr.ctypes.data_as(C.POINTER(C.c_double))
        x = (C.c_double * 3)(*x0); l = (C.c_double * 3)(*lb); u = (C.c_double * 3)(*ub)
        mf, ne = C.c_double(), C.c_int()
        rc = minimizer(fn_ptr, C.byref(pr), x, l, u, C.byref(mf), C.byref(ne))
        res.append((rc, list(x), mf.value, ne.value, tr[:min(pr.n_trace, trace_cap)].copy()))
    return res


def reference_run(probs, trace_cap=TRACE_CAP):
    """genuine NLopt 2.4.2 BOBYQA (oracle/_ref)"""
    from oracle import oracle as orc
    assert orc.optimizer_available(orc.OPT_REF)
    S, _ = _libs()
    L = orc.lib()
    fn = C.cast(S.selftest_callback_address(), orc.OBJ_FN)
    mini = lambda f, data, x, l, u, mf, ne: L.orc_bobyqa_minimize(orc.OPT_REF, 3, f, data, x, l, u, 1e-7, 1000, mf, ne)
    return _run(fn, mini, probs, trace_cap)


def host_state_machine_run(probs, trace_cap=TRACE_CAP):
    """host build of the product's bobyqa3.hpp; also returns the number of rescue_ entries per problem"""
    from oracle import oracle as orc
    S, H = _libs()
    fn = C.cast(S.selftest_callback_address(), orc.OBJ_FN)
    H.bq3_minimize.argtypes = [orc.OBJ_FN, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                               C.POINTER(C.c_double), C.c_double, C.c_int, C.POINTER(C.c_double),
                               C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_int]
    out, resc = [], []
    for pr in probs:
        before = H.bq3_rescue_calls()
        mini = lambda f, data, x, l, u, mf, ne: H.bq3_minimize(f, data, x, l, u, 1e-7, 1000, mf, ne, None, 0)
        out += _run(fn, mini, [pr], trace_cap)
        resc.append(H.bq3_rescue_calls() - before)
    return out, resc


# found by tools/find_rescue_cases.py (seed 1): BOBYQA enters rescue_ 1-3 times on each; results include
# ROUNDOFF_LIMITED (-4), MAXEVAL_REACHED (5), SUCCESS (1) and XTOL_REACHED (4)
RESCUE = [
    (3, [1.358841661874223, -0.8099564189486461, 0.0, 37.48840713389653], [0.0, -15.61061343973213, -11.870338236504136]),
    (3, [-1.2299613431668832, 1.612630559454999, 0.0, 1.1049335575870118], [0.0, 18.598709759189433, -3.934551044459301]),
    (0, [2.9501171760867884, 1.3585478353667995, 2.2082285737751928, 3.9246905140154245e-06, 150.4047083414359, 0.19021193045425375, 2.590951784647335e-11], [0.0, 8.330191769918866, -7.667241978576778]),
    (0, [-2.3350753587221442, -0.3826084962035443, 1.8421629614823267, 0.0006542405397346704, 15660.224251950362, 347.4115187815097, 4.0950260313575645e-14], [0.0, 5.270061565571822, 12.787687311740328]),
    (3, [-0.8509231257930319, -0.2517362760128492, 0.0, 62.19340910120522], [0.0, -1.8988522144270057, 4.092276476378874]),
    (3, [-0.8630828384575917, 0.9684335534164932, 0.0, 0.3759014153159182], [0.0, 3.433363698742795, -18.13501310443246]),
    (3, [-1.6751151439523448, -2.5935929727495255, 0.0, 0.6616882930630047], [0.0, -8.929472166692328, -19.32121429844485]),
    (3, [-1.409866801550176, -1.199635949346478, 0.0, 194.34733021985792], [0.0, -18.072446688504606, -15.130880430188789]),
    (3, [2.1895845068066837, -1.4198316375906541, 0.0, 0.17854728392349598], [0.0, -19.62907669071665, -5.330163669212347]),
    (0, [0.7482566214923487, 1.9490185319397275, 0.7010156146436399, 903144.22330889, 684196.360372782, 6.254467103602167e-05, 1.1554026456811634e-10], [0.0, -1.2249784164662145, -12.534322445447707]),
]
