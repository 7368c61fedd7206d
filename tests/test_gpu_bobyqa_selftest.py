"""The product's BOBYQA state machine ON THE GPU (hpmvs_selftest_bobyqa: bobyqa3.hpp exactly as optimize_kernel runs
it, 32 different problems per wavefront, divergent labels) against the reference's GENUINE NLopt 2.4.2 BOBYQA on
analytic objectives whose arithmetic is identical on both sides (hpmvs_amd/csrc/selftest_obj.h).  Covers what the
photometric objective never reaches on a GPU: rescue_ (bobyqa.c:143-742, entered 1-3 times by 11 of the 64
problems), active bounds, the all-plateau objective, ROUNDOFF_LIMITED and MAXEVAL_REACHED.

Tolerance: none.  Every evaluation point and value, the result code, x*, f* and the evaluation count must be
IDENTICAL (float64 bit patterns), against the library when oracle/_ref travelled and against the committed goldens
(tests/golden/g5_bobyqa_selftest.json, produced by the genuine library in the build container) always."""
import hashlib
import json
import os

import numpy as np
import pytest

import selftest_problems as sp

pytestmark = pytest.mark.gpu


def _device_run(probs, trace_cap=sp.TRACE_CAP):
    import ctypes as C
    from hpmvs_amd import api
    L = api.lib()
    n = len(probs)
    kind = np.array([p[0] for p in probs], np.int32)
    par = np.array([p[1] for p in probs], np.float64)
    x0 = np.array([p[2] for p in probs], np.float64)
    lb = np.array([p[3] for p in probs], np.float64)
    ub = np.array([p[4] for p in probs], np.float64)
    xf = np.zeros((n, 3)); mf = np.zeros(n); rc = np.zeros(n, np.int32); ne = np.zeros(n, np.int32); rs = np.zeros(n, np.int32)
    tr = np.zeros((n, trace_cap, 4))
    L.hpmvs_selftest_bobyqa.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 6 + [C.c_int]
    st = L.hpmvs_selftest_bobyqa(0, n, kind.ctypes.data, par.ctypes.data, x0.ctypes.data, lb.ctypes.data, ub.ctypes.data,
                                 1000, xf.ctypes.data, mf.ctypes.data, rc.ctypes.data, ne.ctypes.data, rs.ctypes.data,
                                 tr.ctypes.data, trace_cap)
    assert st == 0, L.hpmvs_last_error()
    return xf, mf, rc, ne, rs, tr


def test_device_bobyqa_matches_goldens_of_the_genuine_library():
    probs = sp.problems()
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g5_bobyqa_selftest.json")))
    xf, mf, rc, ne, rs, tr = _device_run(probs)
    assert len(gold) == len(probs) == 64
    n_resc = 0
    for k, g in enumerate(gold):
        assert rc[k] == g["result"] and ne[k] == g["nevals"], (k, rc[k], ne[k], g)
        assert mf[k] == g["minf"] and list(xf[k]) == g["x"], k
        assert hashlib.sha256(tr[k, :min(ne[k], sp.TRACE_CAP)].tobytes()).hexdigest() == g["trace_sha256"], k
        assert rs[k] == g["rescue_calls"], (k, rs[k], g["rescue_calls"])   # rescue_ really ran on the device
        n_resc += int(rs[k] > 0)
    assert n_resc >= 10 and {-4, 1, 4, 5} <= set(int(v) for v in rc)


def test_device_bobyqa_matches_the_genuine_library_evaluation_by_evaluation():
    from oracle import oracle as orc
    assert orc.optimizer_available(orc.OPT_REF), "oracle/_ref (genuine NLopt BOBYQA) did not travel to this box"
    probs = sp.problems()
    ref = sp.reference_run(probs)
    xf, mf, rc, ne, rs, tr = _device_run(probs)
    for k, r in enumerate(ref):
        assert (rc[k], list(xf[k]), mf[k], ne[k]) == (r[0], r[1], r[2], r[3]), k
        assert np.array_equal(tr[k, :len(r[4])].view(np.int64), r[4].view(np.int64)), k   # every x and f, bit for bit


def test_device_bobyqa_wave_composition_does_not_matter():
    """The same problems in another order (other lane / wavefront neighbours, other label divergence): same bytes."""
    probs = sp.problems()
    perm = np.random.default_rng(1).permutation(len(probs))
    a = _device_run(probs, trace_cap=64)
    b = _device_run([probs[i] for i in perm], trace_cap=64)
    for x, y in zip(a, b):
        assert np.array_equal(x[perm], y)
