"""Shared test helpers (CPU and GPU tests)."""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
INF = float("inf")
LB = [-INF, -23.99999, -23.99999]  # reference src/hpmvs/PatchOptimizer.cpp:326-336
UB = [INF, 23.99999, 23.99999]


def golden_scene():
    """tests/golden/g1_scene.npz -> (SynthScene, Seeds, expected dict)."""
    from hpmvs_amd import synth
    z = np.load(os.path.join(GOLDEN, "g1_scene.npz"))
    views = [synth.View(int(z["rgb"].shape[2]), int(z["rgb"].shape[1]), float(z["f"][i]), z["q"][i], z["c"][i],
                        np.ascontiguousarray(z["rgb"][i])) for i in range(z["rgb"].shape[0])]
    covis = [[j for j in range(len(views)) if z["covis"][i, j]] for i in range(len(views))]
    scene = synth.SynthScene(views=views, covis=covis, max_level=5)
    seeds = synth.Seeds(center=z["center"], normal=z["normal"], scale=z["scale"], n_images=z["n_images"],
                        images=z["images"], truth=np.zeros((len(z["scale"]), 3)))
    expected = json.load(open(os.path.join(GOLDEN, "g1_expected.json")))
    return scene, seeds, expected


def bq_host():
    """Host build of the product's BOBYQA state machine (tests/native/bq_host.cpp)."""
    import subprocess
    from oracle import oracle as orc
    d = os.path.join(HERE, "native")
    subprocess.run(["make", "-C", d, "-s"], check=True, capture_output=True)
    H = C.CDLL(os.path.join(d, "libbq_host.so"))
    H.bq3_minimize.argtypes = [orc.OBJ_FN, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                               C.POINTER(C.c_double), C.c_double, C.c_int, C.POINTER(C.c_double),
                               C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_int]
    return H


def minimize_both(pyf, x0, lb=LB, ub=UB, xtol=1e-7, maxeval=1000, which=0):
    """Run genuine NLopt BOBYQA (oracle, `which`) and the product state machine on the same Python
    objective; returns [(rc, x, minf, nevals)]*2 and the two evaluation traces."""
    from oracle import oracle as orc
    L, H = orc.lib(), bq_host()
    traces = [[], []]

    def mk(i):
        def cb(n, x, g, d):
            xx = [x[0], x[1], x[2]]
            v = pyf(xx)
            traces[i].append((xx, v))
            return v
        return orc.OBJ_FN(cb)

    res = []
    for i in range(2):
        x = (C.c_double * 3)(*x0); l = (C.c_double * 3)(*lb); u = (C.c_double * 3)(*ub)
        mf, ne = C.c_double(), C.c_int()
        if i == 0:
            rc = L.orc_bobyqa_minimize(which, 3, mk(0), None, x, l, u, xtol, maxeval, C.byref(mf), C.byref(ne))
        else:
            rc = H.bq3_minimize(mk(1), None, x, l, u, xtol, maxeval, C.byref(mf), C.byref(ne), None, 0)
        res.append((rc, list(x), mf.value, ne.value))
    return res, traces


def equals_gpu_with_correctly_rounded_acos(osc, oc, seeds, k, b):
    """Patch k again through the oracle with its acos() correctly rounded (libquadmath; what the device's cr_acos.hpp returns):
    does the oracle then give the GPU's result bit for bit?  DESIGN section 5: glibc's acos is < 1 ulp but not correctly rounded,
    and that last bit of the optimiser's start point is the one known source of GPU-vs-oracle differences."""
    from oracle import oracle as orc
    P = orc.patches_from_seeds(seeds, [k])
    orc.set_libm_mode(1)
    try:
        orc.optimize_batch(osc, P, which=orc.best_optimizer(), options=oc, n_threads=1)
    finally:
        orc.set_libm_mode(0)
    p = P[0]
    if p.stage != b.stage[k]:
        return False
    if p.stage != 0:
        return True
    return (np.array_equal(np.array(p.center[:], dtype=np.float32), b.center[k]) and
            np.array_equal(np.array(p.normal[:], dtype=np.float32), b.normal[k]) and
            np.array_equal(np.array(p.color[:], dtype=np.float32), b.color[k]) and
            list(p.images[:p.n_images]) == list(b.images[k, :b.n_images[k]]) and p.nevals == b.nevals[k])
