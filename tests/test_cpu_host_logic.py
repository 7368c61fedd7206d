"""Host logic of the C++ layer that needs no device: the conflict graph of PatchOptimizer::extendLevel (who would write a depth-map
cell whom reads: bucketed rectangle tests, OpenMP) against the definition itself -- the sets of cell keys -- on random footprints
with blocks at and beyond the image border, writes on every pyramid level and colliding cells (hpmvs_amd/host/hpmvs_host.cpp,
`hpmvs_host_selftest_conflict_graph`).  The walk that uses the graph is held to the reference's sequential loop on the GPU box
(tests/test_gpu_expand_round.py, tests/test_gpu_cpp_interface.py, tools/extend_level_scale.py)."""
import ctypes as C
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host_lib():
    import __graft_entry__  # noqa: F401  (puts the repo root on sys.path)
    path = os.path.join(ROOT, "hpmvs_amd", "libhpmvs_host.so")
    if not os.path.exists(path):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "hpmvs_amd", "csrc")])
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "hpmvs_amd", "host")])
    L = C.CDLL(path)
    L.hpmvs_host_selftest_conflict_graph.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    return L


@pytest.mark.parametrize("seed,n,views,w,h,levels,m", [
    (1, 300, 3, 640, 480, 6, 3),        # configs[0]'s shape: every bucket size the chooser can pick is a single cell or a few
    (2, 3000, 12, 1920, 1080, 6, 8),    # above the OpenMP threshold (2048 candidates)
    (3, 2500, 50, 3840, 2160, 6, 12),
    (4, 500, 2, 64, 48, 6, 4),          # images smaller than the coarsest bucket
    (5, 800, 4, 800, 600, 3, 5),        # fewer pyramid levels
    (6, 1, 1, 640, 480, 6, 2),          # one candidate: no edges
])
def test_conflict_graph_equals_the_definition(host_lib, seed, n, views, w, h, levels, m):
    edges = C.c_int(-1)
    assert host_lib.hpmvs_host_selftest_conflict_graph(seed, n, views, w, h, levels, m, C.byref(edges)) == 0
    print("conflict graph:", n, "candidates,", edges.value, "edges")
    assert edges.value >= (10 * n if n > 1 else 0)   # the footprints do meet: the comparison is not one of empty sets


def test_conflict_graph_selftest_refuses_bad_arguments(host_lib):
    assert host_lib.hpmvs_host_selftest_conflict_graph(1, 0, 3, 640, 480, 6, 3, None) == -1
    assert host_lib.hpmvs_host_selftest_conflict_graph(1, 10, 3, 640, 480, 9, 3, None) == -1
