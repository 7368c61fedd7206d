"""One COMPOSED expansion round on the GPU against the reference's sequential loop.

Reference: main.cpp:146-181 pops the leaves of one priority level and CellProcessor::extend (CellProcessor.cpp:84-178)
runs on each, one after the other: candidate -> octree pre-gate -> optimize -> scale / drift gates -> the three depth-map
counts (Scene.cpp:518-644) -> DynOctTree::addConditional (doctree.h:397-419) -> Scene::setDepths (Scene.cpp:351-381).
The oracle restates that loop (`orc_extend_round`, octree = a caller-owned occupancy set over the level's uniform grid).

The GPU round is the frontier recipe of INTEGRATION.md: ONE hpmvs_expand_batch for all candidates of the level (refined and
scale/drift-gated in a single launch), ONE hpmvs_depth_gates_batch against the depth maps as they are when the round
starts, then the host walks the candidates in the reference's order -- re-checking the pre-optimisation cell against the
live occupancy, applying the count thresholds, inserting -- and ONE hpmvs_set_depths_batch enters the accepted ones.

What must hold, and does:
  * GPU round == the sequential loop with the counts read from the round-start maps (`frozen_gates`), candidate by
    candidate: stage codes, refined fields bit for bit, the three counts, accepted set, occupancy, every depth map.
    Two consecutive priority levels (the accepted candidates of the first are the leaves of the second).
  * Against the TRUE sequential loop (counts from the live maps) the only differences are interactions through the depth
    maps inside one level: the first candidate that differs has different counts in the two modes; everything before it
    is identical.  How many candidates that touches is printed and recorded (tests do not hide it).
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MARGIN = 1.0   # HpmvsOptions::DEPTH_TEST_FACTOR (HpmvsOptions.h:53)
MIN_IMAGES = 3  # MIN_IMAGES_PER_PATCH


def _oracle_patches(par):
    from oracle import oracle as orc
    P = (orc.Patch * par.n)()
    for k in range(par.n):
        P[k].center[:] = par.center[k].tolist()
        P[k].normal[:] = par.normal[k].tolist()
        P[k].scale = float(par.scale[k])
        P[k].n_images = int(par.n_images[k])
        for j in range(par.n_images[k]):
            P[k].images[j] = int(par.images[k, j])
    return P


def _gpu_round(gpu_scene, parents, width, occupied, abs_int):
    """The frontier recipe.  Returns (candidates batch, stage per candidate, counts [6n,3], accepted indices)."""
    from hpmvs_amd import api
    from oracle import oracle as orc
    n = parents.n
    cc = np.zeros((n, 3), np.float32)
    widths = np.full(n, width, np.float32)
    # the candidates before optimize (everything skipped: constructed only) -> their octree cells
    pre = api.expand_batch(gpu_scene, api.EXPAND_EXTEND, parents, cc, widths, np.ones(6 * n, np.uint8))
    pre_key = [orc.cell_key(pre.center[t], width) for t in range(6 * n)]
    skip = np.array([k in occupied for k in pre_key], np.uint8)   # round-start occupancy: saves their refinement
    out = api.expand_batch(gpu_scene, api.EXPAND_EXTEND, parents, cc, widths, skip)
    v, b, f = api.depth_gates_batch(gpu_scene, out, MARGIN, abs_int)   # round-start depth maps
    stage = out.stage.copy()
    counts = np.full((6 * n, 3), -1, np.int32)
    accepted = []
    for t in range(6 * n):   # the reference's order: leaf by leaf, candidate by candidate
        if pre_key[t] in occupied:       # (live occupancy: a cell filled earlier in this round skips the candidate too)
            stage[t] = 20
            continue
        if stage[t] != 0:
            continue
        counts[t] = (v[t], b[t], f[t])
        if not v[t] >= MIN_IMAGES:
            stage[t] = 23
        elif not b[t] < MIN_IMAGES:
            stage[t] = 24
        elif not (f[t] >= MIN_IMAGES - 1 and f[t] * 1.0 / out.n_images[t] > 0.75):
            stage[t] = 25
        else:
            key = orc.cell_key(out.center[t], width)
            if key in occupied:
                stage[t] = 26
            else:
                occupied.add(key)
                accepted.append(t)
    if accepted:
        a = np.array(accepted)
        acc = api.Batch(out.center[a], out.normal[a], out.scale[a], out.n_images[a], out.images[a])
        acc.ok[:] = 1
        api.set_depths_batch(gpu_scene, acc)
    return out, stage, counts, accepted


def _maps_equal(gpu_scene, odepths, scene):
    from hpmvs_amd import api
    for v in range(scene.n_views):
        for l in range(6):
            a = api.depth_level(gpu_scene, v, l)
            b = odepths.level(v, l)
            if not np.array_equal(a, b):
                return False, (v, l, int((a != b).sum()))
    return True, None


@pytest.mark.parametrize("abs_int", [0, 1])
def test_two_composed_expansion_levels_against_the_sequential_loop(abs_int, tiny_scene, tiny_seeds, oracle_scene, gpu_scene):
    from hpmvs_amd import api
    from oracle import oracle as orc
    b0 = api.Batch.from_seeds(tiny_seeds)
    api.optimize_batch(gpu_scene, b0)
    keep = np.nonzero(b0.ok)[0][:60]
    parents = api.Batch(b0.center[keep], b0.normal[keep], b0.scale[keep], b0.n_images[keep], b0.images[keep])
    width = float(np.float32(2.0 * np.median(parents.scale) / 0.9))
    occ0 = {orc.cell_key(parents.center[k], width) for k in range(parents.n)}
    # depth maps as the seed round leaves them: every seed's depths entered
    api.depth_reset(gpu_scene)
    parents.ok[:] = 1
    api.set_depths_batch(gpu_scene, parents)
    OD = {m: orc.OracleDepths(oracle_scene) for m in ("frozen", "live")}
    P0 = _oracle_patches(parents)
    for m in OD:
        for k in range(parents.n):
            OD[m].set_depths(P0[k])
    ok, where = _maps_equal(gpu_scene, OD["frozen"], tiny_scene)
    assert ok, where

    occ_gpu = set(occ0)
    occ = {"frozen": set(occ0), "live": set(occ0)}
    leaves_gpu = parents
    leaves = {"frozen": P0, "live": P0}
    report = {"abs_int": abs_int, "levels": []}
    for level in range(2):
        n = leaves_gpu.n
        out, stage, counts, accepted = _gpu_round(gpu_scene, leaves_gpu, width, occ_gpu, abs_int)
        ref = {}
        for m in ("frozen", "live"):
            ref[m] = orc.extend_round(oracle_scene, OD[m], leaves[m], width, occ[m], MARGIN, abs_int, frozen_gates=(m == "frozen"))
        cand, cnt = ref["frozen"]
        st_f = np.array([cand[t].stage for t in range(6 * n)])
        # ---- the GPU round is the sequential loop with round-start counts, candidate by candidate
        assert np.array_equal(stage, st_f), (level, np.nonzero(stage != st_f)[0][:10], stage[:18], st_f[:18])
        assert np.array_equal(counts, cnt), (level, np.nonzero((counts != cnt).any(axis=1))[0][:10])
        for t in range(6 * n):
            if st_f[t] in (0, 21, 22, 23, 24, 25, 26):   # refined candidates: every field
                assert np.array_equal(np.array(cand[t].center[:], dtype=np.float32), out.center[t]), (level, t)
                assert np.array_equal(np.array(cand[t].normal[:], dtype=np.float32), out.normal[t]), (level, t)
                m_ = cand[t].n_images
                assert m_ == out.n_images[t] and list(cand[t].images[:m_]) == list(out.images[t, :m_]), (level, t)
        assert occ_gpu == occ["frozen"], level
        ok, where = _maps_equal(gpu_scene, OD["frozen"], tiny_scene)
        assert ok, (level, where)
        acc_f = [t for t in range(6 * n) if st_f[t] == 0]
        assert accepted == acc_f and len(accepted) >= 5, (level, len(accepted))
        # ---- against the true sequential loop: only interactions through the depth maps inside the level
        cand_l, cnt_l = ref["live"]
        n_l = len(leaves["live"])   # (from the second level on the true sequential loop follows its own leaves)
        st_l = np.array([cand_l[t].stage for t in range(6 * n_l)])
        lv = {"level": level, "leaves": n, "candidates": 6 * n, "accepted_frontier": len(acc_f),
              "accepted_sequential": int((st_l == 0).sum()), "stage_histogram": {int(k): int(v) for k, v in zip(*np.unique(st_f, return_counts=True))}}
        if level == 0:
            diff = np.nonzero(st_l != st_f)[0]
            lv["candidates_whose_fate_differs"] = int(len(diff))
            if len(diff):
                first = int(diff[0])
                # everything before the first difference is identical, and the first difference is a gate count that the
                # earlier candidates of the same level changed
                assert np.array_equal(st_l[:first], st_f[:first])
                assert not np.array_equal(cnt_l[first], cnt[first]), (first, cnt_l[first], cnt[first])
                lv["first_difference"] = {"candidate": first, "counts_live": cnt_l[first].tolist(), "counts_round_start": cnt[first].tolist(),
                                          "stage_live": int(st_l[first]), "stage_round_start": int(st_f[first])}
        report["levels"].append(lv)
        # next level: the accepted candidates are the new leaves (each mode follows its own)
        a = np.array(acc_f)
        leaves_gpu = api.Batch(out.center[a], out.normal[a], out.scale[a], out.n_images[a], out.images[a])
        leaves = {"frozen": _oracle_patches(leaves_gpu),
                  "live": (orc.Patch * int((st_l == 0).sum()))(*[cand_l[t] for t in range(6 * n_l) if st_l[t] == 0])}
    print("composed expansion round:", json.dumps(report))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "expand_round_report.jsonl"), "a") as fh:
        fh.write(json.dumps(report) + "\n")


# ---- the frontier round with the reference's SEQUENTIAL result (hpmvs_amd/frontier.py: conflict-free waves)
def _sequential_levels(tag, scene, seeds, oscene, gscene, n_parents, abs_int, levels=2):
    import time
    from hpmvs_amd import api, frontier
    from oracle import oracle as orc
    b0 = api.Batch.from_seeds(seeds)
    api.optimize_batch(gscene, b0)
    keep = np.nonzero(b0.ok)[0][:n_parents]
    parents = api.Batch(b0.center[keep], b0.normal[keep], b0.scale[keep], b0.n_images[keep], b0.images[keep])
    width = float(np.float32(2.0 * np.median(parents.scale) / 0.9))
    occ_gpu = {orc.cell_key(parents.center[k], width) for k in range(parents.n)}
    occ_cpu = set(occ_gpu)
    api.depth_reset(gscene)
    parents.ok[:] = 1
    api.set_depths_batch(gscene, parents)
    OD = orc.OracleDepths(oscene)
    leaves_cpu = _oracle_patches(parents)
    for k in range(parents.n):
        OD.set_depths(leaves_cpu[k])
    leaves_gpu = parents
    report = {"scene": tag, "abs_int": abs_int, "levels": []}
    for level in range(levels):
        n = leaves_gpu.n
        t0 = time.perf_counter()
        res = frontier.extend_level(gscene, leaves_gpu, width, occ_gpu, MARGIN, abs_int)
        t_gpu = time.perf_counter() - t0
        t0 = time.perf_counter()
        cand, cnt = orc.extend_round(oscene, OD, leaves_cpu, width, occ_cpu, MARGIN, abs_int, frozen_gates=False)   # the TRUE sequential loop
        t_cpu = time.perf_counter() - t0
        st = np.array([cand[t].stage for t in range(6 * n)])
        # what the plain frontier round (level-start maps) would have done, for the record: on a copy of the state before this level
        differing = np.nonzero(res.stage != st)[0]
        assert len(differing) == 0, (tag, level, "fates differ from the sequential loop", differing[:10], res.stage[differing[:10]], st[differing[:10]])
        assert np.array_equal(res.counts, cnt), (tag, level, np.nonzero((res.counts != cnt).any(axis=1))[0][:10])
        for t in range(6 * n):
            if st[t] in (0, 21, 22, 23, 24, 25, 26):
                assert np.array_equal(np.array(cand[t].center[:], dtype=np.float32), res.candidates.center[t]), (tag, level, t)
                assert np.array_equal(np.array(cand[t].normal[:], dtype=np.float32), res.candidates.normal[t]), (tag, level, t)
        assert occ_gpu == occ_cpu, (tag, level)
        ok, where = _maps_equal(gscene, OD, scene)
        assert ok, (tag, level, where)
        acc = [t for t in range(6 * n) if st[t] == 0]
        assert res.accepted == acc and len(acc) >= 5, (tag, level, len(acc))
        report["levels"].append({"level": level, "leaves": n, "candidates": 6 * n, "accepted": len(acc), "waves": res.waves,
                                 "deferred_per_wave": res.deferred_per_wave, "fates_differing_from_sequential_loop": 0,
                                 "frontier_s": round(t_gpu, 4), "sequential_cpu_oracle_s": round(t_cpu, 4),
                                 "stage_histogram": {int(k): int(v) for k, v in zip(*np.unique(st, return_counts=True))}})
        a = np.array(acc)
        out = res.candidates
        leaves_gpu = api.Batch(out.center[a], out.normal[a], out.scale[a], out.n_images[a], out.images[a])
        leaves_cpu = (orc.Patch * len(acc))(*[cand[t] for t in acc])
    print("sequential frontier:", json.dumps(report))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "expand_round_sequential.jsonl"), "a") as fh:
        fh.write(json.dumps(report) + "\n")
    return report


@pytest.mark.parametrize("abs_int", [0, 1])
def test_frontier_waves_equal_the_sequential_loop_on_configs0(abs_int, tiny_scene, tiny_seeds, oracle_scene, gpu_scene):
    """BASELINE configs[0]: two consecutive priority levels, ZERO candidates whose fate differs from CellProcessor::extend run
    leaf by leaf (the plain frontier round of the test above differs in 27 of 360 on the first level)."""
    rep = _sequential_levels("configs0_3v_640x480", tiny_scene, tiny_seeds, oracle_scene, gpu_scene, 60, abs_int)
    assert all(lv["waves"] >= 1 for lv in rep["levels"])
    assert rep["levels"][0]["waves"] > 1   # the level DOES interact through the maps: one wave would be the plain frontier round


def test_frontier_waves_equal_the_sequential_loop_on_a_12_view_scene():
    """The same on a 12-view scene (more views per candidate: viewBlockTest reads every one of them)."""
    from hpmvs_amd import api, synth
    from oracle import oracle as orc
    scene = synth.make_scene(12, 640, 480, n_waves=24)
    seeds = synth.make_seeds(scene, 600, start_level=2)
    g = api.Scene(scene, device=0)
    try:
        rep = _sequential_levels("12v_640x480", scene, seeds, orc.OracleScene(scene), g, 150, 0)
    finally:
        g.close()
    assert rep["levels"][0]["candidates"] >= 600
