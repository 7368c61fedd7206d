"""The levels COMPOSED the way the reference's progressive loop composes them (main.cpp:145-181 -> CellProcessor::processCell,
src/hpmvs/CellProcessor.cpp:369-420): a level of `extend` at the leaves' width, another one over the patches it added, a level of
`branch` that splits every leaf (its patch taken out of the depth maps, the children entered, half the width), and `extend` again on
the children's level -- the batched device path (hpmvs_amd.frontier: extend_level / branch_level) beside the oracle's sequential
loops (orc_extend_round, orc_branch_round), each side fed with ITS OWN previous output.  After every level: stages, accepted set /
children and split decisions, occupancy, and every cell of every depth map are equal -- so the state one level hands to the next
(maps that hold subtractions, occupancy on a finer grid) is the reference's, not only the result of a single level on fresh maps."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MARGIN = 1.0


def _oracle_patches(batch):
    from oracle import oracle as orc
    arr = (orc.Patch * batch.n)()
    for k in range(batch.n):
        p = arr[k]
        p.center[:] = batch.center[k].tolist(); p.normal[:] = batch.normal[k].tolist()
        p.scale = float(batch.scale[k])
        p.n_images = int(batch.n_images[k])
        for i in range(p.n_images):
            p.images[i] = int(batch.images[k, i])
    return arr


def _maps_equal(gscene, OD, n_views, n_levels):
    from hpmvs_amd import api
    for v in range(n_views):
        for l in range(n_levels):
            a, b = api.depth_level(gscene, v, l), OD.level(v, l)
            if not np.array_equal(a, b):
                return False, (v, l, int((a != b).sum()))
    return True, None


def _same_patches(P, B, idx_cpu, idx_gpu):
    for a, b in zip(idx_cpu, idx_gpu):
        if not (np.array_equal(np.array(P[a].center[:], dtype=np.float32), B.center[b]) and
                np.array_equal(np.array(P[a].normal[:], dtype=np.float32), B.normal[b]) and
                np.float32(P[a].scale) == B.scale[b] and P[a].n_images == B.n_images[b] and
                list(P[a].images[:P[a].n_images]) == list(B.images[b, :B.n_images[b]])):
            return False
    return True


def test_extend_extend_branch_extend_against_the_sequential_loops():
    from hpmvs_amd import api, frontier, synth
    from oracle import oracle as orc
    scene = synth.make_scene(12, 640, 480, n_waves=24)
    seeds = synth.make_seeds(scene, 500, start_level=2, seed=synth.SEED + 77)
    g = api.Scene(scene, device=0)
    osc = orc.OracleScene(scene)
    n_levels = osc.camera(0).n_levels
    try:
        b0 = api.Batch.from_seeds(seeds)
        api.optimize_batch(g, b0)
        keep = np.nonzero(b0.ok)[0][:120]
        leaves_g = api.Batch(b0.center[keep], b0.normal[keep], b0.scale[keep], b0.n_images[keep], b0.images[keep])
        width = float(np.float32(2.0 * np.median(leaves_g.scale) / 0.9))
        occ_g = {orc.cell_key(leaves_g.center[k], width) for k in range(leaves_g.n)}
        occ_c = set(occ_g)
        api.depth_reset(g)
        leaves_g.ok[:] = 1
        api.set_depths_batch(g, leaves_g)
        OD = orc.OracleDepths(osc)
        leaves_c = _oracle_patches(leaves_g)
        for k in range(leaves_g.n):
            OD.set_depths(leaves_c[k])
        all_g, all_c = leaves_g, list(leaves_c)     # every patch in the tree so far (the branch level splits all of them)
        report = []

        def extend(tag, lg, lc, w):
            nonlocal all_g, all_c
            res = frontier.extend_level(g, lg, w, occ_g, MARGIN, 0)
            cand, cnt = orc.extend_round(osc, OD, lc, w, occ_c, MARGIN, 0, frozen_gates=False, which=orc.OPT_REF)
            st = np.array([cand[t].stage for t in range(6 * lg.n)])
            assert np.array_equal(res.stage, st), (tag, np.nonzero(res.stage != st)[0][:10])
            assert np.array_equal(res.counts, cnt), tag
            acc = [t for t in range(6 * lg.n) if st[t] == 0]
            assert res.accepted == acc, tag
            assert _same_patches(cand, res.candidates, acc, acc), tag
            assert occ_g == occ_c, tag
            ok, where = _maps_equal(g, OD, scene.n_views, n_levels)
            assert ok, (tag, where)
            a = np.array(acc, dtype=np.int64)
            out = res.candidates
            new_g = api.Batch(out.center[a], out.normal[a], out.scale[a], out.n_images[a], out.images[a])
            new_c = (orc.Patch * len(acc))(*[cand[t] for t in acc])
            m = max(all_g.max_images, new_g.max_images)
            pad = lambda x: np.pad(x, ((0, 0), (0, m - x.shape[1])), constant_values=-1)
            all_g = api.Batch(np.concatenate([all_g.center, new_g.center]), np.concatenate([all_g.normal, new_g.normal]),
                              np.concatenate([all_g.scale, new_g.scale]), np.concatenate([all_g.n_images, new_g.n_images]),
                              np.concatenate([pad(all_g.images), pad(new_g.images)]))
            all_c = all_c + [new_c[k] for k in range(len(acc))]
            report.append((tag, lg.n, len(acc), res.waves))
            return new_g, new_c

        l1_g, l1_c = extend("extend 1", leaves_g, leaves_c, width)
        assert l1_g.n >= 20
        l2_g, l2_c = extend("extend 2", l1_g, l1_c, width)

        # ---- branch: every leaf is split; its cell is the grid cell that holds its patch
        n = all_g.n
        P_all = (orc.Patch * n)(*all_c)
        w32 = np.float32(width)
        cc = ((np.floor(all_g.center[:, :3] / w32) + np.float32(0.5)) * w32).astype(np.float32)
        cw = np.full(n, w32, np.float32)
        final = np.zeros(n, np.uint8)
        res = frontier.branch_level(g, all_g, cc, cw, final)
        cand, split = orc.branch_round(osc, OD, P_all, cc, cw, final, which=orc.OPT_REF)
        st = np.array([cand[t].stage for t in range(4 * n)])
        assert np.array_equal(np.where(st == 27, 20, st), res.candidates.stage), np.nonzero(np.where(st == 27, 20, st) != res.candidates.stage)[0][:10]
        assert np.array_equal(split, res.split)
        kids = [t for t in range(4 * n) if st[t] == 0]
        assert kids == np.nonzero(res.children.reshape(-1))[0].tolist() and len(kids) >= 50
        assert _same_patches(cand, res.candidates, kids, kids)
        ok, where = _maps_equal(g, OD, scene.n_views, n_levels)
        assert ok, ("branch", where)
        report.append(("branch", n, len(kids), int(split.sum())))

        # ---- the children's level: half the width, occupancy on the finer grid
        width2 = float(np.float32(width / 2))
        a = np.array(kids, dtype=np.int64)
        out = res.candidates
        kids_g = api.Batch(out.center[a], out.normal[a], out.scale[a], out.n_images[a], out.images[a])
        kids_c = (orc.Patch * len(kids))(*[cand[t] for t in kids])
        occ_g.clear(); occ_g.update(orc.cell_key(kids_g.center[k], width2) for k in range(kids_g.n))
        occ_c.clear(); occ_c.update(occ_g)
        all_g, all_c = kids_g, list(kids_c)
        first = min(kids_g.n, 150)
        sub_g = api.Batch(kids_g.center[:first], kids_g.normal[:first], kids_g.scale[:first], kids_g.n_images[:first], kids_g.images[:first])
        sub_c = (orc.Patch * first)(*[kids_c[k] for k in range(first)])
        l3_g, _ = extend("extend 3 (children's level)", sub_g, sub_c, width2)
        print("progressive levels (level, leaves in, accepted / children, waves / split):", report)
        assert report[0][3] > 1 and report[-1][2] >= 5
    finally:
        g.close()
