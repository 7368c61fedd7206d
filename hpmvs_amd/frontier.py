"""One priority level of the expansion as a FRONTIER -- batched on the GPU -- with the reference's SEQUENTIAL result.

Reference: main.cpp:146-181 pops the leaves of one priority level and CellProcessor::extend (CellProcessor.cpp:84-178) runs
on them one after the other: candidate -> octree pre-gate (:118-122) -> optimize -> scale / drift gates (:124-128) ->
the three depth-map counts (:130-139, Scene.cpp:518-644) -> DynOctTree::addConditional (doctree.h:397-419) ->
Scene::setDepths (Scene.cpp:351-381).  Candidate i's counts are read from maps that the candidates accepted BEFORE it in the
same level have already written, and its octree cell may have been taken by one of them.

`extend_level` keeps that meaning and the batching:
  * ONE hpmvs_expand_batch refines every candidate of the level (the refinement reads neither the maps nor the octree);
  * ONE hpmvs_depth_footprints_batch names, per candidate, the map cells its gates read and the cells setDepths would write;
  * the candidates are then decided in WAVES.  A wave = one hpmvs_depth_gates_batch over the still undecided candidates
    against the maps as they are, a walk over them in the reference's order, one hpmvs_set_depths_batch for the accepted.
    In the walk a candidate is DECIDED (accepted or rejected for good) unless something it depends on is still open:
      - a map cell it reads may be written by an earlier candidate of this walk that was accepted or deferred (`dirty`),
      - a cell it would write is read by an earlier deferred candidate (`guard`: that one must not see this write later),
      - its octree cell (before or after refinement) is the possible cell of an earlier deferred candidate (`maybe_occ`),
        or an earlier deferred candidate looks at the cell it would occupy (`occ_guard`);
    then it is DEFERRED to the next wave, and leaves its own possible effects in those sets.  The first undecided candidate
    of a wave always gets decided, so the waves end; their number is the depth of the dependency chains, not the number of
    candidates (tests/test_gpu_expand_round.py records it: a handful).
  The result -- stage codes, counts, accepted set, occupancy, every depth map -- equals the sequential loop's, candidate by
  candidate (asserted against the oracle's `orc_extend_round` on BASELINE configs[0] and on a 12-view scene).

The octree itself stays with the scheduler (out of scope): `occupied` is the caller's set of cell keys,
`cell_key` the caller's map from a point to its leaf (default: the uniform grid of leaf width `width`).
Stage codes as in hpmvs_expand_batch, plus 20 = leaf already taken (no refinement), 23 / 24 / 25 = depthTests /
viewBlockTest / pixelFreeTests threshold, 26 = addConditional found the refined patch's leaf taken.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import api


def cell_key(p, width) -> int:
    """Key of the grid cell floor(p / width) per axis (float32 division, as a leaf look-up on the refined centre)."""
    w = np.float32(width)
    ix, iy, iz = (int(np.floor(np.float32(p[k]) / w)) for k in range(3))
    return ((ix + (1 << 20)) << 42) | ((iy + (1 << 20)) << 21) | (iz + (1 << 20))


def _cell(view, level, x, y):
    return (((int(view) << 3) | int(level)) << 48) | ((int(x) & 0xFFFFFF) << 24) | (int(y) & 0xFFFFFF)


def _full_depth_cells(view, ix0, iy0, n_levels, out):
    """Cells Scene::getFullDepth visits for the 3x3 level-0 pixel block from (ix0, iy0) (Scene.cpp:406-432, 538-550):
    (pixel / DEPTH_SUBSAMPLE) >> level on every level.  Pixels outside the image are never looked up; keeping them would
    only add keys that no write can have."""
    seen = set()
    for py in (iy0, iy0 + 1, iy0 + 2):
        if py < 0:
            continue
        for px in (ix0, ix0 + 1, ix0 + 2):
            if px < 0:
                continue
            c0 = (px >> 1, py >> 1)
            if c0 in seen:
                continue
            seen.add(c0)
            x, y = c0
            for l in range(n_levels):
                out.add(_cell(view, l, x, y))
                x >>= 1; y >>= 1


@dataclass
class LevelResult:
    candidates: api.Batch          # the 6 n candidates (refined where they were refined)
    stage: np.ndarray              # final stage code per candidate (0 = accepted and inserted)
    counts: np.ndarray             # [6 n, 3] depthTests / viewBlockTest / pixelFreeTests at decision time (-1: not reached)
    accepted: list                 # candidate indices in the reference's order
    waves: int                     # gate / setDepths passes it took
    deferred_per_wave: list = field(default_factory=list)


def extend_level(scene: api.Scene, parents: api.Batch, width: float, occupied: set, margin: float = 1.0, abs_int: int = 0,
                 options=None, n_levels: int = 6, key=cell_key, sequential: bool = True) -> LevelResult:
    """CellProcessor::extend over `parents` (the leaves of one priority level, in the scheduler's order).  `occupied` is
    updated in place; the scene's depth maps receive the accepted candidates.  sequential = False gives round 3's plain
    frontier round (every count read from the maps as they are when the level starts): one wave, not the reference's
    result when candidates of a level interact through the maps."""
    o = options or api.default_options()
    MIN = int(o.MIN_IMAGES_PER_PATCH)
    n = parents.n
    N = 6 * n
    cc = np.zeros((n, 3), np.float32)
    widths = np.full(n, width, np.float32)
    # the candidates before optimize (everything skipped: constructed only) -> their leaves
    pre = api.expand_batch(scene, api.EXPAND_EXTEND, parents, cc, widths, np.ones(N, np.uint8), options=o)
    pre_key = [key(pre.center[t], width) for t in range(N)]
    skip = np.array([k in occupied for k in pre_key], np.uint8)   # level-start occupancy: those are never refined
    out = api.expand_batch(scene, api.EXPAND_EXTEND, parents, cc, widths, skip, options=o)
    refined = (out.stage == 0) & (skip == 0)
    post_key = [key(out.center[t], width) if refined[t] else None for t in range(N)]
    wr, fr, at, vb = api.depth_footprints_batch(scene, out)
    V = scene.n_views

    reads_cache, writes_cache = {}, {}

    def reads(t):
        r = reads_cache.get(t)
        if r is None:
            r = set()
            for k in range(int(out.n_images[t])):
                if at[t, k, 0] >= 0:
                    _full_depth_cells(at[t, k, 0], int(at[t, k, 1]), int(at[t, k, 2]), n_levels, r)
                if fr[t, k, 0] >= 0:
                    r.add(_cell(*fr[t, k]))
            for v in range(V):
                if vb[t, v, 0]:
                    _full_depth_cells(v, int(vb[t, v, 1]), int(vb[t, v, 2]), n_levels, r)
            reads_cache[t] = r
        return r

    def writes(t):
        w = writes_cache.get(t)
        if w is None:
            w = {_cell(*wr[t, k]) for k in range(int(out.n_images[t])) if wr[t, k, 0] >= 0}
            writes_cache[t] = w
        return w

    stage = np.where(skip != 0, 20, out.stage).astype(np.int32)
    counts = np.full((N, 3), -1, np.int32)
    accepted = []
    pending = [t for t in range(N) if not skip[t]]
    waves = 0
    deferred_log = []
    while pending:
        waves += 1
        todo = [t for t in pending if refined[t]]
        cnt = {}
        if todo:
            a = np.array(todo)
            sub = api.Batch(out.center[a], out.normal[a], out.scale[a], out.n_images[a], out.images[a])
            v, b, f = api.depth_gates_batch(scene, sub, margin, abs_int)
            cnt = {t: (int(v[i]), int(b[i]), int(f[i])) for i, t in enumerate(todo)}
        dirty, guard, maybe_occ, occ_guard = set(), set(), set(), set()
        deferred, acc_now = [], []

        def defer(t):
            deferred.append(t)
            occ_guard.add(pre_key[t])
            if refined[t]:
                dirty.update(writes(t)); guard.update(reads(t))
                maybe_occ.add(post_key[t]); occ_guard.add(post_key[t])

        for t in pending:
            pk = pre_key[t]
            if pk in occupied:
                stage[t] = 20                       # its leaf was taken (by an earlier candidate: occupancy only grows in order)
                continue
            if sequential and pk in maybe_occ:
                defer(t)
                continue
            if not refined[t]:
                stage[t] = out.stage[t]             # failed in optimize or at the scale / drift gates: reads no map
                continue
            if sequential and not reads(t).isdisjoint(dirty):
                defer(t)
                continue
            v_, b_, f_ = cnt[t]
            counts[t] = (v_, b_, f_)
            if not v_ >= MIN:
                stage[t] = 23
            elif not b_ < MIN:
                stage[t] = 24
            elif not (f_ >= MIN - 1 and f_ * 1.0 / int(out.n_images[t]) > 0.75):
                stage[t] = 25
            else:
                k_ = post_key[t]
                if k_ in occupied:
                    stage[t] = 26
                elif sequential and (k_ in maybe_occ or k_ in occ_guard or not writes(t).isdisjoint(guard)):
                    counts[t] = (-1, -1, -1)
                    defer(t)
                else:
                    occupied.add(k_)
                    stage[t] = 0
                    accepted.append(t); acc_now.append(t)
                    dirty.update(writes(t))
        if acc_now:
            a = np.array(acc_now)
            acc = api.Batch(out.center[a], out.normal[a], out.scale[a], out.n_images[a], out.images[a])
            acc.ok[:] = 1
            api.set_depths_batch(scene, acc)
        deferred_log.append(len(deferred))
        pending = deferred
    accepted.sort()
    return LevelResult(out, stage, counts, accepted, waves, deferred_log)


@dataclass
class BranchResult:
    candidates: api.Batch          # the 4 n children (refined where they were refined; stage 20: not built / not refined)
    support: np.ndarray            # Scene::getLevelSupport of every leaf's patch
    split: np.ndarray              # [n] 1: the leaf was split (its patch's depths taken back, the children's entered)
    children: np.ndarray           # [n, 4] bool: the children that go into the new leaves


def branch_level(scene: api.Scene, parents: api.Batch, cell_center, cell_width, final_level, options=None) -> BranchResult:
    """One priority level of CellProcessor::branch (reference CellProcessor.cpp:210-307) over `parents` -- the patches of the
    level's leaves, in the scheduler's order -- as ONE hpmvs_level_support_batch (the first gate, :221-224), ONE
    hpmvs_expand_batch (the four diagonal children, Cell::contains before and after optimize, :233-258: the refinement reads
    neither the maps nor the tree) and ONE hpmvs_depth_ops_batch that replays the leaves' map updates in the reference's
    order: per split leaf its patch's depths taken back (:276-279), then its children's entered (:296) -- subtraction does not
    commute with the minimum, so the order is part of the result.  `final_level[i]`: nodeLevel(leaf i) >= PATCH_FINAL_MINLEVEL
    (the scheduler's knowledge): such a leaf keeps its patch when no child survived (:265-266).  The tree operations (split,
    the children's leaves, the queue) stay with the caller; the result names what to do.  Equals the sequential loop's
    (`orc_branch_round`): children, split decisions, every depth map (tests/test_gpu_branch_level.py)."""
    o = options or api.default_options()
    n = parents.n
    cc = np.ascontiguousarray(cell_center, dtype=np.float32).reshape(n, 3)
    cw = np.ascontiguousarray(cell_width, dtype=np.float32).reshape(n)
    final = np.ascontiguousarray(final_level).astype(bool).reshape(n)
    support = api.level_support_batch(scene, parents, int(o.MINLEVEL))
    skip = np.repeat(support < 1, 4).astype(np.uint8)   # an exhausted leaf builds nothing
    out = api.expand_batch(scene, api.EXPAND_BRANCH, parents, cc, cw, skip, options=o)
    children = ((out.stage == 0) & (skip == 0)).reshape(n, 4)
    split = (support >= 1) & ~(final & (children.sum(axis=1) == 0))
    # the map updates in the reference's order: leaf by leaf, the old patch out, the children in
    M = max(parents.max_images, out.max_images)
    widen = lambda a: np.pad(a, ((0, 0), (0, M - a.shape[1])), constant_values=-1)
    rows_c, rows_n, rows_s, rows_m, rows_i, sub = [], [], [], [], [], []
    pimg, cimg = widen(parents.images), widen(out.images)
    for i in np.nonzero(split)[0]:
        rows_c.append(parents.center[i]); rows_n.append(parents.normal[i]); rows_s.append(parents.scale[i])
        rows_m.append(parents.n_images[i]); rows_i.append(pimg[i]); sub.append(1)
        for k in np.nonzero(children[i])[0]:
            t = 4 * i + k
            rows_c.append(out.center[t]); rows_n.append(out.normal[t]); rows_s.append(out.scale[t])
            rows_m.append(out.n_images[t]); rows_i.append(cimg[t]); sub.append(0)
    if sub:
        ops = api.Batch(np.array(rows_c), np.array(rows_n), np.array(rows_s), np.array(rows_m), np.array(rows_i))
        ops.ok[:] = 1
        api.depth_ops_batch(scene, ops, np.array(sub, np.uint8))
    return BranchResult(out, support, split.astype(np.uint8), children)
