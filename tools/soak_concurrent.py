"""Soak: T host threads call hpmvs_optimize_batch on random slices of one seed set -- 1 to 3000 patches, rows of 64 or 120 ids, so that
calls go through the open batch, its stage-100 fallback, small host batches, staged batches and the wide kernel at the same time -- and
every returned patch is held to the one batch over all seeds.   soak_concurrent.py [seconds threads]   (round 6: written after the
workspace-clear bug; profiles/r06/soak_concurrent.txt)"""
import os, sys, threading, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hpmvs_amd import api, synth
SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
T = int(sys.argv[2]) if len(sys.argv) > 2 else 12
scene = synth.make_scene(80, 640, 360, n_waves=16, device=torch.device('cuda', 0))
n = 3000
seeds = synth.make_seeds(scene, n, start_level=1, max_images=256, seed=synth.SEED + 5)
g = api.Scene(scene)
def run(idx, rows):
    b = api.Batch.from_seeds(seeds, np.asarray(idx))
    b = api.Batch(b.center, b.normal, b.scale, b.n_images, b.images[:, :rows])
    api.optimize_batch(g, b)
    return b
full = run(np.arange(n), 120)
print('reference batch: refined', int(full.ok.sum()), 'of', n, '; lists longer than 64:', int((full.n_images[full.ok != 0] > 64).sum()), '; wide', g.last_wide_patches(), flush=True)
stop = time.time() + SECS
bad, calls, patches = [], [0] * T, [0] * T
lock = threading.Lock()
def worker(t):
    rng = np.random.default_rng(100 + t)
    while time.time() < stop:
        m = int(rng.choice([1, 1, 1, 2, 3, 4, 8, 64, 500, 3000], p=[.25, .15, .1, .1, .05, .05, .1, .1, .07, .03]))
        lo = int(rng.integers(0, n - m + 1))
        rows = int(rng.choice([64, 120]))
        if seeds.n_images[lo:lo + m].max() > rows:
            rows = 120
        idx = np.arange(lo, lo + m)
        b = run(idx, rows)
        fits = full.n_images[idx] <= rows
        okf = full.ok[idx] != 0
        same = np.where(okf & fits, (b.ok != 0) & np.all(b.center == full.center[idx], axis=1) & (b.nevals == full.nevals[idx]) & (b.n_images == full.n_images[idx]),
                        np.where(okf & ~fits, (b.ok == 0) & (b.stage == 100), (b.ok == 0) & (b.stage == full.stage[idx])))
        calls[t] += 1; patches[t] += m
        if not same.all():
            with lock: bad.append((t, lo, m, rows, np.nonzero(~same)[0][:5].tolist()))
# ... and, beside them, ONE thread that runs the same level of CellProcessor::extend and the same level of ::branch over and over on the
# scene's depth maps (frontier.extend_level / branch_level: refinement launches, footprints, gate passes, ordered depth ops): every
# repetition must give the first one's result
from hpmvs_amd import frontier
lv_bad, lv_reps = [], [0]
def level_worker():
    keep = np.nonzero(full.ok)[0][:300]
    leaves = api.Batch(full.center[keep], full.normal[keep], full.scale[keep], full.n_images[keep], full.images[keep])
    width = float(np.float32(2.0 * np.median(leaves.scale) / 0.9))
    occ0 = {frontier.cell_key(leaves.center[k], width) for k in range(leaves.n)}
    first = None
    while time.time() < stop:
        api.depth_reset(g)
        leaves.ok[:] = 1
        api.set_depths_batch(g, leaves)
        occ = set(occ0)
        r = frontier.extend_level(g, leaves, width, occ, 1.0, 0)
        cw = (leaves.scale * np.float32(2.0 / 0.9)).astype(np.float32)
        br = frontier.branch_level(g, leaves, leaves.center[:, :3].copy(), cw, (np.arange(leaves.n) & 1).astype(np.uint8))
        v, b, f = api.depth_gates_batch(g, leaves, 1.0, 0)
        sig = (r.stage.tobytes(), r.counts.tobytes(), tuple(r.accepted), r.waves, br.candidates.stage.tobytes(), br.split.tobytes(), v.tobytes(), b.tobytes(), f.tobytes(),
               r.candidates.center.tobytes(), br.candidates.center.tobytes())
        if first is None: first = sig
        elif sig != first: lv_bad.append(lv_reps[0])
        lv_reps[0] += 1
ths = [threading.Thread(target=worker, args=(t,)) for t in range(T)] + [threading.Thread(target=level_worker)]
[t.start() for t in ths]; [t.join() for t in ths]
print('level thread: repetitions', lv_reps[0], 'differing from the first', len(lv_bad), lv_bad[:5])
bad = bad + [('level', r) for r in lv_bad]
print('threads', T, 'seconds', SECS, 'calls', sum(calls), 'patches', sum(patches), 'calls with a differing patch', len(bad), bad[:8])
sys.exit(1 if bad else 0)
