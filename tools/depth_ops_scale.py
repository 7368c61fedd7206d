"""hpmvs_depth_ops_batch at scale: a long random sequence of Scene::setDepths(patch, subtract) calls (reference Scene.cpp:351-381) --
patches entered, taken back, entered again, and copies pushed along a viewing ray so that different depths meet in one cell --
as ONE call, against the loop that makes the calls one after the other (the oracle); every cell of every map compared.

    python tools/depth_ops_scale.py [views w h seeds calls]      (default: 50 3840 2160 300000 1000000)
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hpmvs_amd import api, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

V, W_, H_, NSEEDS, NCALLS = (int(a) for a in sys.argv[1:6]) if len(sys.argv) > 5 else (50, 3840, 2160, 300000, 1000000)
dev = torch.device("cuda", 0)
scene = synth.make_scene(V, W_, H_, n_waves=24, device=dev)
g = api.Scene(scene)
seeds = synth.make_seeds(scene, NSEEDS, start_level=3, max_images=min(V, api.MAX_IMAGES))
b0 = api.Batch.from_seeds(seeds)
api.optimize_batch(g, b0)
keep = np.nonzero(b0.ok)[0]
R = api.Batch(b0.center[keep], b0.normal[keep], b0.scale[keep], b0.n_images[keep], b0.images[keep])
n0 = R.n
rng = np.random.default_rng(5)
cams = np.array([scene.views[int(R.images[k, 0])].c for k in range(n0)], dtype=np.float64)
fac = rng.choice([0.999, 0.9995, 1.0005, 1.001], size=n0)
center2 = R.center.copy()
center2[:, :3] = (cams + (R.center[:, :3].astype(np.float64) - cams) * fac[:, None]).astype(np.float32)
pool = dict(c=np.concatenate([R.center, center2]), n=np.concatenate([R.normal, R.normal]), s=np.concatenate([R.scale, R.scale]),
            m=np.concatenate([R.n_images, R.n_images]), i=np.concatenate([R.images, R.images]))
idx = np.concatenate([np.arange(n0), rng.integers(0, 2 * n0, size=max(0, NCALLS - n0))])
sub = np.zeros(len(idx), np.uint8)
sub[n0:] = (rng.random(len(idx) - n0) < 0.4).astype(np.uint8)
ops = api.Batch(pool["c"][idx], pool["n"][idx], pool["s"][idx], pool["m"][idx], pool["i"][idx])
ops.ok[:] = 1
api.depth_reset(g)
api.depth_ops_batch(g, ops, sub)   # (warm-up: allocations)
api.depth_reset(g)
t0 = time.perf_counter()
api.depth_ops_batch(g, ops, sub)
t_gpu = time.perf_counter() - t0

osc = orc.OracleScene(scene)
OD = orc.OracleDepths(osc)
P = (orc.Patch * (2 * n0))()
for k in range(2 * n0):
    P[k].center[:] = pool["c"][k].tolist(); P[k].scale = float(pool["s"][k]); P[k].n_images = int(pool["m"][k])
    for j in range(int(pool["m"][k])):
        P[k].images[j] = int(pool["i"][k, j])
t0 = time.perf_counter()
for t in range(len(idx)):
    OD.set_depths(P[int(idx[t])], subtract=bool(sub[t]))
t_cpu = time.perf_counter() - t0
n_levels = osc.camera(0).n_levels
cells, differing, filled = 0, 0, 0
for v in range(scene.n_views):
    for l in range(n_levels):
        a, b = api.depth_level(g, v, l), OD.level(v, l)
        cells += a.size; differing += int((a != b).sum()); filled += int((b < 1000.0).sum())
rec = {"scene": f"{V} x {W_}x{H_}", "build": api.lib().hpmvs_build_id().decode(), "patches": int(n0), "calls": int(len(idx)),
       "subtractions": int(sub.sum()), "cell_writes": int(ops.n_images.sum()), "map_cells": cells, "cells_holding_a_depth": filled,
       "cells_differing": differing, "depth_ops_batch_s_host_arrays": round(t_gpu, 4),
       "oracle_loop_s_incl_python_call_overhead": round(t_cpu, 2)}
print(json.dumps(rec))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "depth_ops_scale.jsonl"), "a") as fh:
    fh.write(json.dumps(rec) + "\n")
sys.exit(1 if differing else 0)
