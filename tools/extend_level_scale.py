"""One priority level of CellProcessor::extend at production size: the C++ host layer's PatchOptimizer::extendLevel (one batched
refinement + conflict-free waves, tests/native/bench_extend_level) against the oracle's TRUE sequential loop (orc_extend_round,
live maps; reference CellProcessor.cpp:84-178 leaf by leaf), every candidate's fate compared.

    python tools/extend_level_scale.py [views w h seeds leaves levels [start_level]]      (default: 12 1920 1080 100000 16384 1 4)
(a lower START_LEVEL makes the patches -- and the leaves -- smaller: the same number of leaves covers less of the surface, fewer
candidates find their leaf taken and more of them are refined and interact through the depth maps)

Compared per candidate: stage (the C++ result folds the refinement / scale / drift failures into 1), the three depth-gate counts at
decision time, centre and normal of every candidate that was refined (bit for bit), the accepted set.  Writes one JSON line.
"""
import json
import os
import struct
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hpmvs_amd import api, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

V, W_, H_, NSEEDS, NLEAVES, LEVELS = (int(a) for a in sys.argv[1:7]) if len(sys.argv) > 6 else (12, 1920, 1080, 100000, 16384, 1)
SL = int(sys.argv[7]) if len(sys.argv) > 7 else 4
MARGIN = 1.0
dev = torch.device("cuda", 0)
scene = synth.make_scene(V, W_, H_, n_waves=24, device=dev)
g = api.Scene(scene)
seeds = synth.make_seeds(scene, NSEEDS, start_level=SL, max_images=min(V, api.MAX_IMAGES))
b0 = api.Batch.from_seeds(seeds)
api.optimize_batch(g, b0)
keep = np.nonzero(b0.ok)[0][:NLEAVES]
leaves = api.Batch(b0.center[keep], b0.normal[keep], b0.scale[keep], b0.n_images[keep], b0.images[keep])
width = float(np.float32(2.0 * np.median(leaves.scale) / 0.9))
g.close()

# ---- the C++ host layer on its own scene object (what a C++ scheduler links)
tmp = os.environ.get("TMPDIR", "/tmp")
dump, outp = os.path.join(tmp, "extend_level_scene.bin"), os.path.join(tmp, "extend_level_out.bin")
with open(dump, "wb") as f:
    f.write(struct.pack("i", scene.n_views))
    for v in scene.views:
        f.write(struct.pack("iid4d3d", v.width, v.height, v.f, *v.q, *v.c))
        rgb = v.rgb.cpu().numpy() if hasattr(v.rgb, "cpu") else v.rgb
        f.write(np.ascontiguousarray(rgb, dtype=np.uint8).tobytes())
    for lst in scene.covis:
        f.write(struct.pack("i", len(lst)) + struct.pack(f"{len(lst)}i", *lst))
    f.write(struct.pack("i", leaves.n))
    for k in range(leaves.n):
        m = int(leaves.n_images[k])
        f.write(leaves.center[k].astype(np.float32).tobytes() + leaves.normal[k].astype(np.float32).tobytes())
        f.write(struct.pack("fi", float(leaves.scale[k]), m) + struct.pack(f"{m}i", *leaves.images[k, :m]))
    f.write(struct.pack("f", width))
exe = os.path.join(ROOT, "tests", "native", "bench_extend_level")
t0 = time.perf_counter()
r = subprocess.run([exe, dump, outp, str(LEVELS), "0"], capture_output=True, text=True, env=dict(os.environ, HPMVS_LEVEL_TIMES="1"))
t_proc = time.perf_counter() - t0
if r.returncode != 0:
    raise SystemExit(f"bench_extend_level failed: {r.stdout}\n{r.stderr}")
print(r.stdout.strip())
print("\n".join(l for l in r.stderr.splitlines() if l.startswith("extendLevel")))

cand_t = np.dtype([("stage", "<i4"), ("counts", "<i4", 3), ("center", "<f4", 3), ("normal", "<f4", 3), ("scale", "<f4"), ("n_images", "<i4")])
gpu_levels = []
with open(outp, "rb") as f:
    for _ in range(LEVELS):
        head = f.read(32)
        if len(head) < 32:
            break
        n, waves, nacc, w_, secs, _ = struct.unpack("<iiifdd", head)
        gpu_levels.append((n, waves, nacc, secs, np.frombuffer(f.read(cand_t.itemsize * 6 * n), dtype=cand_t)))

# ---- the oracle's true sequential loop on the same leaves
osc = orc.OracleScene(scene)
OD = orc.OracleDepths(osc)


def oracle_patches(par):
    P = (orc.Patch * par.n)()
    for k in range(par.n):
        P[k].center[:] = par.center[k].tolist()
        P[k].normal[:] = par.normal[k].tolist()
        P[k].scale = float(par.scale[k])
        P[k].n_images = int(par.n_images[k])
        for j in range(par.n_images[k]):
            P[k].images[j] = int(par.images[k, j])
    return P


leaves_cpu = oracle_patches(leaves)
occ = {orc.cell_key(leaves.center[k], width) for k in range(leaves.n)}
for k in range(leaves.n):
    OD.set_depths(leaves_cpu[k])
u32 = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
rec = {"scene": f"{V} x {W_}x{H_}", "start_level": SL, "build": api.lib().hpmvs_build_id().decode(), "width": width, "levels": []}
for lv, (n, waves, nacc, secs, got) in enumerate(gpu_levels):
    assert n == len(leaves_cpu), (n, len(leaves_cpu))
    t0 = time.perf_counter()
    cand, cnt = orc.extend_round(osc, OD, leaves_cpu, width, occ, MARGIN, 0, frozen_gates=False, which=orc.OPT_REF)
    t_cpu = time.perf_counter() - t0
    st = np.array([cand[t].stage for t in range(6 * n)])
    folded = np.where(np.isin(st, (0, 20, 23, 24, 25, 26)), st, 1)
    differing = np.nonzero(folded != got["stage"])[0]
    cnt_diff = np.nonzero((cnt != got["counts"]).any(axis=1))[0]
    refined = np.isin(st, (0, 23, 24, 25, 26))
    cen = np.array([cand[t].center[:3] for t in np.nonzero(refined)[0]], dtype=np.float32).reshape(-1, 3)
    nor = np.array([cand[t].normal[:3] for t in np.nonzero(refined)[0]], dtype=np.float32).reshape(-1, 3)
    geo_diff = int((~(np.all(u32(cen) == u32(got["center"][refined]), axis=1) & np.all(u32(nor) == u32(got["normal"][refined]), axis=1))).sum())
    acc_cpu = [t for t in range(6 * n) if st[t] == 0]
    acc_gpu = np.nonzero(got["stage"] == 0)[0].tolist()
    rec["levels"].append({
        "level": lv, "leaves": n, "candidates": 6 * n, "refined_candidates": int(refined.sum()),
        "accepted_sequential_oracle": len(acc_cpu), "accepted_extendLevel": nacc, "accepted_sets_equal": acc_cpu == acc_gpu,
        "waves": waves, "fates_differing": int(len(differing)), "counts_differing": int(len(cnt_diff)),
        "refined_candidates_whose_centre_or_normal_differs": geo_diff,
        "extendLevel_s": round(secs, 4), "sequential_cpu_oracle_s_1_thread": round(t_cpu, 3),
        "candidates_per_s_extendLevel": round(6 * n / secs), "candidates_per_s_oracle": round(6 * n / t_cpu),
        "stage_histogram": {int(k): int(v) for k, v in zip(*np.unique(folded, return_counts=True))},
        "first_differing": [int(t) for t in differing[:5]],
    })
    leaves_cpu = (orc.Patch * len(acc_cpu))(*[cand[t] for t in acc_cpu])
rec["process_s_incl_scene_upload"] = round(t_proc, 2)
print(json.dumps(rec))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "extend_level_scale.jsonl"), "a") as fh:
    fh.write(json.dumps(rec) + "\n")
bad = [l for l in rec["levels"] if l["fates_differing"] or l["counts_differing"] or not l["accepted_sets_equal"]]
sys.exit(1 if bad else 0)
