"""One priority level of CellProcessor::branch at production size: hpmvs_amd.frontier.branch_level (one level-support call, one
batched refinement of the 4 n children, one ordered depth-ops call) against the oracle's leaf-by-leaf loop (orc_branch_round;
reference CellProcessor.cpp:210-307), every child, split decision and map cell compared.

    python tools/branch_level_scale.py [views w h seeds leaves [start_level]]      (default: 12 1920 1080 100000 16384 3)
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hpmvs_amd import api, frontier, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

V, W_, H_, NSEEDS, NLEAVES = (int(a) for a in sys.argv[1:6]) if len(sys.argv) > 5 else (12, 1920, 1080, 100000, 16384)
SL = int(sys.argv[6]) if len(sys.argv) > 6 else 3
dev = torch.device("cuda", 0)
scene = synth.make_scene(V, W_, H_, n_waves=24, device=dev)
g = api.Scene(scene)
seeds = synth.make_seeds(scene, NSEEDS, start_level=SL, max_images=min(V, api.MAX_IMAGES))
b0 = api.Batch.from_seeds(seeds)
api.optimize_batch(g, b0)
keep = np.nonzero(b0.ok)[0][:NLEAVES]
R = api.Batch(b0.center[keep], b0.normal[keep], b0.scale[keep], b0.n_images[keep], b0.images[keep])
n = R.n
rng = np.random.default_rng(3)
R.scale[rng.random(n) < 0.1] *= np.float32(0.1)          # some leaves without level support
width = (R.scale * np.float32(2.0 / 0.9)).astype(np.float32)
cc = R.center[:, :3].copy()
final = (rng.random(n) < 0.5).astype(np.uint8)
api.depth_reset(g)
R.ok[:] = 1
api.set_depths_batch(g, R)

osc = orc.OracleScene(scene)
P = (orc.Patch * n)()
for k in range(n):
    P[k].center[:] = R.center[k].tolist(); P[k].normal[:] = R.normal[k].tolist()
    P[k].scale = float(R.scale[k]); P[k].n_images = int(R.n_images[k])
    for j in range(R.n_images[k]):
        P[k].images[j] = int(R.images[k, j])

t0 = time.perf_counter()
res = frontier.branch_level(g, R, cc, width, final)
t_gpu = time.perf_counter() - t0
u32 = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
n_levels = osc.camera(0).n_levels


def against_the_loop():
    """the oracle's leaf-by-leaf loop on fresh maps; what differs from the batched level"""
    OD = orc.OracleDepths(osc)
    for k in range(n):
        OD.set_depths(P[k])
    t0 = time.perf_counter()
    cand, split = orc.branch_round(osc, OD, P, cc, width, final, which=orc.OPT_REF)
    t_cpu = time.perf_counter() - t0
    st = np.array([cand[t].stage for t in range(4 * n)])
    st20 = np.where(st == 27, 20, st)
    ch = np.nonzero(st == 0)[0]
    cen = np.array([cand[t].center[:3] for t in ch], dtype=np.float32).reshape(-1, 3)
    nor = np.array([cand[t].normal[:3] for t in ch], dtype=np.float32).reshape(-1, 3)
    geo = int((~(np.all(u32(cen) == u32(res.candidates.center[ch, :3]), axis=1) & np.all(u32(nor) == u32(res.candidates.normal[ch, :3]), axis=1))).sum())
    cells = 0
    for v in range(scene.n_views):
        for l in range(n_levels):
            cells += int((api.depth_level(g, v, l) != OD.level(v, l)).sum())
    return dict(st20=st20, ch=ch, split=split, geo=geo, cells=cells, t_cpu=t_cpu,
                stages=int((st20 != res.candidates.stage).sum()), splits=int((split != res.split).sum()))


A = against_the_loop()
st20, ch, split, geo_diff, cells_diff, t_cpu = A["st20"], A["ch"], A["split"], A["geo"], A["cells"], A["t_cpu"]
attributed = None
if A["stages"] or A["splits"] or geo_diff or cells_diff:
    # the one known source of differences: start points whose acos() this host's libm does not round correctly
    orc.set_libm_mode(1)
    B = against_the_loop()
    orc.set_libm_mode(0)
    attributed = {"with_a_correctly_rounded_acos_in_the_oracle": {"stages_differing": B["stages"], "split_decisions_differing": B["splits"],
                  "children_whose_centre_or_normal_differs": B["geo"], "map_cells_differing": B["cells"]}}
rec = {"scene": f"{V} x {W_}x{H_}", "start_level": SL, "build": api.lib().hpmvs_build_id().decode(), "leaves": n, "children_built": int((st20 != 20).sum()),
       "children_kept": int(len(ch)), "leaves_split": int(split.sum()), "leaves_without_level_support": int((res.support < 1).sum()),
       "stages_differing": A["stages"], "split_decisions_differing": A["splits"],
       "children_whose_centre_or_normal_differs": geo_diff, "map_cells_differing": cells_diff,
       "branch_level_s": round(t_gpu, 4), "sequential_cpu_oracle_s_1_thread": round(t_cpu, 3),
       "children_per_s_branch_level": round(4 * n / t_gpu), "children_per_s_oracle": round(4 * n / t_cpu)}
if attributed:
    rec.update(attributed)
print(json.dumps(rec))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "branch_level_scale.jsonl"), "a") as fh:
    fh.write(json.dumps(rec) + "\n")
left = attributed["with_a_correctly_rounded_acos_in_the_oracle"] if attributed else None
sys.exit(1 if (left and any(left.values())) else 0)
