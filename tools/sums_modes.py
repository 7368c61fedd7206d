"""What does bit-identity cost, and what would the tolerance north_star asks for buy?  

Runs the refinement kernel in its two window-sum modes -- `exact` (the reference's sequential order, the default and what
every parity test runs) and `tree` (HPMVS_SUMS=tree: row sums first, kernel_optimize_tree.hip) -- on a workload, times
both, and compares each with the CPU oracle on the first `m` seeds: decisions, success-set Jaccard, bit-identical
patches, mean |d ncc|, same-basin share.  For scale, the oracle's own Eigen-order variants (SSE2 / SSE3 reductions,
Eigen-3.0 reciprocal division: profiles/r03/eigen_variant_sensitivity.json) are the spread "the reference" itself has.

    python tools/sums_modes.py [views w h n m]        (default: the C1 scene; C2: 50 3840 2160 1000000 32768)
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(args, out):
    import torch
    from hpmvs_amd import api, synth
    V, W, H, N, M = args
    dev = torch.device("cuda", 0)
    scene = synth.make_scene(V, W, H, n_waves=24, device=dev)
    g = api.Scene(scene)
    seeds = synth.make_seeds(scene, N, start_level=4, max_images=32)
    ms = []
    for _ in range(3):
        b = api.Batch.from_seeds(seeds)
        api.optimize_batch(g, b)
        ms.append(g.last_kernel_ms())
    np.savez(out, ok=b.ok[:M], center=b.center[:M], normal=b.normal[:M], fmin=b.fmin[:M], nevals=b.nevals[:M],
             stage=b.stage[:M], n_images=b.n_images[:M], images=b.images[:M], color=b.color[:M], ms=np.array(ms),
             ok_all=np.array([b.ok.mean()]), evals_all=np.array([b.nevals[b.nevals > 0].mean()]))


def compare(P, g, scale):
    okc = np.array([p.stage == 0 for p in P])
    okg = g["ok"].astype(bool)
    both = np.nonzero(okc & okg)[0]
    unrob = lambda f: f / (1.0 - 3.0 * f)
    dncc, basin, ident = [], [], []
    for k in both:
        p = P[k]
        ident.append(np.array_equal(np.array(p.center[:], dtype=np.float32), g["center"][k]) and
                     np.array_equal(np.array(p.normal[:], dtype=np.float32), g["normal"][k]) and
                     p.nevals == g["nevals"][k] and list(p.images[:p.n_images]) == list(g["images"][k, :g["n_images"][k]]))
        dncc.append(abs(unrob(p.fmin) - unrob(g["fmin"][k])))
        dd = np.linalg.norm(np.array(p.center[:3]) - g["center"][k, :3]) / float(scale[k])
        na, nb = np.array(p.normal[:3], dtype=np.float64), g["normal"][k, :3].astype(np.float64)
        basin.append(dd < 0.25 and np.degrees(np.arctan2(np.linalg.norm(np.cross(na, nb)), np.dot(na, nb))) < 2.0)
    return {"sample": len(okc), "decision_agreement": float((okc == okg).mean()),
            "success_set_jaccard": float((okc & okg).sum() / max(1, (okc | okg).sum())),
            "stage_agreement": float((np.array([p.stage for p in P]) == g["stage"]).mean()),
            "bit_identical_patch_fraction": float(np.mean(ident)) if ident else None,
            "mean_abs_ncc_delta": float(np.mean(dncc)) if dncc else None,
            "p95_abs_ncc_delta": float(np.percentile(dncc, 95)) if dncc else None,
            "same_basin_fraction": float(np.mean(basin)) if basin else None}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child([int(v) for v in sys.argv[2:7]], sys.argv[7])
        sys.exit(0)
    a = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else [12, 1920, 1080, 100000, 32768]
    from hpmvs_amd import synth
    from oracle import oracle as orc
    V, W, H, N, M = a
    import torch
    # the oracle's copy of the same deterministic scene (rendered on the GPU like the children's, as bench.py does)
    scene = synth.make_scene(V, W, H, n_waves=24, device=torch.device("cuda", 0) if torch.cuda.is_available() else None)
    seeds = synth.make_seeds(scene, N, start_level=4, max_images=32)
    osc = orc.OracleScene(scene)
    P = orc.patches_from_seeds(seeds, np.arange(M))
    orc.optimize_batch(osc, P, which=orc.best_optimizer(), n_threads=16)
    rep = {"workload": dict(zip(("views", "w", "h", "seeds", "oracle_sample"), a)), "modes": {}}
    for mode in ("exact", "tree"):
        out = f"/tmp/sums_{mode}.npz"
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + [str(v) for v in a] + [out],
                           env=dict(os.environ, HPMVS_SUMS=mode), capture_output=True, text=True, timeout=3000)
        if r.returncode != 0:
            print(r.stderr[-3000:])
            sys.exit(1)
        g = np.load(out)
        rec = compare(P, g, seeds.scale)
        rec["kernel_ms"] = [float(v) for v in g["ms"]]
        rec["kernel_ms_min"] = float(g["ms"].min())
        rec["success_fraction_full_batch"] = float(g["ok_all"][0])
        rec["mean_evals_full_batch"] = float(g["evals_all"][0])
        rep["modes"][mode] = rec
        print(mode, json.dumps(rec), flush=True)
    e, t = rep["modes"]["exact"], rep["modes"]["tree"]
    rep["tree_vs_exact_time"] = t["kernel_ms_min"] / e["kernel_ms_min"]
    print("REPORT " + json.dumps(rep))
