"""What BASELINE configs[3] (the C2 batch of 1e6 seeds cut across N GPUs + one all-gather) should give at N = 2, 4, 8,
projected from ONE GPU: bench.py's own N > 1 leg (HPMVS_BENCH_FORCE_DIST=1: refine the shard, hpmvs_pack_records, a 1-rank
RCCL all-gather) on a shard of 1e6 / N seeds.  The all-gather of the real job moves N times the records of a shard; its
time is estimated from the byte count and xGMI's per-link rate (7 links x ~153 GB/s per GPU, point to point:
/opt/skills/guides/MI355X_MICROARCH.md) and listed separately, so that the first SCALE run of the driver can be compared
with every column.

    python tools/strong_scaling_projection.py [steps]     -> JSON on stdout (profiles/r04/strong_scaling_projection.json)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = sys.argv[1] if len(sys.argv) > 1 else "5"
rows = []
for n_gpus in (1, 2, 4, 8):
    shard = 1_000_000 // n_gpus
    env = dict(os.environ, HPMVS_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + n_gpus), RANK="0",
               WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c3_50v_4k_1e6_sharded", "--gpus", "1",
                        "--steps", steps, "--warmup", "2", "--no-cpu", "--patches", str(shard)],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=1800)
    if r.returncode != 0:
        print(r.stderr[-2000:])
        sys.exit(1)
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    pr = line["per_rank"]
    # the real all-gather: every GPU receives (N - 1) shards of 192-byte records over its 7 xGMI links (153 GB/s each)
    gathered_bytes = 192.0 * shard * (n_gpus - 1)
    allgather_ms_est = 0.0 if n_gpus == 1 else 0.05 + 1e3 * gathered_bytes / (min(7, n_gpus - 1) * 153e9)
    step_ms = pr["kernel_ms"][0] + pr["pack_plus_allgather_ms"][0] + allgather_ms_est
    rows.append({"n_gpus": n_gpus, "seeds_per_gpu": shard, "kernel": line["roofline"]["kernel"],
                 "kernel_ms": pr["kernel_ms"][0], "pack_plus_1rank_allgather_ms": pr["pack_plus_allgather_ms"][0],
                 "allgather_ms_estimated_for_n": allgather_ms_est, "step_ms_projected": step_ms,
                 "wall_ms_per_step_measured_1gpu_shard": line["ms_per_step"]})
t1 = rows[0]["step_ms_projected"]
for r_ in rows:
    r_["patches_per_s_projected"] = 1e6 / (r_["step_ms_projected"] * 1e-3)
    r_["efficiency_projected"] = t1 / (r_["n_gpus"] * r_["step_ms_projected"])
print(json.dumps({"what": "projection of BASELINE configs[3] (strong scaling of the 1e6-seed C2 batch) from one-GPU shard runs of "
                          "bench.py's own N > 1 leg; no multi-GPU hardware was available to the builder", "rows": rows}, indent=1))
