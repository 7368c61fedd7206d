"""bench.py's multi-GPU leg on ONE GPU: HPMVS_BENCH_FORCE_DIST=1 makes the single rank initialise RCCL and run the
round exactly as the N-rank job does -- refine the rank's shard (configs[3]: the c2 batch cut with shard_range),
hpmvs_pack_records (the product's packing kernel), ONE all-gather of the 192-byte records -- and --check-gather
compares the gathered records with the un-gathered results byte for byte."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_forced_dist_round_on_one_gpu():
    env = dict(os.environ, HPMVS_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0",
               WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c3_50v_4k_1e6_sharded", "--gpus", "1",
                        "--steps", "2", "--warmup", "1", "--no-cpu", "--patches", "200000", "--check-gather"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["scaling"] == "strong" and line["value"] > 0
    g = line["gather_check"]
    assert g["collective"] == "all_gather_into_tensor over RCCL (1 rank)" and g["records"] == 200000
    assert g["bytes_equal_python_packer"] and g["fields_equal_results"], g
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_c3_forced_dist_n1.json"), "w") as fh:
        fh.write(json.dumps(line) + "\n")
