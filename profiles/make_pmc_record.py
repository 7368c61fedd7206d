#!/usr/bin/env python3
"""Turn one `profiles/collect.sh <tag>` collection into the record bench.py prints as roofline.traffic /
roofline_issue:   make_pmc_record.py gpurun_out/prof_<tag> <tag> [profiles/pmc_traffic.json]

The record carries the library build id printed by the traced bench.py run (hpmvs_build_id()): bench.py
reports the counters only while the loaded library is that build."""
import json
import os
import sys


def counters(d, tag, i, kernel="optimize_kernel"):
    f = os.path.join(d, f"{tag}_pmc{i}_counters.json")
    return json.load(open(f))[kernel]["per_dispatch"] if os.path.exists(f) else {}


def main():
    d, tag = sys.argv[1], sys.argv[2]
    dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_traffic.json")
    line = [l for l in open(os.path.join(d, "bench_trace.log")) if l.startswith("{")][-1]
    bench = json.loads(line)
    c = {}
    for i in range(1, 7):
        c.update(counters(d, tag, i))
    n_xcd = 8
    rec = {
        "seeds_per_gpu": bench["config"]["seeds_per_gpu"],
        "library_build_id": bench.get("library_build_id"),
        "n_cus": bench.get("n_cus", 256),
        "hbm_bytes_per_launch": (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0,
        "fetch_size_kb": c["FETCH_SIZE"], "write_size_kb": c["WRITE_SIZE"],
        "tcc_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
        "kernel_cycles": c["GRBM_GUI_ACTIVE"] / n_xcd,  # the counter is summed over the 8 XCDs
        "valu_thread_cycles": c["SQ_THREAD_CYCLES_VALU"],
        "valu_insts_per_patch": c["SQ_INSTS_VALU"] / bench["config"]["seeds_per_gpu"],
        "valu_lane_utilization": c["SQ_THREAD_CYCLES_VALU"] / c["SQ_INSTS_VALU"] / 64.0,
        "valu_issue_slot_frac": c["SQ_INSTS_VALU"] * 4.0 / (bench.get("n_cus", 256) * 4 * c["GRBM_GUI_ACTIVE"] / n_xcd),
        "vmem_insts_per_patch": c.get("SQ_INSTS_FLAT", 0.0) / bench["config"]["seeds_per_gpu"],
        "lds_insts_per_patch": c.get("SQ_INSTS_LDS", 0.0) / bench["config"]["seeds_per_gpu"],
        "wave_wait_frac": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
        "source": f"profiles/{tag}/{tag}_pmc1..6_counters.json (rocprofv3 --pmc, separate passes; hbm_bytes = (FETCH_SIZE + "
                  "WRITE_SIZE) * 1024, calibration in profiles/calibration/; lane-cycles = SQ_THREAD_CYCLES_VALU against "
                  "CUs x 4 SIMDs x 16 lanes x GRBM_GUI_ACTIVE / 8)",
    }
    allrec = json.load(open(dst)) if os.path.exists(dst) else {}
    allrec[bench["config"]["workload"]] = rec
    json.dump(allrec, open(dst, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
