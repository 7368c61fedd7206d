#!/usr/bin/env python3
"""Reduce rocprofv3 CSV output to small, committable summaries.

usage: summarize.py <rocprof_out_dir> <out_prefix> [kernel-substring ...]
Writes <out_prefix>_kernel_stats.csv (per-kernel count/total/avg/min/max of every kernel whose
name matches one of the substrings, plus the top 15 kernels by total time) and
<out_prefix>_counters.json (per-kernel sums of every collected PMC counter, averaged per dispatch).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def main():
    d, prefix = sys.argv[1], sys.argv[2]
    pats = sys.argv[3:] or ["optimize_kernel", "stream_kernel", "half_resize_kernel", "objective_kernel", "inccs_kernel"]
    traces = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    stats = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    for f in traces:
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name") or r.get("kernel_name") or ""
            t = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3  # us
            s = stats[name]
            s[0] += 1; s[1] += t; s[2] = min(s[2], t); s[3] = max(s[3], t)
    if stats:
        rows = sorted(stats.items(), key=lambda kv: -kv[1][1])
        total = sum(v[1] for v in stats.values())
        with open(prefix + "_kernel_stats.csv", "w") as fh:
            fh.write("kernel,calls,total_us,avg_us,min_us,max_us,percent\n")
            keep = [kv for kv in rows if any(p in kv[0] for p in pats)] + rows[:15]
            seen = set()
            for name, (c, tot, mn, mx) in keep:
                if name in seen:
                    continue
                seen.add(name)
                fh.write(f"\"{name[:120]}\",{c},{tot:.1f},{tot / c:.1f},{mn:.1f},{mx:.1f},{100 * tot / total:.2f}\n")
    counters = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    agg = defaultdict(lambda: defaultdict(float))
    ndisp = defaultdict(set)
    for f in counters:
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name") or ""
            if not any(p in name for p in pats):
                continue
            key = next(p for p in pats if p in name)
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
            ndisp[key].add(r.get("Dispatch_Id"))
    if agg:
        out = {}
        for k, cs in agg.items():
            n = max(1, len(ndisp[k]))
            out[k] = {"dispatches": n, "per_dispatch": {c: v / n for c, v in cs.items()}}
        json.dump(out, open(prefix + "_counters.json", "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
