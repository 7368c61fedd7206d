"""Why does the CPU leg scale so badly?  (256 hardware threads gave 9.2x one core.)

Prints what the container may actually use -- the affinity mask, the cgroup CPU quota, NUMA nodes -- and times the
oracle's OpenMP loop over a fixed sample of a workload's seeds at 1, 2, 4, ... threads, with the thread placement
options that matter on a two-socket box (OMP_PROC_BIND / OMP_PLACES are read by libgomp when the oracle library is
loaded, so each setting runs in its own process).

    python tools/cpu_scaling.py [views w h n_seeds start_level]      (default: the C1 scene, 65536 seeds)
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def host_facts():
    f = {"os_cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us",
              "/sys/fs/cgroup/cpuset.cpus.effective", "/sys/devices/system/node/online"):
        try:
            f[p] = open(p).read().strip()
        except OSError:
            pass
    try:
        info = open("/proc/cpuinfo").read()
        f["model"] = [l.split(":", 1)[1].strip() for l in info.splitlines() if l.startswith("model name")][0]
        f["sockets"] = len({l.split(":", 1)[1].strip() for l in info.splitlines() if l.startswith("physical id")})
        f["cores_per_socket"] = int([l.split(":", 1)[1] for l in info.splitlines() if l.startswith("cpu cores")][0])
    except Exception:
        pass
    try:
        f["loadavg"] = open("/proc/loadavg").read().strip()
    except OSError:
        pass
    return f


def child(args):
    import numpy as np
    if os.environ.get("CPU_SCALING_NO_TORCH") != "1":
        import torch  # noqa: F401  (as in bench.py: torch's bundled libgomp is the OpenMP runtime the oracle then runs on)
    sys.path.insert(0, ROOT)
    from hpmvs_amd import synth
    from oracle import oracle as orc
    views, w, h, n, lvl = args
    scene = synth.make_scene(views, w, h, n_waves=24)
    seeds = synth.make_seeds(scene, n, start_level=lvl, max_images=32)
    osc = orc.OracleScene(scene)
    which = orc.best_optimizer()
    out = []
    threads = [1, 2, 4, 8, 16, 32, 64, 128, 256]
    threads = [t for t in threads if t <= (os.cpu_count() or 1)]
    for t in threads:
        m = min(n, max(2048, 512 * t))
        P = orc.patches_from_seeds(seeds, np.arange(m))
        t0 = time.perf_counter()
        orc.optimize_batch(osc, P, which=which, n_threads=t)
        dt = time.perf_counter() - t0
        out.append({"threads": t, "patches": m, "seconds": dt, "patches_per_s": m / dt})
        print(json.dumps(out[-1]), flush=True)
    r1 = out[0]["patches_per_s"]
    for o in out:
        o["speedup_vs_1"] = o["patches_per_s"] / r1
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child([int(v) for v in sys.argv[2:7]])
        sys.exit(0)
    a = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else [12, 1920, 1080, 65536, 4]
    report = {"host": host_facts(), "workload": dict(zip(("views", "w", "h", "seeds", "start_level"), a)), "runs": {}}
    print(json.dumps(report["host"], indent=1), flush=True)
    settings = {
        "default": {},
        "bind_spread_cores": {"OMP_PROC_BIND": "spread", "OMP_PLACES": "cores"},
        "bind_close_threads": {"OMP_PROC_BIND": "close", "OMP_PLACES": "threads"},
        "system_libgomp_no_torch": {"CPU_SCALING_NO_TORCH": "1"},
    }
    for name, env in settings.items():
        print("==", name, env, flush=True)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + [str(v) for v in a],
                           env=dict(os.environ, **env), capture_output=True, text=True, timeout=1500)
        sys.stdout.write(r.stdout)
        if r.returncode != 0:
            sys.stdout.write(r.stderr[-2000:])
        for line in r.stdout.splitlines():
            if line.startswith("RESULT "):
                report["runs"][name] = json.loads(line[7:])
    report["host_after"] = host_facts()
    print("REPORT " + json.dumps(report))
