#!/usr/bin/env python3
"""bench.py -- refined patches/sec of the MI355X-native HPMVS refinement path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one expansion round: every rank runs the full optimize() (image selection, BOBYQA
refinement of depth + normal against NCC over the attached views, reference-image selection) on
its shard of synthetic seed patches, all inputs already resident in HBM, and -- for N > 1 -- one
RCCL all-gather of the fixed-size refined-patch records (north_star's exchange step).

Default workload: at N = 1 the configuration BASELINE.json's metric is quoted on (configs[2], "c2": 50
synthetic 3840x2160 views, 1e6 seed patches); at N > 1 BASELINE configs[3] ("c3"): THE SAME batch of
1e6 seeds, identical on every rank, cut with `shard_range` (reference Scene.cpp:94-96,114: the OpenMP
loop over one seed vector; the round structure is main.cpp:145-181) and the refined records
all-gathered, so every rank ends the step holding the full refined set -- `scaling: "strong"`, and
N = 1 of that is exactly c2.  `--workload c2_50v_4k_1e6 --scaling weak` with N > 1 gives every rank
its own 1e6 seeds instead.  The N > 1 line also carries per-rank kernel and exchange times
(`per_rank`), so that a scaling curve can be read against the one-GPU shard timings of DESIGN section 7.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline_issue : the compute-side view of the same kernel (it is instruction-issue / latency bound, not HBM
                 bound): VALU lane-cycles used / the machine's lane-cycles in the kernel's time, from the PMC
                 pass recorded in profiles/pmc_traffic.json -- only when that record carries the build id of
                 the library that is loaded (a kernel change without a new PMC pass prints no stale counters);
  roofline     : algorithmic image bytes of the refinement kernel (588 B per sampleTexture that
                 passed its gates, counted by the kernel itself) / kernel time from HIP events on
                 the launch stream, against the 8 TB/s HBM peak;
  cpu_baseline : the CPU oracle (restated photometric path around the reference's genuine NLopt
                 BOBYQA when oracle/_ref travelled) on a bounded sample of the same seeds, all
                 host cores;
  parity       : GPU vs that CPU run on the same sample (the metric's "mean NCC-score delta").
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from hpmvs_amd import api, synth  # noqa: E402
from hpmvs_amd import distributed as dd  # noqa: E402

WORKLOADS = {
    # name: (views, width, height, seeds [per GPU when weak, in total when strong], start_level, default scaling)
    "c0_tiny": (3, 640, 480, 200, 2, "weak"),
    "c1_12v_1080p_1e5": (12, 1920, 1080, 100_000, 4, "weak"),
    "c2_50v_4k_1e6": (50, 3840, 2160, 1_000_000, 4, "weak"),               # BASELINE configs[2]: the headline
    "c3_50v_4k_1e6_sharded": (50, 3840, 2160, 1_000_000, 4, "strong"),     # configs[3]: c2's batch cut across the ranks
    "c4_200v_4k_1e7": (200, 3840, 2160, 10_000_000, 4, "strong"),          # configs[4]: 6.6 GB of pyramids per GPU
    # stress: START_LEVEL 1 puts the 7x7 windows in pyramid levels 0-1 (1.55 GB for 50 views, far beyond
    # the 256 MB MALL), the case in which image bytes really come from HBM
    "s1_50v_4k_1e6_level1": (50, 3840, 2160, 1_000_000, 1, "weak"),
}
DEFAULT_WORKLOAD = "c2_50v_4k_1e6"              # N = 1
DEFAULT_WORKLOAD_MULTI = "c3_50v_4k_1e6_sharded"  # N > 1: the same batch cut across the ranks (BASELINE configs[3])
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
RECORD_BYTES = dd.RECORD_BYTES  # all-gathered refined-patch record


def device_batch(seeds, dev, max_images):
    """SoA batch resident in HBM (torch tensors are just the allocation + stream plumbing)."""
    t = {}
    t["center"] = torch.from_numpy(seeds.center).to(dev)
    t["normal"] = torch.from_numpy(seeds.normal).to(dev)
    t["scale"] = torch.from_numpy(seeds.scale).to(dev)
    t["n_images"] = torch.from_numpy(seeds.n_images).to(dev)
    t["images"] = torch.from_numpy(np.ascontiguousarray(seeds.images[:, :max_images])).to(dev)
    n = len(seeds.scale)
    t["ok"] = torch.zeros(n, dtype=torch.uint8, device=dev)
    t["color"] = torch.zeros((n, 3), dtype=torch.float32, device=dev)
    t["ncc"] = torch.zeros(n, dtype=torch.float32, device=dev)
    t["fmin"] = torch.zeros(n, dtype=torch.float64, device=dev)
    t["x"] = torch.zeros((n, 3), dtype=torch.float64, device=dev)
    t["result"] = torch.zeros(n, dtype=torch.int32, device=dev)
    t["nevals"] = torch.zeros(n, dtype=torch.int32, device=dev)
    t["stage"] = torch.zeros(n, dtype=torch.int32, device=dev)
    t["ngrabs"] = torch.zeros(n, dtype=torch.int32, device=dev)
    return t


def c_batch(t, n, max_images):
    b = api.PatchBatch()
    b.n, b.max_images = n, max_images
    for k in ("center", "normal", "scale", "n_images", "images", "ok", "color", "ncc", "fmin", "x", "result",
              "nevals", "stage", "ngrabs"):
        setattr(b, k, t[k].data_ptr())
    return b


def check_record_ranges(t):
    """The record stores counts and ids as uint16: validated once in the warm-up steps (a device sync), not in
    the timed ones.  (dd.pack_records(check=True) is the same test followed by the torch packer.)"""
    dd.pack_records(t["center"][:1], t["normal"][:1], t["color"][:1], t["scale"][:1], t["fmin"][:1], t["ok"][:1],
                    t["n_images"][:1], t["images"][:1], check=False)  # layout / column-count errors raise here
    nmax = int(t["n_images"].max())
    if nmax > t["images"].shape[1] or nmax > dd.MAX_RECORD_IMAGES:
        raise ValueError(f"a patch has {nmax} images but only {min(t['images'].shape[1], dd.MAX_RECORD_IMAGES)} ids can travel")


class Plan:
    """What this rank refines in one step.  Pure arithmetic on (workload, scaling, patches, rank, world): covered at
    world size 2 by tests/test_cpu_distributed.py without a GPU."""

    def __init__(self, workload, scaling, patches, rank, world):
        self.workload = workload
        self.views, self.W, self.H, n_total, self.start_level, default_scaling = WORKLOADS[workload]
        self.scaling = scaling or default_scaling
        if patches > 0:
            n_total = patches
        self.n_total, self.rank, self.world = n_total, rank, world
        if self.scaling == "strong":
            # one batch, the same on every rank; rank r refines [lo, hi) of it
            self.lo, self.hi = dd.shard_range(n_total, rank, world)
            self.counts = [dd.shard_range(n_total, r, world)[1] - dd.shard_range(n_total, r, world)[0] for r in range(world)]
            self.seed_value = synth.SEED
            self.n_job = n_total
        else:
            self.lo, self.hi = 0, n_total
            self.counts = [n_total] * world
            self.seed_value = synth.SEED + 977 * rank
            self.n_job = world * n_total
        self.n_seeds = self.hi - self.lo  # patches this rank refines per step

    def my_slice(self, gathered):
        """This rank's own records inside the gathered round."""
        if self.scaling == "strong":
            return gathered[self.lo:self.hi]
        off = sum(self.counts[:self.rank])
        return gathered[off:off + self.n_seeds]


def default_workload(world):
    return DEFAULT_WORKLOAD if world <= 1 else DEFAULT_WORKLOAD_MULTI


class GpuBackend:
    """The product path: HBM-resident scene, hpmvs_optimize_batch + hpmvs_pack_records through the C ABI, RCCL."""
    name = "hip"
    dist_backend = "nccl"

    def __init__(self, local_rank):
        if not torch.cuda.is_available() or api.device_count() < 1:
            raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
        torch.cuda.set_device(local_rank)
        self.local_rank = local_rank
        self.device = torch.device("cuda", local_rank)
        self.L = api.lib()
        self.opts = api.default_options()

    def init_dist(self):
        dist.init_process_group("nccl", device_id=self.device)  # RCCL over xGMI

    def make_scene(self, plan):
        scene = synth.make_scene(plan.views, plan.W, plan.H, n_waves=24, device=self.device)
        self.gpu_scene = api.Scene(scene, device=self.local_rank)
        torch.cuda.synchronize()
        return scene

    def stream(self):
        return torch.cuda.current_stream().cuda_stream

    def mark(self):
        """A point in time on the stream the kernels are launched on (HIP event)."""
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    @staticmethod
    def ms_between(a, b):
        return a.elapsed_time(b)

    def refine(self, work, n, max_images):
        b = c_batch(work, n, max_images)
        rc = self.L.hpmvs_optimize_batch(self.gpu_scene.h, C.byref(self.opts), C.byref(b), 1, C.c_void_p(self.stream()))
        if rc != 0:
            raise RuntimeError(self.L.hpmvs_last_error().decode())

    def pack(self, work, n, max_images, records):
        # the product path of INTEGRATION.md: hpmvs_pack_records (kernel) fills the round's exchange buffer
        b = c_batch(work, n, max_images)
        rc = self.L.hpmvs_pack_records(C.byref(b), C.c_void_p(records.data_ptr()), C.c_void_p(self.stream()))
        if rc != 0:
            raise RuntimeError(self.L.hpmvs_last_error().decode())

    def pack_tails(self, work, n, max_images):
        """The tails of the lists longer than a record's 64 ids (rows wider than 64 only: a 200-view scene); a device tensor."""
        if max_images <= dd.MAX_RECORD_IMAGES:
            return torch.zeros((0, dd.TAIL_BYTES), dtype=torch.uint8, device=self.device)
        if getattr(self, "_tails", None) is None or self._tails.shape[0] < n:
            self._tails = torch.zeros((n, dd.TAIL_BYTES), dtype=torch.uint8, device=self.device)
        b = c_batch(work, n, max_images)
        nt = C.c_int32(0)
        self.L.hpmvs_pack_record_tails.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_void_p]
        rc = self.L.hpmvs_pack_record_tails(C.byref(b), C.c_void_p(self._tails.data_ptr()), n, C.byref(nt), C.c_void_p(self.stream()))
        if rc != 0:
            raise RuntimeError(self.L.hpmvs_last_error().decode())
        return self._tails[:nt.value]

    def sync(self):
        torch.cuda.synchronize()


def main(argv=None, backend=None):
    """`backend` (tests only): an object with GpuBackend's interface; tests/test_cpu_distributed.py runs this very
    function at world size 2 on gloo with the kernel call stubbed, to execute the shard / gather / timing logic."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=None, choices=list(WORKLOADS),
                    help=f"default: {DEFAULT_WORKLOAD} at N = 1, {DEFAULT_WORKLOAD_MULTI} (the same batch, sharded) at N > 1")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="weak: seeds per GPU fixed; strong: one batch cut across the ranks (default: the workload's)")
    ap.add_argument("--patches", type=int, default=0, help="override the workload's seed count")
    ap.add_argument("--order", default="as-is", choices=["as-is", "tiles"],
                    help="tiles: host-side bucket sort of the round by (reference image, 256-px tile), SURVEY 8e")
    ap.add_argument("--cpu-sample", type=int, default=32768, help="patches of the parity sample")
    ap.add_argument("--cpu-timing-sample", type=int, default=262144, help="patches of the all-core CPU-baseline run")
    ap.add_argument("--cpu-sample-1core", type=int, default=16384, help="patches of the 1-core CPU run")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--check-gather", action="store_true",
                    help="after the timed steps: compare the gathered records with the un-gathered results (tests)")
    args = ap.parse_args(argv)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    be = backend if backend is not None else GpuBackend(local_rank)
    dev = be.device
    use_dist = world > 1 or os.environ.get("HPMVS_BENCH_FORCE_DIST") == "1"  # the latter: 1-GPU test of the RCCL path
    if use_dist:
        be.init_dist()

    workload = args.workload or default_workload(world)
    plan = Plan(workload, args.scaling, args.patches, rank, world)
    views, W, H, start_level, scaling = plan.views, plan.W, plan.H, plan.start_level, plan.scaling
    n_total, n_seeds, n_job, lo, hi, counts = plan.n_total, plan.n_seeds, plan.n_job, plan.lo, plan.hi, plan.counts
    # rows of the batch's image lists: a list cannot hold more than the scene's views (addImages never attaches a view twice), so
    # rows of min(views, HPMVS_MAX_IMAGES) ids hold every list that can come back -- no patch of any workload ends at stage 100
    # for want of room.  (Lists that outgrow 64 ids on the way are redone by the wide kernel behind the batch kernel.)
    max_images = min(views, api.MAX_IMAGES)

    # ---- scene: rendered on the GPU, replicated on every rank, resident in HBM
    t0 = time.time()
    scene = be.make_scene(plan)
    t_scene = time.time() - t0
    seeds = synth.make_seeds(scene, n_total, start_level=start_level, max_images=max_images, seed=plan.seed_value)
    if scaling == "strong" and world > 1:
        seeds = synth.Seeds(center=seeds.center[lo:hi], normal=seeds.normal[lo:hi], scale=seeds.scale[lo:hi],
                            n_images=seeds.n_images[lo:hi], images=seeds.images[lo:hi], truth=seeds.truth[lo:hi])
    if args.order == "tiles":
        P0 = np.array([list(api.camera_from_nvm(v.f, v.q, v.c, v.width, v.height, scene.max_level).P0) for v in scene.views])
        perm = dd.tile_order(seeds.center, seeds.images, seeds.n_images, P0)
        seeds = synth.Seeds(center=seeds.center[perm], normal=seeds.normal[perm], scale=seeds.scale[perm],
                            n_images=seeds.n_images[perm], images=seeds.images[perm], truth=seeds.truth[perm])
    pristine = device_batch(seeds, dev, max_images)
    work = {k: v.clone() for k, v in pristine.items()}
    gathered = None
    gathered_tails = None
    # the round's exchange buffer: n_seeds records of 192 bytes, filled by the product's own packing kernel
    records = torch.empty((n_seeds, RECORD_BYTES), dtype=torch.uint8, device=dev) if use_dist else None

    marks = []

    def step(timed=False):
        for k in ("center", "normal", "n_images", "images"):
            work[k].copy_(pristine[k])
        # HIP events on the stream the kernel is launched on (torch's current stream is handed to the C ABI)
        m0 = be.mark()
        be.refine(work, n_seeds, max_images)
        m1 = be.mark()
        m2 = None
        if use_dist:
            nonlocal gathered
            if not timed:
                check_record_ranges(work)
            be.pack(work, n_seeds, max_images, records)
            if max_images > dd.MAX_RECORD_IMAGES and hasattr(be, "pack_tails"):
                # rows wider than a record's 64 ids (a 200-view scene): a second all-gather, made only if some list IS longer
                nonlocal gathered_tails
                gathered, gathered_tails = dd.all_gather_round(records, be.pack_tails(work, n_seeds, max_images), counts)
            else:
                gathered = dd.all_gather_records(records, counts)  # ONE collective per round
            m2 = be.mark()
        if timed:
            marks.append((m0, m1, m2))

    def fence():
        be.sync()
        if use_dist:
            dist.barrier()
        be.sync()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(timed=True)
    fence()
    elapsed_rank = time.perf_counter() - t0
    # average launch duration of the refinement kernel over the timed region (and of pack + all-gather behind it)
    last_ms = float(np.mean([be.ms_between(a, b) for a, b, _ in marks]))
    exch_ms = float(np.mean([be.ms_between(b, c) for _, b, c in marks])) if use_dist else 0.0
    lib_ms = be.gpu_scene.last_kernel_ms() if hasattr(be, "gpu_scene") else last_ms  # the library's own event pair (cross-check)
    elapsed = elapsed_rank
    per_rank = None
    if use_dist:
        # the job's time is the slowest rank's; every rank's own figures travel to rank 0 beside it
        mine = torch.tensor([elapsed_rank, last_ms, exch_ms, float(n_seeds)], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        tmax = torch.tensor([elapsed_rank], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        rows = [[float(v) for v in t.cpu()] for t in every]
        # what the collective layer itself saw (the first SCALE record checks itself: ranks, one device per rank, the
        # gathered round has every rank's shard)
        dev_idx = int(dev.index or 0) if dev.type == "cuda" else -1
        seen = torch.tensor([float(rank), float(dev_idx)], dtype=torch.float64, device=dev)
        seen_all = [torch.zeros_like(seen) for _ in range(world)]
        dist.all_gather(seen_all, seen)
        collective = {"backend": dist.get_backend(), "world_size_seen": int(dist.get_world_size()),
                      "ranks_seen": [int(t[0].item()) for t in seen_all], "device_of_rank": [int(t[1].item()) for t in seen_all],
                      "gathered_records": int(gathered.shape[0]) if gathered is not None else 0,
                      "gathered_equals_job": bool(gathered is not None and int(gathered.shape[0]) == (n_total if scaling == "strong" else world * n_total)),
                      "shards_sum_to_job": bool(sum(int(r[3]) for r in rows) == n_job)}
        per_rank = {"patches": [int(r[3]) for r in rows], "kernel_ms": [r[1] for r in rows], "collective": collective,
                    "pack_plus_allgather_ms": [r[2] for r in rows], "ms_per_step": [1e3 * r[0] / args.steps for r in rows],
                    "what": "per rank: its shard, HIP-event time of its refinement kernel, of hpmvs_pack_records + the "
                            "all-gather behind it, and its wall time per step between the barriers"}

    # ---- SURVEY 8d's wall definition ("seeds on host -> refined records on host"): the same step with the
    # seed arrays coming from pinned host memory and the refined fields copied back, timed separately (PCIe
    # inclusive; reported beside `value`, never as `value`: the bench contract times inputs resident in HBM)
    h2h = None
    if not use_dist and backend is None:
        in_keys = ("center", "normal", "scale", "n_images", "images")
        out_keys = ("center", "normal", "n_images", "images", "ok", "color", "fmin")
        host_in = {k: pristine[k].cpu().pin_memory() for k in in_keys}
        host_out = {k: torch.empty_like(work[k], device="cpu").pin_memory() for k in out_keys}
        def h2h_step():
            for k in in_keys:
                work[k].copy_(host_in[k], non_blocking=True)
            be.refine(work, n_seeds, max_images)
            for k in out_keys:
                host_out[k].copy_(work[k], non_blocking=True)
        h2h_step()
        be.sync()
        reps = max(1, args.steps)   # SURVEY 8d's metric, timed over the same number of steps as `value`
        th = time.perf_counter()
        for _ in range(reps):
            h2h_step()
        be.sync()
        th = (time.perf_counter() - th) / reps
        h2h = {"value": n_seeds / th, "unit": "patches/s", "ms_per_step": 1e3 * th, "steps": reps,
               "bytes_in": int(sum(host_in[k].numel() * host_in[k].element_size() for k in in_keys)),
               "bytes_out": int(sum(host_out[k].numel() * host_out[k].element_size() for k in out_keys)),
               "what": "pinned host seeds -> H2D -> optimize_kernel -> D2H of the refined Patch3d fields, one stream "
                       "(SURVEY 8d's wall definition)"}
        # the same wall with NO staging: the batch's pointers are the pinned host buffers themselves (mapped into the GPU's
        # address space: `on_device` only promises device-accessible pointers).  A patch's seed record is read once when a
        # lane fetches it and its refined fields are written once when it finishes, so the kernel streams ~350 B per patch
        # over PCIe while it runs instead of waiting for two copies around it.
        try:
            work_host = {k: (host_in[k].clone().pin_memory() if k in in_keys else torch.zeros_like(v, device="cpu").pin_memory())
                         for k, v in work.items()}
            def zc_step():
                for k in in_keys:
                    work_host[k].copy_(host_in[k])   # (host -> host: the caller's buffers are updated in place by a step)
                t0_ = time.perf_counter()
                be.refine(work_host, n_seeds, max_images)
                be.sync()
                return time.perf_counter() - t0_
            zc_step()
            tz = min(zc_step() for _ in range(min(3, reps)))
            same = all(torch.equal(work_host[k], work[k].cpu()) for k in ("ok", "center", "normal", "n_images", "images", "color", "fmin", "nevals"))
            h2h["zero_copy"] = {"value": n_seeds / tz, "unit": "patches/s", "ms_per_step": 1e3 * tz, "results_equal_staged": bool(same),
                                "what": "optimize_kernel launched on the pinned host buffers themselves (no H2D / D2H copies)"}
        except Exception as e:  # noqa: BLE001 -- reported, never fatal for the bench line
            h2h["zero_copy"] = {"error": str(e)[:200]}

    ok = work["ok"].cpu().numpy().astype(bool)
    ngrabs = work["ngrabs"].cpu().numpy().astype(np.int64)
    nevals = work["nevals"].cpu().numpy()
    nimg_out = work["n_images"].cpu().numpy()
    stage = work["stage"].cpu().numpy()
    # algorithmic bytes of one launch: 588 B per grab + 12 B per colour fetch + the record I/O
    alg_bytes = 588.0 * float(ngrabs.sum()) + 12.0 * float(nimg_out[ok].sum()) + 2.0 * float((36 + 4 * max_images) * n_seeds)
    achieved = alg_bytes / (last_ms * 1e-3) / 1e9

    # which build of the kernel the library picked for this batch size (kernel_launch.hip: 64 slots per wavefront from
    # 768 patches per CU upwards, unless HPMVS_SLOTS forces one)
    n_cus_dev = int(torch.cuda.get_device_properties(dev).multi_processor_count) if dev.type == "cuda" else 256
    forced = os.environ.get("HPMVS_SLOTS", "")
    kernel_name = "optimize_kernel_big" if (forced in ("58", "60", "64") or (forced != "29" and n_seeds >= 768 * n_cus_dev)) else "optimize_kernel"
    # HPMVS_SUMS=tree: the opt-in throughput build of the kernel (row sums first; NOT bit-identical to the oracle, and
    # never the default: a line measured in that mode says so here and in config.sums)
    sums_mode = "tree" if os.environ.get("HPMVS_SUMS") == "tree" else "exact"
    if sums_mode == "tree":
        kernel_name += "_tree"
    out = {
        "metric": "refined_patches_per_sec",
        "value": n_job * args.steps / elapsed,
        "unit": "patches/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "f32 textures / f64 optimiser",
        "data": "synthetic",
        "config": {"workload": workload, "views": views, "image": f"{W}x{H}", "pyramid_levels": 6,
                   "seeds_per_gpu": n_seeds, "seeds_per_step_all_gpus": n_job, "start_level": start_level,
                   "max_images": max_images, "order": args.order, "sums": sums_mode,
                   "parallelism": f"patch-shard x{world} + all-gather of {RECORD_BYTES} B records"},
        "value_is": "inputs resident in HBM when the timed region starts (the bench contract: a PCIe-inclusive rate is "
                    "never `value`); SURVEY 8d's host -> host rate of the same step is host_to_host.value",
        "host_to_host": h2h,
        "per_rank": per_rank,
        "successes_per_sec_rank0_extrapolated": n_job * float(ok.mean()) * args.steps / elapsed,
        "success_fraction": float(ok.mean()),
        "mean_evals_per_refined_patch": float(nevals[nevals > 0].mean()) if (nevals > 0).any() else 0.0,
        "mean_grabs_per_patch": float(ngrabs.mean()),
        "scene_bytes_hbm": int(be.gpu_scene.bytes()) if hasattr(be, "gpu_scene") else 0,
        "scene_setup_s": t_scene,
        "roofline": {"bound": "hbm", "kernel": kernel_name, "slots_per_wavefront": 64 if "_big" in kernel_name else 29,
                     "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel_ms": last_ms, "kernel_ms_last_launch_library_events": lib_ms,
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "grabs_per_launch": int(ngrabs.sum())},
    }
    if use_dist and args.check_gather and rank == 0:
        # the gathered bytes of this rank's shard against (a) the torch packer applied to the results and (b) the
        # result fields themselves
        mine = plan.my_slice(gathered)
        ref = dd.pack_records(work["center"], work["normal"], work["color"], work["scale"], work["fmin"], work["ok"],
                              work["n_images"], work["images"], check=True)
        u = dd.unpack_records(mine)
        okb = work["ok"].cpu().numpy().astype(bool)
        fields = (np.array_equal(u["center"], work["center"].cpu().numpy()) and np.array_equal(u["ok"], okb) and
                  np.array_equal(u["n_images"], work["n_images"].cpu().numpy()) and
                  np.array_equal(u["fmin"], work["fmin"].cpu().numpy()))
        coll = "RCCL" if getattr(be, "dist_backend", "nccl") == "nccl" else getattr(be, "dist_backend", "?")
        out["gather_check"] = {"collective": f"all_gather_into_tensor over {coll} ({world} rank{'s' if world > 1 else ''})",
                               "records": int(mine.shape[0]), "gathered_total": int(gathered.shape[0]),
                               "bytes_equal_python_packer": bool(torch.equal(mine, ref)), "fields_equal_results": bool(fields)}
    if backend is not None:  # tests: hand the gathered round back with the line
        out["_gathered"] = gathered

    # counters of a PMC pass over this very workload (profiles/collect.sh): only if they were collected from the
    # library build that is loaded now
    build_id = api.build_id() if backend is None else "stub"
    out["library_build_id"] = build_id
    out["n_cus"] = n_cus_dev
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc) and backend is None:
        try:
            rec = json.load(open(pmc)).get(workload if world == 1 else "")
            if rec and rec.get("seeds_per_gpu") == n_seeds and rec.get("library_build_id") == build_id:
                out["roofline"]["traffic"] = rec["hbm_bytes_per_launch"]
                out["roofline"]["traffic_source"] = rec.get("source")
                if "valu_thread_cycles" in rec and "kernel_cycles" in rec:
                    out["roofline_issue"] = issue_roofline(rec)
            elif rec:
                out["roofline"]["traffic_note"] = ("profiles/pmc_traffic.json holds counters of another build "
                                                   f"({rec.get('library_build_id')}); not reported")
        except Exception as e:  # a malformed record must not cost the line
            out["roofline"]["traffic_note"] = f"pmc_traffic.json unreadable: {e}"

    # ---- CPU baseline + parity on a bounded sample (rank 0, N = 1 only)
    if rank == 0 and world == 1 and not args.no_cpu and backend is None:
        try:
            cpu_leg(out, args, scene, seeds, n_seeds, work, ok, nevals, stage)
        except Exception as e:  # the GPU line must survive a box without the checker
            out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and backend is None:
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return out


def issue_roofline(rec):
    """The compute-side view of the refinement kernel from a PMC record: VALU lane-cycles used / what the machine
    offers in the kernel's time.  What the machine offers per SIMD and clock is MEASURED (tools/valu_calibrate.hip ->
    profiles/calibration/valu_rates.json: dense independent v_fma_f32 at two wavefronts per SIMD), not assumed."""
    lanes_per_clk, src = 16.0, "assumed: 16 lanes per SIMD per clock (no calibration record found)"
    cal = os.path.join(ROOT, "profiles", "calibration", "valu_rates.json")
    try:
        tests = json.load(open(cal))["tests"]
        f32 = [t for t in tests if t["instruction"] == "v_fma_f32" and t["waves_per_simd"] == 2]
        if f32:
            lanes_per_clk = float(f32[0]["lanes_per_cycle_per_simd"])
            src = "measured: profiles/calibration/valu_rates.json (v_fma_f32, 2 wavefronts per SIMD)"
    except Exception:
        pass
    # SQ_THREAD_CYCLES_VALU over CUs x 4 SIMDs x 16 x cycles is (active lanes per VALU instruction / 64) x (share of
    # 4-cycle issue slots with a VALU instruction): lane-cycles of a SIMD that issues 16 lanes per clock.  A SIMD that
    # really issues `lanes_per_clk` offers lanes_per_clk / 16 times as many.
    used = float(rec["valu_thread_cycles"])
    peak16 = float(rec["n_cus"]) * 4 * 16 * float(rec["kernel_cycles"])
    peak = peak16 * lanes_per_clk / 16.0
    return {"bound": "valu", "lane_cycles_used": used, "lane_cycles_peak": peak, "frac": used / peak,
            "lanes_per_simd_per_clock": lanes_per_clk, "peak_source": src,
            "frac_if_16_lanes_per_clock": used / peak16,
            "valu_insts_per_patch": rec.get("valu_insts_per_patch"),
            "valu_lane_utilization": rec.get("valu_lane_utilization"),
            "issue_slot_frac": rec.get("valu_issue_slot_frac"),
            "source": rec.get("source")}


def usable_cores(hw_threads):
    """How many cores this process may really use: the affinity mask and the cgroup CPU quota.  The GPU boxes of this
    pool are 2 x 64-core EPYC (256 hardware threads visible) behind a CFS quota of 16 CPUs (cpu.max = "1600000 100000"):
    256 OpenMP threads then share 16 cores' worth of time and are throttled in bursts -- the CPU leg of rounds 1-3 ran
    that way and 'scaled' 9x (profiles/r04/cpu_scaling_c1.txt: linear to 16 threads, 25x at 32, falling beyond).  The
    baseline is timed with as many threads as the quota grants and says so."""
    n = hw_threads
    note = "none found"
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            lim = max(1, int(float(q) / float(per) + 0.5))
            note = f"cgroup cpu.max = {q} {per}: {lim} CPUs"
            n = min(n, lim)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                lim = max(1, int(q / per + 0.5))
                note = f"cgroup cfs quota {q}/{per}: {lim} CPUs"
                n = min(n, lim)
        except Exception:
            pass
    return max(1, n), note


def cpu_leg(out, args, scene, seeds, n_seeds, work, ok, nevals, stage):
    """CPU baseline + parity on a bounded sample of the same seeds (the oracle is the checker, never the product)."""
    from oracle import oracle as orc
    m = min(args.cpu_sample, n_seeds)
    idx = np.arange(m)
    osc = orc.OracleScene(scene)
    P = orc.patches_from_seeds(seeds, idx)
    hw_threads = os.cpu_count() or 1
    cores, quota_note = usable_cores(hw_threads)
    which = orc.best_optimizer()
    orc.optimize_batch(osc, P, which=which, n_threads=cores)   # parity sample
    # timing: a bigger bounded sample so that thread start-up does not dominate (a few seconds of all-core work)
    mt = min(args.cpu_timing_sample, n_seeds)
    Pt = orc.patches_from_seeds(seeds, np.arange(mt))
    tc = time.perf_counter()
    orc.optimize_batch(osc, Pt, which=which, n_threads=cores)
    tc = time.perf_counter() - tc
    okt = int(sum(1 for p in Pt if p.stage == 0))
    del Pt
    # the same loop on ONE core, smaller sample
    m1 = min(args.cpu_sample_1core, m)
    P1 = orc.patches_from_seeds(seeds, np.arange(m1))
    t1 = time.perf_counter()
    orc.optimize_batch(osc, P1, which=which, n_threads=1)
    t1 = time.perf_counter() - t1
    cpu_model, sockets = "unknown", 0
    try:
        info = open("/proc/cpuinfo").read()
        names = [l.split(":", 1)[1].strip() for l in info.splitlines() if l.startswith("model name")]
        cpu_model = names[0] if names else "unknown"
        sockets = len({l.split(":", 1)[1].strip() for l in info.splitlines() if l.startswith("physical id")})
    except Exception:
        pass
    okc = np.array([p.stage == 0 for p in P])
    okg = ok[:m]
    both = np.nonzero(okc & okg)[0]
    fmin_g = work["fmin"][:m].cpu().numpy()
    cen_g = work["center"][:m].cpu().numpy()
    nor_g = work["normal"][:m].cpu().numpy()
    unrob = lambda f: f / (1.0 - 3.0 * f)
    dncc, basin, ident, ddepth, dangle = [], [], [], [], []
    col_g = work["color"][:m].cpu().numpy()
    nim_g = work["n_images"][:m].cpu().numpy()
    img_g = work["images"][:m].cpu().numpy()
    nev_g = nevals[:m]
    for k in both:
        p = P[k]
        ident.append(np.array_equal(np.array(p.center[:], dtype=np.float32), cen_g[k]) and
                     np.array_equal(np.array(p.normal[:], dtype=np.float32), nor_g[k]) and
                     np.array_equal(np.array(p.color[:], dtype=np.float32), col_g[k]) and
                     p.nevals == nev_g[k] and list(p.images[:p.n_images]) == list(img_g[k, :nim_g[k]]))
        dncc.append(abs(unrob(p.fmin) - unrob(fmin_g[k])))
        ddist = np.linalg.norm(np.array(p.center[:3]) - cen_g[k, :3]) / float(seeds.scale[k])
        na, nb = np.array(p.normal[:3], dtype=np.float64), nor_g[k, :3].astype(np.float64)
        ang = float(np.degrees(np.arctan2(np.linalg.norm(np.cross(na, nb)), np.dot(na, nb))))  # (0 for identical vectors; arccos of a float32 dot is not)
        ddepth.append(ddist); dangle.append(ang)
        basin.append(ddist < 0.25 and ang < 2.0)
    out["cpu_baseline"] = {
        "value": mt / tc, "unit": "patches/s", "cores": cores,
        "kind": "port",
        "kind_detail": ("port with the reference's own optimiser: photometric path = oracle/hpmvs_oracle.c (C restatement of PatchOptimizer.cpp / "
                        "Camera / Image / Patch2d), optimiser = the reference's genuine NLopt 2.4.2 BOBYQA compiled from /root/reference (oracle/_ref)"
                        if which == orc.OPT_REF else "port: photometric path and optimiser both restated (oracle/_ref did not travel)"),
        "sample": f"first {mt} seeds of the same workload, {cores} OpenMP threads (all this container may use), schedule(dynamic,64) over patches like "
                  f"Scene::initPatches; photometric path = oracle/ C restatement, optimiser = "
                  f"{'genuine NLopt 2.4.2 BOBYQA (oracle/_ref)' if which == orc.OPT_REF else 'oracle BOBYQA port'}",
        "seconds": tc, "successes": okt,
        "optimizer_is_reference_nlopt": bool(which == orc.OPT_REF),
        "cpu_model": cpu_model, "sockets": sockets, "hardware_threads": hw_threads,
        "threads_used": cores, "cpu_quota": quota_note,
        "one_core": {"value": m1 / t1, "unit": "patches/s", "cores": 1, "sample": f"first {m1} seeds", "seconds": t1},
        # the loop over independent patches must scale with the cores it names
        "parallel_efficiency": (mt / tc) / (cores * (m1 / t1)),
    }
    out["parity"] = {
        "sample": m, "decision_agreement": float((okc == okg).mean()),
        "mean_abs_ncc_delta": float(np.mean(dncc)) if dncc else None,
        "same_basin_fraction": float(np.mean(basin)) if basin else None,
        "bit_identical_patch_fraction": float(np.mean(ident)) if ident else None,
        "stage_agreement": float((np.array([p.stage for p in P]) == stage[:m]).mean()),
        # success-set agreement and the geometric deltas of the patches both paths refined
        "success_set_jaccard": float((okc & okg).sum() / max(1, (okc | okg).sum())),
        "abs_ddepth_over_scale": {"median": float(np.median(ddepth)) if ddepth else None,
                                  "p95": float(np.percentile(ddepth, 95)) if ddepth else None,
                                  "max": float(np.max(ddepth)) if ddepth else None},
        "normal_angle_deg": {"median": float(np.median(dangle)) if dangle else None,
                             "p95": float(np.percentile(dangle, 95)) if dangle else None,
                             "max": float(np.max(dangle)) if dangle else None},
    }


if __name__ == "__main__":
    main()
