"""Small-batch behaviour: an expansion round of the real scheduler hands over 1e3 - 1e5 patches, not
1e6.  A batch is a set of serial chains (a patch = ~90, up to 1000, evaluation rounds), so a lone small launch is
latency bound; consecutive rounds can be launched on several streams with disjoint workspaces (the C ABI hands out
16 per scene) so that one round's ramp-down hides under the next round's start.  Prints patches/s for batch sizes
1e3 / 1e4 / 1e5 launched back to back on 1, 2, 4 and 8 streams, device-resident inputs (C1 scene)."""
import ctypes as C, json, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from hpmvs_amd import api, synth
import bench

dev = torch.device('cuda', 0)
scene = synth.make_scene(12, 1920, 1080, n_waves=24, device=dev)
g = api.Scene(scene)
L = api.lib()
opts = api.default_options()
total = 400_000
seeds = synth.make_seeds(scene, total, start_level=4, max_images=32)
rows = []
for n in (1000, 10_000, 100_000):
    nb = min(64, total // n)
    batches = []
    for k in range(nb):
        sl = slice(k * n, (k + 1) * n)
        s = synth.Seeds(center=seeds.center[sl], normal=seeds.normal[sl], scale=seeds.scale[sl], n_images=seeds.n_images[sl],
                        images=seeds.images[sl], truth=seeds.truth[sl])
        batches.append(bench.device_batch(s, dev, 32))
    pristine = [{k: t[k].clone() for k in ("center", "normal", "n_images", "images")} for t in batches]
    def restore():
        for t, p in zip(batches, pristine):
            for k, v in p.items():
                t[k].copy_(v)
        torch.cuda.synchronize()
    for ns in (1, 2, 4, 8):
        if ns > nb:
            continue
        streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
        def run():
            for k, t in enumerate(batches):
                st = streams[k % ns]
                b = bench.c_batch(t, n, 32)
                rc = L.hpmvs_optimize_batch(g.h, C.byref(opts), C.byref(b), 1, C.c_void_p(st.cuda_stream))
                assert rc == 0, L.hpmvs_last_error()
        restore(); run(); torch.cuda.synchronize()   # warm
        best = 1e9
        for _ in range(3):
            restore()   # every timed pass refines the SEED patches again (a refined patch re-refines much faster)
            t0 = time.perf_counter(); run(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        rows.append(dict(batch=n, batches=nb, streams=ns, seconds=best, patches_per_s=nb * n / best))
        print(rows[-1], flush=True)
json.dump(rows, open('gpurun_out/small_batch.json', 'w'), indent=1)
