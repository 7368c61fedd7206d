"""Stream kernel (kernel_stream.hip) against the wavefront-bound kernel (kernel_optimize.hip) on the same batch:
every output array must be byte-identical; prints both kernel times.
    python tools/stream_check.py [views w h n start_level]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, '.')
from hpmvs_amd import api, synth
a = sys.argv[1:]
if a and a[0].endswith('.so'):
    api.LIB_PATH = a[0]; a = a[1:]
views, w, h, n, sl = (int(a[0]), int(a[1]), int(a[2]), int(a[3]), int(a[4])) if len(a) >= 5 else (12, 1920, 1080, 100000, 4)
dev = torch.device('cuda', 0)
scene = synth.make_scene(views, w, h, n_waves=24, device=dev)
g = api.Scene(scene)
seeds = synth.make_seeds(scene, n, start_level=sl, max_images=32)
res = {}
for mode in ('wave', 'stream', 'stream', 'wave'):
    os.environ['HPMVS_KERNEL'] = mode
    b = api.Batch.from_seeds(seeds)
    t0 = time.perf_counter(); api.optimize_batch(g, b); t1 = time.perf_counter()
    print(mode, 'kernel ms %.2f' % g.last_kernel_ms(), 'wall ms %.1f' % ((t1 - t0) * 1e3), 'ok', float(b.ok.mean()), flush=True)
    res.setdefault(mode, b)
bad = 0
for f in ('ok', 'center', 'normal', 'n_images', 'images', 'color', 'ncc', 'fmin', 'x', 'result', 'nevals', 'stage', 'ngrabs'):
    x, y = getattr(res['wave'], f), getattr(res['stream'], f)
    same = np.array_equal(x, y)
    if not same:
        bad += 1
        d = np.nonzero((x != y).reshape(len(x), -1).any(axis=1))[0]
        print('MISMATCH', f, len(d), 'patches, first', d[:5])
print('IDENTICAL' if bad == 0 else 'DIFFERENT', 'n', n)
