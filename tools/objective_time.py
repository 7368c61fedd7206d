"""Time of the stand-alone objective kernel (pure photometric service: wave_eval_rows without the optimiser) for an alternative
build of the library:  objective_time.py <lib.so> [views w h n reps]"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from hpmvs_amd import api, synth
api.LIB_PATH = sys.argv[1]
V, W_, H_, N_, R_ = (int(a) for a in sys.argv[2:7]) if len(sys.argv) > 6 else (12, 1920, 1080, 100000, 5)
dev = torch.device('cuda', 0)
scene = synth.make_scene(V, W_, H_, n_waves=24, device=dev)
g = api.Scene(scene)
seeds = synth.make_seeds(scene, N_, start_level=4, max_images=32)
b = api.Batch.from_seeds(seeds)
x = np.zeros((b.n, 3)); x[:, 1:] = 1.0
ts = []
for _ in range(R_):
    t0 = time.perf_counter(); f, gr = api.objective_batch(g, b, x); ts.append(time.perf_counter() - t0)
print(sys.argv[1], 'objective_batch wall ms min %.2f' % (1e3 * min(ts)), 'grabs', int(gr.sum()), 'checksum', float(f.sum()))
