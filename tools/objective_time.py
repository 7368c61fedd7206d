"""Kernel time of the stand-alone objective kernel (pure photometric service: wave_eval_rows without the optimiser) for an
alternative build of the library, device-resident inputs, HIP events:  objective_time.py <lib.so> [views w h n reps]"""
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
from hpmvs_amd import api, synth
api.LIB_PATH = sys.argv[1]
V, W_, H_, N_, R_ = (int(a) for a in sys.argv[2:7]) if len(sys.argv) > 6 else (12, 1920, 1080, 100000, 5)
dev = torch.device('cuda', 0)
scene = synth.make_scene(V, W_, H_, n_waves=24, device=dev)
g = api.Scene(scene)
seeds = synth.make_seeds(scene, N_, start_level=4, max_images=32)
t = dict(center=torch.from_numpy(seeds.center).to(dev), normal=torch.from_numpy(seeds.normal).to(dev), scale=torch.from_numpy(seeds.scale).to(dev),
         n_images=torch.from_numpy(seeds.n_images).to(dev), images=torch.from_numpy(np.ascontiguousarray(seeds.images)).to(dev))
b = api.PatchBatch(); b.n, b.max_images = N_, seeds.images.shape[1]
for k, v in t.items(): setattr(b, k, v.data_ptr())
x = torch.zeros((N_, 3), dtype=torch.float64, device=dev); x[:, 1:] = 1.0
f = torch.zeros(N_, dtype=torch.float64, device=dev); gr = torch.zeros(N_, dtype=torch.int32, device=dev)
o = api.default_options(); L = api.lib(); st = torch.cuda.current_stream().cuda_stream
ms = []
for _ in range(R_):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); rc = L.hpmvs_objective_batch(g.h, C.byref(o), C.byref(b), x.data_ptr(), f.data_ptr(), gr.data_ptr(), 1, C.c_void_p(st)); e1.record()
    assert rc == 0; torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
G = int(gr.sum())
print(sys.argv[1], 'objective kernel ms min %.3f' % min(ms), 'grabs', G, 'G grabs/s %.2f' % (G / min(ms) / 1e6), 'checksum', float(f.sum()))
