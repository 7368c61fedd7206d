"""The two kernels behind hpmvs_objective_batch on the same requests: values byte for byte, kernel time (HIP events).
   objective_ab.py [views w h n reps]      (default: the C2 scene, 1e6 patches)
Needs the measurement build:  make -C hpmvs_amd/csrc LANE=1 OUT=../../tools/ab/lane/libhpmvs_amd.so
rows = the refinement kernel's evaluator stand-alone (seven lanes per grab); lane = kernel_objective_lane.hip (one lane per grab)."""
import sys, os, json, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
from hpmvs_amd import api, synth
api.LIB_PATH = os.environ.get('HPMVS_AB_LIB', 'tools/ab/lane/libhpmvs_amd.so')
V, W_, H_, N_, R_ = (int(a) for a in sys.argv[1:6]) if len(sys.argv) > 5 else (50, 3840, 2160, 1000000, 5)
dev = torch.device('cuda', 0)
scene = synth.make_scene(V, W_, H_, n_waves=24, device=dev)
g = api.Scene(scene)
seeds = synth.make_seeds(scene, N_, start_level=4, max_images=32)
t = dict(center=torch.from_numpy(seeds.center).to(dev), normal=torch.from_numpy(seeds.normal).to(dev), scale=torch.from_numpy(seeds.scale).to(dev),
         n_images=torch.from_numpy(seeds.n_images).to(dev), images=torch.from_numpy(np.ascontiguousarray(seeds.images)).to(dev))
b = api.PatchBatch(); b.n, b.max_images = N_, seeds.images.shape[1]
for k, v in t.items(): setattr(b, k, v.data_ptr())
rng = np.random.default_rng(5)
xh = np.zeros((N_, 3)); xh[:, 0] = rng.normal(0, 0.02, N_); xh[:, 1:] = rng.uniform(-6, 6, (N_, 2))   # depth steps and angles an optimiser visits
xh[: N_ // 4] = 0.0
x = torch.from_numpy(xh).to(dev)
o = api.default_options(); L = api.lib(); st = torch.cuda.current_stream().cuda_stream
out = {}
for kern in ('rows', 'lane'):
    os.environ['HPMVS_OBJECTIVE_KERNEL'] = kern
    f = torch.full((N_,), -1.0, dtype=torch.float64, device=dev); gr = torch.full((N_,), -1, dtype=torch.int32, device=dev)
    ms = []
    for _ in range(R_):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rc = L.hpmvs_objective_batch(g.h, C.byref(o), C.byref(b), x.data_ptr(), f.data_ptr(), gr.data_ptr(), 1, C.c_void_p(st)); e1.record()
        assert rc == 0, L.hpmvs_last_error(); torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
    out[kern] = dict(ms=min(ms), f=f.cpu().numpy(), g=gr.cpu().numpy())
G = int(out['rows']['g'].sum())
same_f = bool(np.array_equal(out['rows']['f'].view(np.uint64), out['lane']['f'].view(np.uint64)))
same_g = bool(np.array_equal(out['rows']['g'], out['lane']['g']))
rec = dict(scene=f"{V} x {W_}x{H_}", patches=N_, grabs=G, plateau=int((out['rows']['f'] == 2.0).sum()),
           rows_ms=round(out['rows']['ms'], 3), lane_ms=round(out['lane']['ms'], 3),
           rows_Ggrabs_s=round(G / out['rows']['ms'] / 1e6, 3), lane_Ggrabs_s=round(G / out['lane']['ms'] / 1e6, 3),
           values_identical=same_f, grab_counts_identical=same_g, build=L.hpmvs_build_id().decode())
print(json.dumps(rec))
if not (same_f and same_g):
    d = np.flatnonzero(out['rows']['f'].view(np.uint64) != out['lane']['f'].view(np.uint64))
    print('differing', d.size, d[:10], out['rows']['f'][d[:10]], out['lane']['f'][d[:10]], out['rows']['g'][d[:10]], out['lane']['g'][d[:10]])
    sys.exit(1)
