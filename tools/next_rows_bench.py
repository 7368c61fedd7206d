"""Measurement of the SURVEY 8f rows built next to the hot path, on the C1 scene (12 x 1080p):
pyramid build, the Scene::initPatches seed loop, CellProcessor::extend / ::branch frontiers -- GPU (C ABI,
host pointers, wall clock incl. PCIe staging) beside the CPU oracle on all host threads."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from hpmvs_amd import api, synth
from oracle import oracle as orc

dev = torch.device('cuda', 0)
scene = synth.make_scene(12, 1920, 1080, n_waves=24, device=dev)
out = {}
# ---- row 1: pyramid (5 half-resizes per view); timed through the stand-alone entry on a level-0 image
img = scene.views[0].rgb if isinstance(scene.views[0].rgb, np.ndarray) else scene.views[0].rgb.cpu().numpy()
t0 = time.perf_counter(); lv = img
for _ in range(5): lv = api.build_pyramid(lv)
t_gpu = time.perf_counter() - t0
t0 = time.perf_counter(); lv = img
for _ in range(5): lv = orc.half_resize(lv)
t_cpu = time.perf_counter() - t0
out['pyramid_1080p_5_levels'] = dict(gpu_wall_ms_incl_pcie=t_gpu * 1e3, cpu_oracle_1_thread_ms=t_cpu * 1e3)
t0 = time.perf_counter(); g = api.Scene(scene); torch.cuda.synchronize(); t_scene = time.perf_counter() - t0
out['scene_upload_12_views_incl_pyramids_ms'] = t_scene * 1e3
osc = orc.OracleScene(scene)
ncpu = os.cpu_count() or 1
# ---- row 2: seed loop
n = 100000
xyz, off, mi = synth.make_nvm_points(scene, n, start_level=4)
t_gpu = 1e9
for _ in range(4):
    t0 = time.perf_counter(); b = api.init_patches_batch(g, xyz, off, mi, start_level=4, max_images=32); t_gpu = min(t_gpu, time.perf_counter() - t0)
m = 16384
t0 = time.perf_counter(); P = orc.init_patches(osc, xyz[:m], off[:m + 1], mi[:off[m]], start_level=4, n_threads=ncpu); t_cpu = time.perf_counter() - t0
out['init_patches_1e5_points'] = dict(gpu_wall_ms=t_gpu * 1e3, gpu_kernel_ms=g.last_kernel_ms(), gpu_points_per_s=n / t_gpu,
                                      survivors=int(b.ok.sum()), cpu_points_per_s=m / t_cpu, cpu_threads=ncpu, cpu_sample=m)
# ---- row 3: frontiers
keep = np.nonzero(b.ok)[0][:16384]
par = api.Batch(b.center[keep], b.normal[keep], b.scale[keep], b.n_images[keep], b.images[keep])
for mode, name, wf in ((api.EXPAND_EXTEND, 'extend', 1.0), (api.EXPAND_BRANCH, 'branch', 1.6)):
    width = (par.scale * np.float32(2.0 / 0.9 * wf)).astype(np.float32)
    cc = np.ascontiguousarray(par.center[:, :3])
    t_gpu = 1e9
    for _ in range(4):   # (the best of four: a call now and then stalls for tens of ms in the allocator)
        t0 = time.perf_counter(); c = api.expand_batch(g, mode, par, cc, width); t_gpu = min(t_gpu, time.perf_counter() - t0)
    mm = 2048
    OP = (orc.Patch * mm)()
    for k in range(mm):
        OP[k].center[:] = par.center[k].tolist(); OP[k].normal[:] = par.normal[k].tolist(); OP[k].scale = float(par.scale[k])
        OP[k].n_images = int(par.n_images[k])
        for j in range(par.n_images[k]): OP[k].images[j] = int(par.images[k, j])
    t0 = time.perf_counter(); orc.expand_batch(osc, mode, OP, cc[:mm], width[:mm], n_threads=ncpu); t_cpu = time.perf_counter() - t0
    N = 6 if mode == api.EXPAND_EXTEND else 4
    out['%s_%d_parents' % (name, par.n)] = dict(candidates=par.n * N, accepted=int(c.ok.sum()), gpu_wall_ms=t_gpu * 1e3,
                                                gpu_kernel_ms=g.last_kernel_ms(), gpu_candidates_per_s=par.n * N / t_gpu,
                                                cpu_candidates_per_s=mm * N / t_cpu, cpu_threads=ncpu, cpu_sample_parents=mm)
print(json.dumps(out, indent=1))
