"""EVERY patch of a workload against the oracle (genuine NLopt inside), not a sample:  full_parity.py <views> <w> <h> <n> [start_level [max_images]]
GPU: one hpmvs_optimize_batch call.  CPU: the oracle on all cores the container grants, in chunks.  Compared bit for bit: the failing
stage of every patch; centre, normal, colour, image list and evaluation count of every refined one.  Every differing patch is
then run through the oracle again with the oracle's libm calls replaced by what the device computes (orc_set_libm_mode: acos()
correctly rounded like cr_acos.hpp, sin() / cos() of setCenterNorm by the device's series): a patch that then equals the GPU's
result differs because of the host libm's last bits and nothing else."""
import sys, os, time, json, math, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
from hpmvs_amd import api, synth
from oracle import oracle as orc


def run(V, W_, H_, N_, SL=4, MI=32, verbose=False):
    """MI: rows of the batch's image lists (a list that outgrows them ends the patch at stage 100)"""
    dev = torch.device('cuda', 0)
    scene = synth.make_scene(V, W_, H_, n_waves=24, device=dev)
    seeds = synth.make_seeds(scene, N_, start_level=SL, max_images=MI)
    g = api.Scene(scene)
    b = api.Batch.from_seeds(seeds)
    t0 = time.perf_counter(); api.optimize_batch(g, b); t_gpu = time.perf_counter() - t0
    kernel_ms = g.last_kernel_ms()
    n_wide = g.last_wide_patches()   # patches whose list outgrew the batch kernel's 64-id rows (redone by kernel_wide.hip)
    osc = orc.OracleScene(scene)
    cores = min(16, os.cpu_count() or 1)
    try:
        q, p_ = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max': cores = max(1, min(os.cpu_count() or 1, int(int(q) / int(p_))))
    except Exception:
        pass
    f32 = np.float32
    differing, refined_both, t_cpu = [], 0, 0.0
    CH = 250000
    for lo in range(0, N_, CH):
        hi = min(N_, lo + CH); m = hi - lo
        arr = (orc.Patch * m)()
        v = orc.patch_view(arr)
        v['center'] = seeds.center[lo:hi]; v['normal'] = seeds.normal[lo:hi]; v['scale'] = seeds.scale[lo:hi]
        v['n_images'] = seeds.n_images[lo:hi]
        v['images'][:, :seeds.images.shape[1]] = np.where(seeds.images[lo:hi] < 0, 0, seeds.images[lo:hi])
        t0 = time.perf_counter(); orc.optimize_batch(osc, arr, which=orc.best_optimizer(), n_threads=cores); t_cpu += time.perf_counter() - t0
        v = orc.patch_view(arr)
        okc = v['stage'] == 0
        okg = b.ok[lo:hi].astype(bool)
        bad = v['stage'] != b.stage[lo:hi]
        both = okc & okg
        refined_both += int(both.sum())
        same = (np.all(v['center'].view(np.uint32) == b.center[lo:hi].view(np.uint32), axis=1) &
                np.all(v['normal'].view(np.uint32) == b.normal[lo:hi].view(np.uint32), axis=1) &
                np.all(v['color'].view(np.uint32) == b.color[lo:hi].view(np.uint32), axis=1) &
                (v['nevals'] == b.nevals[lo:hi]) & (v['n_images'] == b.n_images[lo:hi]))
        live = np.arange(seeds.images.shape[1])[None, :] < np.clip(v['n_images'], 0, seeds.images.shape[1])[:, None]
        same &= np.all((v['images'][:, :seeds.images.shape[1]] == b.images[lo:hi]) | ~live, axis=1)
        bad |= both & ~same
        for k in np.nonzero(bad)[0]:
            differing.append((lo + int(k), int(v['images'][k, 0]) if v['n_images'][k] > 0 else int(seeds.images[lo + k, 0]),
                              int(v['stage'][k]), int(b.stage[lo + k]), int(v['nevals'][k]), int(b.nevals[lo + k])))
        if verbose: print('chunk', lo, hi, 'differing so far', len(differing), flush=True)

    # ---- attribution: the differing patches again, with the oracle's libm calls replaced by what the device uses
    def rerun(mode, ks):
        orc.set_libm_mode(mode)
        arr = (orc.Patch * len(ks))()
        v = orc.patch_view(arr)
        v['center'] = seeds.center[ks]; v['normal'] = seeds.normal[ks]; v['scale'] = seeds.scale[ks]; v['n_images'] = seeds.n_images[ks]
        v['images'][:, :seeds.images.shape[1]] = np.where(seeds.images[ks] < 0, 0, seeds.images[ks])
        orc.optimize_batch(osc, arr, which=orc.best_optimizer(), n_threads=cores)
        orc.set_libm_mode(0)
        v = orc.patch_view(arr)
        out = []
        for r, k in enumerate(ks):
            n = int(v['n_images'][r])
            out.append(bool(v['stage'][r] == b.stage[k] and (v['stage'][r] != 0 or (
                np.array_equal(v['center'][r].view(np.uint32), b.center[k].view(np.uint32)) and
                np.array_equal(v['normal'][r].view(np.uint32), b.normal[k].view(np.uint32)) and
                np.array_equal(v['color'][r].view(np.uint32), b.color[k].view(np.uint32)) and
                v['nevals'][r] == b.nevals[k] and n == b.n_images[k] and list(v['images'][r, :n]) == list(b.images[k, :n])))))
        return out
    ks = np.array([d[0] for d in differing], dtype=np.int64)
    rows, tally = [], dict(equal_with_correctly_rounded_acos=0, equal_with_device_sincos=0, equal_with_both=0, still_differing=0)
    if len(ks):
        m1, m2, m3 = rerun(1, ks), rerun(2, ks), rerun(3, ks)
        for r, (k, v0, so, sg, no, ng) in enumerate(differing):
            why = 'acos' if m1[r] else ('sincos' if m2[r] else ('acos+sincos' if m3[r] else 'unexplained'))
            tally['equal_with_correctly_rounded_acos' if m1[r] else 'equal_with_device_sincos' if m2[r] else 'equal_with_both' if m3[r] else 'still_differing'] += 1
            rows.append(dict(patch=k, stage_oracle=so, stage_gpu=sg, nevals_oracle=no, nevals_gpu=ng, oracle_equals_gpu_with=why))
    known, other = len(ks) - tally['still_differing'], tally['still_differing']
    rec = dict(scene=f"{V} x {W_}x{H_}", start_level=SL, max_images=MI, patches=N_, refined_gpu=int(b.ok.sum()), refined_by_both=refined_both,
               redone_by_the_wide_kernel=int(n_wide), stage_100_gpu=int((b.stage == 100).sum()), longest_refined_list=int(b.n_images[b.ok.astype(bool)].max()) if b.ok.any() else 0,
               differing=len(differing), attributed_to_libm=int(known), unexplained=int(other), attribution=tally,
               bit_identical_fraction_of_refined=1.0 - len(differing) / max(1, refined_both),
               gpu_kernel_ms=kernel_ms, gpu_call_s=round(t_gpu, 3), oracle_s=round(t_cpu, 1), oracle_threads=cores,
               oracle_optimizer='genuine NLopt 2.4.2 BOBYQA' if orc.best_optimizer() == orc.OPT_REF else 'restated', build=api.lib().hpmvs_build_id().decode(),
               differing_patches=rows[:64])
    g.close()
    return rec


if __name__ == '__main__':
    a = [int(t) for t in sys.argv[1:]]
    print(json.dumps(run(a[0], a[1], a[2], a[3], a[4] if len(a) > 4 else 4, a[5] if len(a) > 5 else 32, verbose=True)))
