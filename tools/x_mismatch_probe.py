"""Which jointly refined patches return float64 x[3] that differ from the oracle's (genuine NLopt), and by how much."""
import sys, numpy as np, os
sys.path.insert(0, '.')
from hpmvs_amd import api, synth
from oracle import oracle as orc
scene = synth.make_scene(3, 640, 480, n_waves=16)
seeds = synth.make_seeds(scene, 3000, start_level=2)
g = api.Scene(scene, device=0)
b = api.Batch.from_seeds(seeds); api.optimize_batch(g, b)
osc = orc.OracleScene(scene)
P = orc.patches_from_seeds(seeds)
orc.optimize_batch(osc, P, which=orc.best_optimizer(), n_threads=os.cpu_count())
both = [k for k, p in enumerate(P) if p.stage == 0 and b.ok[k]]
bad = [k for k in both if list(P[k].x[:]) != list(b.x[k])]
print('both', len(both), 'x differs', len(bad))
for k in bad[:25]:
    p = P[k]
    d = [abs(float(p.x[i]) - float(b.x[k, i])) / max(abs(float(p.x[i])), 1e-300) for i in range(3)]
    print(k, 'nev', p.nevals, b.nevals[k], 'rel diff', ['%.1e' % v for v in d],
          'fmin eq', p.fmin == b.fmin[k], 'center eq', list(np.array(p.center[:], dtype=np.float32)) == list(b.center[k]))

# the start point: x0 = parametersFromCenterNorm, where libm (asinf, cos, acos) enters -- recomputed on the host
print('x mismatch with different nevals:', sum(1 for k in bad if P[k].nevals != b.nevals[k]), 'of', len(bad))
