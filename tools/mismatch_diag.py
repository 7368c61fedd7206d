"""What a GPU-vs-oracle mismatch of tools/random_parity_sweep.py starts with:  mismatch_diag.py <seed> <opts|-> <scene index>
Replays the sweep's random choices up to that scene, runs it, and for every differing patch prints both results and the
float64 start point of the optimiser recomputed on the host two ways: with this host's libm (what the oracle uses) and correctly
rounded (mpmath; what the device's cr_acos.hpp returns).  DESIGN section 5: the one known source of differences is glibc's
acos(), which is < 1 ulp but not correctly rounded for ~0.06 % of arguments; BOBYQA occasionally amplifies that last bit."""
import sys, math, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import mpmath
from random_parity_sweep import draw, run_scene
from hpmvs_amd import api
seed, ropts, target = int(sys.argv[1]), sys.argv[2] == 'opts', int(sys.argv[3])
rng = np.random.default_rng(seed)
for it in range(target + 1): c = draw(rng, ropts)
b, P, seeds, scene, bad = run_scene(c, torch.device('cuda', 0))
print('scene', target, c)
print('differing patches', bad)
f32 = np.float32
def nrm(v):
    v = np.asarray(v, f32); n2 = f32(f32(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])
    return v / np.sqrt(n2) if n2 > 0 else v
def dot3(a, b_): return f32(f32(a[0] * b_[0] + a[1] * b_[1]) + a[2] * b_[2])
mpmath.mp.prec = 200
for k in bad:
    p = P[k]
    print('patch', k, 'stage gpu/oracle', int(b.stage[k]), p.stage, 'nevals', int(b.nevals[k]), p.nevals, 'fmin', float(b.fmin[k]), p.fmin,
          'n_images', int(b.n_images[k]), p.n_images)
    print('   center gpu', b.center[k][:3], 'oracle', np.array(p.center[:3], f32), ' x gpu', b.x[k], 'oracle', list(p.x[:]))
    # the start point's x[1] (parametersFromCenterNorm, PatchOptimizer.cpp:416-446) for the reference image the ORACLE ended with
    # (the start point is taken after setRefImage; if the lists differ already the difference starts earlier)
    v0 = int(p.images[0]) if p.n_images > 0 else int(seeds.images[k, 0])
    vw = scene.views[v0]
    cam = api.camera_from_nvm(vw.f, vw.q, vw.c, vw.width, vw.height)
    X, Y, Z = nrm(list(cam.xaxis)), nrm(list(cam.yaxis)), nrm(list(cam.zaxis))
    n = np.asarray(seeds.normal[k][:3], f32)
    fx, fy, fz = dot3(X, n), dot3(Y, n), dot3(Z, n)
    x2 = float(np.arcsin(f32(fy)))
    t = max(-1.0, min(1.0, x2)); cosb = f32(math.cos(t))
    if cosb == 0: print('   cosb == 0'); continue
    cosa = float(f32(-fz) / cosb); u = max(-1.0, min(1.0, cosa))
    a_host = math.acos(u)
    a_cr = float(mpmath.acos(mpmath.mpf(u)))
    print('   start x[1]: acos(%r) host libm %r correctly rounded %r  -> %s' % (u, a_host, a_cr, 'DIFFER (the known class)' if a_host != a_cr else 'equal'))
