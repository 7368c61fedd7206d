"""How far do refined patches move when Eigen's 4-float reductions use another evaluation order?  CPU only (oracle +
genuine NLopt): left-to-right (what the HIP path restates) vs the SSE2 packet order and the SSE3 haddps order.
Round 3: also Eigen 3.0/3.1's vector / scalar = multiplication by the reciprocal.
Output committed as profiles/r03/eigen_variant_sensitivity.json (r02: eigen_order_sensitivity.json); DESIGN.md section 5 quotes it."""
import sys, os, json, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hpmvs_amd import synth
from oracle import oracle as orc
def run(scene, seeds, order, div_mode=0):
    orc.set_dot4_order(order); orc.set_vec_div_mode(div_mode)
    osc = orc.OracleScene(scene)   # (the camera tables are built under the same settings)
    P = orc.patches_from_seeds(seeds)
    orc.optimize_batch(osc, P, which=orc.best_optimizer(), n_threads=os.cpu_count() or 8)
    orc.set_dot4_order(0); orc.set_vec_div_mode(0)
    return P
def cmp(P0, P1, seeds):
    ok0 = np.array([p.stage==0 for p in P0]); ok1=np.array([p.stage==0 for p in P1])
    both = np.nonzero(ok0&ok1)[0]
    unrob = lambda f: f/(1-3*f)
    dncc = np.array([abs(unrob(P0[k].fmin)-unrob(P1[k].fmin)) for k in both])
    dd = np.array([np.linalg.norm(np.array(P0[k].center[:3])-np.array(P1[k].center[:3]))/float(seeds.scale[k]) for k in both])
    ang = np.array([np.degrees(np.arccos(np.clip(np.dot(np.array(P0[k].normal[:3]),np.array(P1[k].normal[:3])),-1,1))) for k in both])
    ident = np.array([list(P0[k].center)==list(P1[k].center) and list(P0[k].normal)==list(P1[k].normal) for k in both])
    return dict(n=len(P0), ok0=int(ok0.sum()), decision_agreement=float((ok0==ok1).mean()), both=len(both),
                mean_dncc=float(dncc.mean()), p95_dncc=float(np.percentile(dncc,95)), max_dncc=float(dncc.max()),
                same_basin=float(((dd<0.25)&(ang<2)).mean()), bit_identical=float(ident.mean()),
                median_dd=float(np.median(dd)), p95_dd=float(np.percentile(dd,95)))
scene = synth.make_scene(12, 960, 540, n_waves=24)
seeds = synth.make_seeds(scene, 4000, start_level=3, max_images=32)
P0 = run(scene, seeds, 0)
out = {}
for name, o, dm in (("sse2_order", 1, 0), ("sse3_order", 2, 0), ("reciprocal_division_eigen30_31", 0, 1), ("sse2_order+reciprocal_division", 1, 1)):
    out[name] = cmp(P0, run(scene, seeds, o, dm), seeds)
    print(name, json.dumps(out[name]))
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/eigen_variants.json", "w"), indent=1)
