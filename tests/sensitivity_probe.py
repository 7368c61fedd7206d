"""Not a test (not collected): measures how far the reference refinement moves under a tiny relative
perturbation of the objective (CPU only; product BOBYQA state machine + oracle objective).
Numbers quoted in DESIGN.md "Parity"."""
import ctypes as C, numpy as np, sys, math, time
sys.path.insert(0,'/root/repo')
from oracle import oracle as orc
from hpmvs_amd import synth
L = orc.lib()
H = C.CDLL('/root/repo/tests/native/libbq_host.so')
H.bq3_minimize.argtypes=[orc.OBJ_FN, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_int]
sc = synth.make_scene(3, 640, 480, n_waves=24)
osc = orc.OracleScene(sc)
INF=float('inf'); lb=[-INF,-23.99999,-23.99999]; ub=[INF,23.99999,23.99999]
seeds = synth.make_seeds(sc, 200, start_level=2)
P = orc.patches_from_seeds(seeds)
rng=np.random.default_rng(1)
for eps in (1e-7, 1e-9, 1e-12):
    dd=[];da=[];df=[];ne=[]
    for k in range(len(P)):
        p=P[k]
        if p.n_images<3: continue
        x0 = orc.initial_parameters(osc,p)
        outs=[]
        for pert in (0,1):
            def cb(n,x,g,d):
                v=orc.objective_at(osc,p,[x[0],x[1],x[2]])
                if pert and v<2.0: v*= (1+eps*rng.uniform(-1,1))
                return v
            x=(C.c_double*3)(*x0); l=(C.c_double*3)(*lb); u=(C.c_double*3)(*ub); mf=C.c_double(); nev=C.c_int()
            rc=H.bq3_minimize(orc.OBJ_FN(cb),None,x,l,u,1e-7,1000,C.byref(mf),C.byref(nev),None,0)
            outs.append((rc,list(x),mf.value,nev.value))
        if outs[0][0] in (1,4) and outs[1][0] in (1,4):
            a,b=outs[0][1],outs[1][1]
            dd.append(abs(a[0]-b[0])/p.scale); da.append(max(abs(a[1]-b[1]),abs(a[2]-b[2]))*3.75); df.append(abs(outs[0][2]-outs[1][2])); ne.append(outs[0][3])
    dd=np.array(dd);da=np.array(da)
    print('eps',eps,'n',len(dd),'same basin (dd<.25 & ang<2deg):',np.mean((dd<0.25)&(da<2.0)),'median dd',np.median(dd),'p90 dd',np.percentile(dd,90),'median dang',np.median(da),'mean |df|',np.mean(df))
