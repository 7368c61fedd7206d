import sys, ctypes as C, numpy as np, torch
sys.path.insert(0,'.')
from hpmvs_amd import api, synth
api.LIB_PATH='/tmp/libhpmvs_prof.so' if len(sys.argv)<2 else sys.argv[1]
L=api.lib()
dev=torch.device('cuda',0)
V,W_,H_,N_=(int(sys.argv[2]),int(sys.argv[3]),int(sys.argv[4]),int(sys.argv[5])) if len(sys.argv)>5 else (12,1920,1080,100000)
scene=synth.make_scene(V,W_,H_,n_waves=24,device=dev)
g=api.Scene(scene)
seeds=synth.make_seeds(scene,N_,start_level=4,max_images=32)
b=api.Batch.from_seeds(seeds)
api.optimize_batch(g,b)
b=api.Batch.from_seeds(seeds)
api.optimize_batch(g,b)
out=(C.c_ulonglong*68)()
L.hpmvs_debug_counters.argtypes=[C.c_void_p,C.POINTER(C.c_ulonglong)]
L.hpmvs_debug_counters(g.h,out)
tA,tB,nr,nq=[int(v) for v in out[:4]]
names=['trsbox','altmov','update','bq_step','L20','L60(trsbox+)','L90(shift)','L210(altmov+)','L230(vlag,beta)','L360','L650','L680','L720','-']
for i,nm in enumerate(names):
    tot,cnt=int(out[4+2*i]),int(out[5+2*i])
    print(nm,'lane-calls',cnt,'avg cycles per call %.0f'%(tot/max(cnt,1)))
for i,nm in [(23,'after_eval'),(13,'B:prep'),(14,'B:taps+stage'),(15,'B:means+var'),(20,'B:normalise+products'),(21,'B:ncc sum'),(22,'B:deliver'),(24,'list services (wave passes)'),(25,'region: lock wait'),(27,'region: page in'),(26,'region: page out + release'),(16,'finish (colour, stores)'),(17,'calc_patch_axes'),(18,'trsbox: one pass of the CG loop'),(19,'trsbox: one pass of the boundary loop')]:
    tot,cnt=int(out[4+2*i]),int(out[5+2*i]); print(nm,'wave-passes',cnt,'avg cycles %.0f'%(tot/max(cnt,1)),'total %.3g'%tot)
print('trsbox per-lane: calls',int(out[4+59]),'CG iters/call %.2f'%(int(out[4+56])/max(int(out[4+59]),1)),'ALT iters/call %.2f'%(int(out[4+57])/max(int(out[4+59]),1)),'angle steps per ALT iter %.1f'%(int(out[4+58])/max(int(out[4+57]),1)))
print('photometric batches', int(out[4+62]), 'sum of longest request per batch', int(out[4+60]), 'sum of ideal passes ceil(grabs/9)', int(out[4+61]), 'long (>8 grabs) setINCCs requests', int(out[4+63]))
print('kernel ms',g.last_kernel_ms(),'phaseA cycles',tA,'phaseB cycles',tB,'rounds',nr,'requests',nq)
print('A share %.2f  cycles/round A %.0f  B per request %.0f  req/round %.2f'%(tA/(tA+tB), tA/nr, tB/max(nq,1), nq/nr))
print('evals mean', b.nevals[b.nevals>0].mean(), 'ok', b.ok.mean(), 'nevals hist', np.percentile(b.nevals,[50,90,99,100]))
