"""Where the stream kernel's time goes (needs a -DSK_STATS build: tools/lib_sk_stats.so)."""
import os, sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
from hpmvs_amd import api, synth
api.LIB_PATH = sys.argv[1] if len(sys.argv) > 1 else 'tools/lib_sk_stats.so'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
views, W, H = (int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (12, 1920, 1080)
os.environ['HPMVS_KERNEL'] = 'stream'
dev = torch.device('cuda', 0)
scene = synth.make_scene(views, W, H, n_waves=24, device=dev)
g = api.Scene(scene)
seeds = synth.make_seeds(scene, n, start_level=4, max_images=32)
b = api.Batch.from_seeds(seeds); api.optimize_batch(g, b)
ms = g.last_kernel_ms()
st = (C.c_ulonglong * 80)()
api.lib().hpmvs_debug_stream_stats.argtypes = [C.c_void_p, C.c_void_p]
api.lib().hpmvs_debug_stream_stats(g.h, st)
st = np.array(st, dtype=np.float64)
Q = ['EVAL', 'AE', 'L60', 'L90', 'L210', 'L230', 'RESC', 'PIPE', 'FETCH']
tot = st[28]
print('kernel ms %.1f  waves*cycles %.3g  evaluations %d  wave-cycles per evaluation %.0f' % (ms, tot, int(b.nevals.sum()), tot / max(1, b.nevals.sum())))
for i, q in enumerate(Q):
    print('%-6s visits %9d  time share %.3f' % (q, st[i], st[16 + i] / tot))
print('failed pops %d share %.3f; idle loops %d share %.3f' % (st[10], st[26] / tot, st[11], st[27] / tot))
print('block visits %d, patches/visit %.2f; pop %.3f pagein %.3f compute %.3f pageout %.3f push %.3f (share of all wave cycles)' % (
    st[46], st[45] / max(1, st[46]), st[40] / tot, st[41] / tot, st[42] / tot, st[43] / tot, st[44] / tot))
print('eval visits %d, requests/visit %.2f; pop %.3f load %.3f eval %.3f store+push %.3f' % (
    st[55], st[54] / max(1, st[55]), st[50] / tot, st[51] / tot, st[52] / tot, st[53] / tot))
ev = st[56:72]
if ev[8] > 0:
    print('evaluator: phase1 %.3f (%d passes, %.0f cycles each); phase2 passes %d (%.1f grabs each): batch %.0f prep %.0f grab %.0f normalize %.0f dot %.0f deliver %.0f cycles per pass' % (
        ev[0] / tot, ev[1], ev[0] / max(1, ev[1]), ev[8], ev[9] / ev[8], ev[2] / ev[8], ev[3] / ev[8], ev[4] / ev[8], ev[5] / ev[8], ev[6] / ev[8], ev[7] / ev[8]))
