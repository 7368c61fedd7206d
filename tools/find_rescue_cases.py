"""Random search (host only) for parameters of the selftest objectives on which BOBYQA enters rescue_."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import selftest_problems as sp
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
found = []
P = lambda *a: list(a) + [0.0] * (8 - len(a))
tries = 0
while len(found) < 10 and tries < 60000:
    tries += 1
    kind = 0 if rng.random() < 0.6 else 3
    a = rng.uniform(-3, 3, 3)
    if kind == 0:
        s = 10.0 ** rng.uniform(-6, 6, 3)
        noise = 10.0 ** rng.uniform(-16, -3)
        p = [*a, *s, noise]
    else:
        p = [a[0], a[1], 0.0, 10.0 ** rng.uniform(-1, 6)]
    x0 = [0.0, float(rng.uniform(-20, 20)), float(rng.uniform(-20, 20))]
    pr = (kind, P(*p), x0, sp.LB, sp.UB)
    res, resc = sp.host_state_machine_run([pr], trace_cap=1)
    if resc[0] > 0:
        found.append((kind, [float(v) for v in p], x0, resc[0], res[0][0], res[0][3]))
        print(found[-1], flush=True)
print('tries', tries)
print('RESCUE = [')
for k, p, x0, *_ in found:
    print('    (%d, %r, %r),' % (k, p, x0))
print(']')
