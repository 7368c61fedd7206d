#!/usr/bin/env python3
"""Generates tests/golden/g5_bobyqa_selftest.json IN THE BUILD CONTAINER: the reference's GENUINE NLopt 2.4.2 BOBYQA
(oracle/_ref) on the 64 analytic problems of tests/selftest_problems.py.  Data only: per problem the nlopt result
code, x*, f*, evaluation count and a SHA-256 of the full (x, f) evaluation trace."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import selftest_problems as sp

probs = sp.problems()
ref = sp.reference_run(probs)
_, resc = sp.host_state_machine_run(probs, trace_cap=1)
out = [dict(result=r[0], x=r[1], minf=r[2], nevals=r[3], trace_sha256=hashlib.sha256(r[4].tobytes()).hexdigest(),
            rescue_calls=int(c)) for r, c in zip(ref, resc)]
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "g5_bobyqa_selftest.json"), "w"), indent=0)
print(len(out), "problems;", sum(1 for o in out if o["rescue_calls"]), "enter rescue_; results",
      sorted({o["result"] for o in out}))
