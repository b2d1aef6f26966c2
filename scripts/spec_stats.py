"""How often would 'accepted, lambda' = max(lambda/3, lambda_min)' predict the LM decision? (development aid)"""
import sys

import numpy as np

sys.path.insert(0, ".")
import multicam_calibration_amd as m

C, F = 6, int(sys.argv[1]) if len(sys.argv) > 1 else 10000
p = m.synth.make_problem(C, F, seed=0)
x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
prob = m.ops.Problem(p["uvs"], p["obj"])
lm = m.solver.LevenbergMarquardt(prob, ftol=0.0, xtol=0.0, gtol=0.0)
lm.start(x0)
lam_prev = lm.lam
hits = []
for it in range(230):
    lm.iterate()
    spec = max(lam_prev / 3.0, lm.lam_min)
    hit = lm.accepted and abs(lm.lam - spec) <= 1e-15 * spec
    hits.append(hit)
    if it < 40 or not hit:
        h = lm.history[-1]
        print(it, "acc" if lm.accepted else "REJ", "ratio %.3g lam_used %.3g lam_next %.3g spec %.3g %s" % (h[4], h[5], lm.lam, spec, "hit" if hit else "MISS"))
    lam_prev = lm.lam
hits = np.array(hits)
print("hit rate first 30: %.2f ; after: %.3f" % (hits[:30].mean(), hits[30:].mean()))
prob.close()
