"""Tick-by-tick comparison of the fused (k_solve_backsub) and the two-launch back-substitution (development aid).
usage: python scripts/fuse_diff.py [C F ticks]"""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
import multicam_calibration_amd as m

C, F, T = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (6, 10000, 60)
p = m.synth.make_problem(C, F, seed=5)
x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
traj = {}
for mode in ("1", "0"):
    os.environ["MCBA_FUSE_BACKSUB"] = mode
    prob = m.ops.Problem(p["uvs"], p["obj"])
    lm = m.solver.LevenbergMarquardt(prob, ftol=0.0, xtol=0.0, gtol=0.0, reduced_solver="device", depth=2)
    lm.start(x0)
    rows = []
    for t in range(T):
        lm.iterate(always_linearize=True)
        if os.environ.get("FUSE_DIFF_SYNC", "1") == "1":
            prob.synchronize()
            rows.append((prob.get_params(0).copy(), prob.get_params(1).copy(), lm.cost, lm.lam))
    prob.synchronize()
    lm.finalize()
    rows.append((prob.get_params(0).copy(), prob.get_params(1).copy(), lm.cost, lm.lam))
    traj[mode] = rows
    traj["h" + mode] = np.array(lm.history)
    prob.close()
n = 12 * C
ha, hb = traj["h1"], traj["h0"]
print("history rows", ha.shape, hb.shape)
m_ = min(len(ha), len(hb))
bad = np.nonzero((ha[:m_] != hb[:m_]).any(axis=1))[0]
if bad.size:
    t = bad[0]
    print("first differing history row %d:" % t)
    print("  fused  ", np.array2string(ha[t], precision=17))
    print("  unfused", np.array2string(hb[t], precision=17))
    if t > 0:
        print("  previous row (equal)", np.array2string(ha[t - 1], precision=17))
else:
    print("histories identical")
for t, (a, b) in enumerate(zip(traj["1"], traj["0"])):
    for s in (0, 1):
        d = a[s] != b[s]
        if d.any():
            idx = np.nonzero(d)[0]
            fr = np.unique((idx[idx >= n] - n) // 6)
            print("tick %d slot %d: %d elements differ (camera part %d, frames %d: blocks %s), max abs %.3e; cost %.17g vs %.17g, lambda %g vs %g"
                  % (t, s, d.sum(), (idx < n).sum(), fr.size, np.unique(fr // 64)[:12], np.abs(a[s] - b[s]).max(), a[2], b[2], a[3], b[3]))
            break
    else:
        continue
    break
else:
    print("identical over %d ticks" % len(traj["1"]))
