"""k_gram time vs number of board points / frames, both variants (development aid)."""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
import multicam_calibration_amd as m

for (C, F, rows, cols) in [(6, 10000, 6, 9), (6, 10000, 3, 9), (6, 10000, 1, 2), (6, 10000, 12, 9), (6, 40000, 6, 9), (6, 40000, 1, 2)]:
    p = m.synth.make_problem(C, F, rows=rows, cols=cols, seed=0)
    x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    out = []
    for mode in ("0", "1"):
        os.environ["MCBA_GRAM_SPLIT"] = mode
        prob = m.ops.Problem(p["uvs"], p["obj"])
        prob.set_params(0, x0)
        for _ in range(3):
            prob.linearize(0)
        prob.profile_enable(True)
        prob.profile_read()
        for _ in range(10):
            prob.linearize(0)
            prob.cost(0)
        pr = prob.profile_read()
        out.append("split=%s gram %.1f us cost %.1f us" % (mode, 1e3 * pr["k_gram"][0] / pr["k_gram"][1], 1e3 * pr["k_cost"][0] / pr["k_cost"][1]))
        prob.close()
    print("C=%d F=%d N=%d : %s" % (C, F, rows * cols, " | ".join(out)))
