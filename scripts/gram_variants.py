"""k_gram fused vs split-role at a given shape (development aid)."""
import os
import sys

sys.path.insert(0, ".")
import multicam_calibration_amd as m

C, F = int(sys.argv[1]), int(sys.argv[2])
rows, cols = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (6, 9)
p = m.synth.make_problem(C, F, rows=rows, cols=cols, seed=0)
x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
for mode in ("0", "1"):
    os.environ["MCBA_GRAM_SPLIT"] = mode
    prob = m.ops.Problem(p["uvs"], p["obj"])
    prob.set_params(0, x0)
    prob.profile_enable(True)
    for it in range(12):
        prob.linearize(0)
        if it == 1:
            prob.profile_read()
    prob.synchronize()
    ms, n = prob.profile_read()["k_gram"]
    print("C=%d F=%d N=%d split=%s: k_gram %.1f us" % (C, F, rows * cols, mode, 1e3 * ms / n), flush=True)
    prob.close()
