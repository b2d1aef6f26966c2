"""k_gram variants at a given shape (development aid): 0 fused, 1 split roles, 2 fused whole rounds + split-role tail."""
import os
import sys

sys.path.insert(0, ".")
import multicam_calibration_amd as m

C, F = int(sys.argv[1]), int(sys.argv[2])
rows, cols = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (6, 9)
p = m.synth.make_problem(C, F, rows=rows, cols=cols, seed=0)
x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
ref = None
for mode in ("0", "1", "2"):
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
    prob.build_reduced(1e-3)
    red = prob.get_reduced()
    chk = float(red["scal"][0]), float(abs(red["S0"]).sum())
    ref = ref or chk
    print("C=%d F=%d N=%d mode=%s: k_gram %.1f us per launch%s  (cost %.12g, |S0| rel diff %.1e)" % (C, F, rows * cols, mode, 1e3 * ms / n * (2 if mode == "2" else 1), " (2 launches)" if mode == "2" else "", chk[0], abs(chk[1] - ref[1]) / ref[1]), flush=True)
    prob.close()
