"""k_gram variants at a given shape (development aid): 0 fused, 1 split roles, 2 fused whole rounds + split-role tail.
usage: python scripts/gram_variants.py C F [rows cols] [modes]"""
import os
import sys

sys.path.insert(0, ".")
import multicam_calibration_amd as m

C, F = int(sys.argv[1]), int(sys.argv[2])
rows, cols = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (6, 9)
modes = sys.argv[5] if len(sys.argv) > 5 else "012"
p = m.synth.make_problem(C, F, rows=rows, cols=cols, seed=0, missing=0.1 if os.environ.get("MISSING") else 0.0)
x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
ref = None
for mode in modes:
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
    red = {k: v.copy() for k, v in prob.get_reduced().items()}
    ref = ref or red
    dS = abs(red["S0"] - ref["S0"]).max() / abs(ref["S0"]).max()
    dr = abs(red["rhs"] - ref["rhs"]).max() / abs(ref["rhs"]).max()
    print("C=%d F=%d N=%d mode=%s: k_gram %.1f us per launch%s  (cost %.15g, S0 rel diff %.1e, rhs %.1e, pairs %g)" % (C, F, rows * cols, mode, 1e3 * ms / n * (2 if mode == "2" else 1), " (2 launches)" if mode == "2" else "", red["scal"][0], dS, dr, red["scal"][1]), flush=True)
    prob.close()
