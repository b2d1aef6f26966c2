"""k_solve_cam phase stamps (build with MCBA_HIPCC_FLAGS=-DMCBA_SOLVE_TIMING) and launch time.
usage: python scripts/solve_time.py [cameras[:6] ...]      (MCBA_LIB = an alternative build; ":6" = the 6-wide camera block)
Beyond 9 cameras (right-looking variant) the stamps of slots 27 / 29 / 30 are the backward sweep and intervals A / B of block step
k = launch number - 1: MCBA_SOLVE_LAUNCHES=N prints them for the first N - 1 launches.
Round 6: the stamps (and the mcba_debug_* export this script reads) live in profiles/round6/patches/experiments_and_stamps.patch, not in the product sources:
`cd multicam-calibration_amd && git apply -p0 ../profiles/round6/patches/experiments_and_stamps.patch` first."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import multicam_calibration_amd as m
import os
if os.environ.get("MCBA_LIB"):
    m.ops.LIB_PATH = os.environ["MCBA_LIB"]

for arg in sys.argv[1:] or ["6", "24"]:   # "C" or "C:6" (the 6-wide camera block: intrinsics fixed)
    C = int(arg.split(":")[0])
    p = m.synth.make_problem(C, 256, seed=0)
    x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    prob = m.ops.Problem(p["uvs"], p["obj"])
    if arg.endswith(":6"):
        assert prob.set_camera_block(6)
    prob.set_params(0, x0)
    prob.linearize(0)
    prob.build_reduced(1e-3)
    red = prob.get_reduced()
    prob.lm_set_state(float(red["scal"][0]), 1e-3, 2.0, 0)
    prob.lm_auto_config(0.0, 0.0, 0.0, 1e-12, 1e12, None)
    prob.profile_enable(True)
    per_launch = []
    for s in range(1, int(os.environ.get("MCBA_SOLVE_LAUNCHES", "12"))):
        prob.lm_auto_solve(s)
        st = prob.lm_auto_wait(s).copy()
        per_launch.append((int(st[27]), int(st[29]), int(st[30])))
    if os.environ.get("MCBA_SOLVE_LAUNCHES"):
        print("per launch (slots 27, 29, 30):", per_launch)
    prof = prob.profile_read()
    ms, n = prof["k_solve_cam"]
    print("C=%d n=%d: k_solve_cam %.2f us avg; stamps (cycles): %s" % (C, prob.n, 1e3 * ms / n, np.array2string(st[25:31], precision=6)))
    prob.close()
