"""k_gram and LM-iteration time over shapes that are not one round of the wavefront slots (round 4) -> one JSON object.
usage: python scripts/gram_shapes.py [C,F,rows,cols[;...]]      env MCBA_GRAM_SPLIT / MCBA_GRAM_NPW select the launch variant,
SHAPES_FIXED_INTRINSICS=1 holds the intrinsics fixed (camera block 6 wide)"""
import json
import os
import sys
import time

sys.path.insert(0, ".")
import multicam_calibration_amd as m

SHAPES = ((2, 50, 6, 9), (6, 1000, 6, 9), (6, 2130, 5, 7), (6, 5000, 6, 9), (6, 10000, 6, 9), (6, 12500, 6, 9), (24, 6250, 10, 20))
if len(sys.argv) > 1:
    SHAPES = tuple(tuple(int(v) for v in sh.split(",")) for sh in sys.argv[1].split(";"))
out = {"env": {k: v for k, v in os.environ.items() if k.startswith("MCBA_") or k.startswith("SHAPES_")}}
for C, F, rows, cols in SHAPES:
    p = m.synth.make_problem(C, F, rows=rows, cols=cols, seed=0)
    x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    prob = m.ops.Problem(p["uvs"], p["obj"])
    if os.environ.get("SHAPES_FIXED_INTRINSICS"):   # BASELINE configs[1]: the 6-wide camera block
        assert prob.set_camera_block(6)
    lm = m.solver.LevenbergMarquardt(prob, ftol=0.0, xtol=0.0, gtol=0.0, reduced_solver="device", depth=2)
    lm.start(x0)
    for _ in range(60):
        lm.iterate(always_linearize=True)
    prob.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        K = 100
        for _ in range(K):
            lm.iterate(always_linearize=True)
        prob.synchronize()
        best = min(best, (time.perf_counter() - t0) / K)
    prob.profile_enable(True)
    for _ in range(30):
        lm.iterate(always_linearize=True)
    prof = {k: round(1e3 * ms / n, 2) for k, (ms, n) in prob.profile_read().items() if n}
    prob.profile_enable(False)
    out["%dx%dx%d" % (C, F, rows * cols)] = {"us_per_iteration": round(best * 1e6, 1), "kernels_us_by_hip_events": prof, "cost": lm.cost}
    prob.close()
    del p
print(json.dumps(out, indent=1))
