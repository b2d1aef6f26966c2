"""LM iteration time and per-kernel times at the shapes DESIGN.md quotes next to the bench workload -> one JSON object:
2 x 50 (configs[0]), 6 x 1 000 with the intrinsics held fixed (configs[1]) and free, the reference tutorial's 6 x 2 130 x 35, 6 x 5 000, 6 x 10 000 (configs[2]),
6 x 12 500 (a configs[3] shard), 24 x 6 250 x 200 (a configs[4] shard), 6 x 100 000 (configs[3] on one GPU).  A fifth number 1 in a shape = intrinsics held fixed.
usage: python scripts/other_shapes.py > profiles/roundN/other_shapes_TAG.json"""
import json
import sys
import time

sys.path.insert(0, ".")
import multicam_calibration_amd as m

import os

out = {}
SHAPES = ((2, 50, 6, 9), (6, 1000, 6, 9, 1), (6, 1000, 6, 9), (6, 2130, 5, 7), (6, 5000, 6, 9), (6, 10000, 6, 9), (6, 12500, 6, 9), (24, 6250, 10, 20), (6, 100000, 6, 9))
if os.environ.get("MCBA_SHAPES"):  # e.g. MCBA_SHAPES="6,12500,6,9;24,6250,10,20" (under rocprofv3: one shape per run)
    SHAPES = tuple(tuple(int(v) for v in sh.split(",")) for sh in os.environ["MCBA_SHAPES"].split(";"))
for shape in SHAPES:
    C, F, rows, cols = shape[:4]
    fixed = len(shape) > 4 and shape[4]
    p = m.synth.make_problem(C, F, rows=rows, cols=cols, seed=0)
    x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    prob = m.ops.Problem(p["uvs"], p["obj"])
    if fixed:
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
    out["%dx%dx%d%s" % (C, F, rows * cols, "-fixed-intrinsics" if fixed else "")] = {"us_per_iteration": round(best * 1e6, 1), "it_per_s": round(1.0 / best, 1), "kernels_us_by_hip_events": prof, "cost": lm.cost}
    prob.close()
    del p
print(json.dumps(out, indent=1))
