"""LM iteration time and per-kernel times at the shapes DESIGN.md quotes next to the bench workload -> one JSON object:
6 x 1 000 (configs[1] size), 6 x 12 500 (a configs[3] shard), 24 x 6 250 x 200 (a configs[4] shard), 6 x 100 000 (configs[3] on one GPU).
usage: python scripts/other_shapes.py > profiles/roundN/other_shapes_TAG.json"""
import json
import sys
import time

sys.path.insert(0, ".")
import multicam_calibration_amd as m

import os

out = {}
SHAPES = ((6, 1000, 6, 9), (6, 10000, 6, 9), (6, 12500, 6, 9), (24, 6250, 10, 20), (6, 100000, 6, 9))
if os.environ.get("MCBA_SHAPES"):  # e.g. MCBA_SHAPES="6,12500,6,9;24,6250,10,20" (under rocprofv3: one shape per run)
    SHAPES = tuple(tuple(int(v) for v in sh.split(",")) for sh in os.environ["MCBA_SHAPES"].split(";"))
for C, F, rows, cols in SHAPES:
    p = m.synth.make_problem(C, F, rows=rows, cols=cols, seed=0)
    x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    prob = m.ops.Problem(p["uvs"], p["obj"])
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
    out["%dx%dx%d" % (C, F, rows * cols)] = {"us_per_iteration": round(best * 1e6, 1), "it_per_s": round(1.0 / best, 1), "kernels_us_by_hip_events": prof, "cost": lm.cost}
    prob.close()
    del p
print(json.dumps(out, indent=1))
