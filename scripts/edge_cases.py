"""Degenerate inputs through the full solver (development aid): no crash, no hang, finite results."""
import contextlib, io, sys
import numpy as np
sys.path.insert(0, ".")
import multicam_calibration_amd as m

def run(tag, p, **kw):
    with contextlib.redirect_stdout(io.StringIO()):
        e, i, po, use, res = m.bundle_adjust(p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=None, verbose=0, max_nfev=60, return_jac=False, **kw)
    print(tag, "status", res.status, "nfev", res.nfev, "cost %.6g" % res.cost, "finite", bool(np.isfinite(res.x).all()), "frames", len(use), flush=True)

p = m.synth.make_problem(3, 40, seed=1)
p["uvs"][2] = np.nan                      # a camera that never sees the board
run("blind camera", p)
p = m.synth.make_problem(2, 3, seed=2)   # fewer frames than a wavefront, tiny problem
run("3 frames", p)
p = m.synth.make_problem(2, 65, seed=3, rows=1, cols=2)   # 2 points per board: rank-deficient frames
run("2 points", p)
p = m.synth.make_problem(2, 30, seed=4)
p["uvs"][:, 5] = np.nan                   # a frame nobody sees (dropped by the pre-filter)
run("empty frame", p)
p = m.synth.make_problem(2, 30, seed=5)
p["poses"][:] += 300.0                    # terrible start: first steps rejected, damping grows
run("bad start", p)
p = m.synth.make_problem(9, 70, seed=6, missing=0.5)      # 9 cameras: largest LDS-resident reduced solve
run("9 cameras", p)
p = m.synth.make_problem(10, 70, seed=7, missing=0.5)     # 10 cameras: reduced solve from the L2 scratch
run("10 cameras", p)
p = m.synth.make_problem(2, 30, seed=8)
run("huber", p, loss="huber", f_scale=0.5)
run("arctan", p, loss="arctan", f_scale=2.0)
run("linear", p, loss="linear")
