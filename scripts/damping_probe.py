"""Evaluations to convergence for different damping schedules (lam0, floor of Nielsen's factor) over a set of problems -> table
(NOTES_round3 section 5).  usage: python scripts/damping_probe.py"""
import contextlib, io, sys
import numpy as np
sys.path.insert(0, ".")
import multicam_calibration_amd as m

def problems():
    yield "6x10000 default ftol", m.synth.make_problem(6, 10000, seed=0), dict()
    yield "6x10000 tight", m.synth.make_problem(6, 10000, seed=0), dict(ftol=1e-12, xtol=1e-12, gtol=1e-8, max_nfev=100)
    yield "6x1000 fixed intr tight", m.synth.make_problem(6, 1000, seed=0), dict(fix_intrinsics=True, ftol=1e-12, xtol=1e-12, gtol=1e-8, max_nfev=100)
    yield "3x70 missing tight", m.synth.make_problem(3, 70, seed=5, missing=0.2), dict(ftol=1e-12, xtol=1e-12, gtol=1e-8, max_nfev=100)
    yield "2x50 cauchy tight", m.synth.make_problem(2, 50, seed=3), dict(loss="cauchy", ftol=1e-12, xtol=1e-12, gtol=1e-8, max_nfev=200)
    yield "2x50 huber outliers", m.synth.make_problem(2, 50, seed=4, outlier_frames=5), dict(loss="huber", ftol=1e-12, xtol=1e-12, gtol=1e-8, max_nfev=200, outlier_threshold=1e9)
    p = m.synth.make_problem(6, 300, seed=86); p["poses"][::7, 3:] += 40.0
    yield "6x300 bad start tight", p, dict(ftol=1e-12, xtol=1e-12, gtol=1e-9, max_nfev=200, outlier_threshold=1e9)
    yield "24x100x200 tight", m.synth.make_problem(24, 100, rows=10, cols=20, seed=2), dict(ftol=1e-12, xtol=1e-12, gtol=1e-8, max_nfev=100)
    yield "9x20 missing 0.4", m.synth.make_problem(9, 20, seed=6, missing=0.4), dict(ftol=1e-10, xtol=1e-10, gtol=1e-8, max_nfev=100)

combos = [(1e-4, 1 / 3), (1e-2, 1 / 3), (1e-2, 0.2), (1e-2, 0.1), (1e-2, 0.05), (1e-3, 0.1), (1e-1, 0.1)]
print("%-26s" % "problem" + "".join("  lam0=%g,fl=%.2f" % c for c in combos))
for name, p, kw in problems():
    row = "%-26s" % name
    for lam0, fl in combos:
        np.random.seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            res = m.bundle_adjust(p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=None, verbose=0, return_jac=False, lam0=lam0, dec_floor=fl, **kw)[4]
        row += "  %3d/%3d s%d %.9g" % (res.nfev, res.lm["iterations"], res.status, res.cost)
    print(row, flush=True)
