"""Wall time of the user-level call at BASELINE configs[2] (6 x 10 000 x 54): bundle_adjust() from host arrays to the 5-tuple."""
import contextlib, io, sys, time
import numpy as np
sys.path.insert(0, ".")
import multicam_calibration_amd as m

p = m.synth.make_problem(6, 10000, seed=0)
for kw in (dict(return_jac=False), dict(return_jac=True)):
    for rep in range(2):
        np.random.seed(0)
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            e, i, po, use, res = m.bundle_adjust(p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=10000, verbose=0, **kw)
        dt = time.perf_counter() - t0
        print("%s run %d: %.3f s  (status %d, nfev %d, cost %.6g%s)" % (kw, rep, dt, res.status, res.nfev, res.cost, ", jac nnz %d" % res.jac.nnz if kw["return_jac"] else ""), flush=True)
