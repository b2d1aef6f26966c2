"""Full solves on odd rig shapes (development aid): convergence, finite results, restart = fixed point."""
import contextlib, io, sys, time
import numpy as np
sys.path.insert(0, ".")
import multicam_calibration_amd as m

bad = 0
for C, F, rows, cols, miss in [(1, 30, 6, 9, 0.0), (2, 2, 6, 9, 0.0), (3, 64, 2, 2, 0.2), (4, 65, 2, 2, 0.5), (5, 63, 1, 3, 0.3), (9, 40, 3, 3, 0.3), (10, 40, 3, 3, 0.3), (16, 33, 2, 3, 0.2), (17, 33, 2, 3, 0.2), (26, 20, 2, 3, 0.2),
                            (27, 20, 2, 3, 0.2), (33, 20, 2, 3, 0.2), (40, 12, 2, 3, 0.1), (6, 700, 1, 1, 0.0), (6, 200, 1, 2, 0.1)]:
    p = m.synth.make_problem(C, F, rows=rows, cols=cols, seed=7 * C + F, missing=miss)
    kw = dict(n_frames=None, ftol=1e-12, xtol=1e-12, gtol=1e-8, verbose=0, max_nfev=100, return_jac=False)
    t0 = time.time()
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            e, it, po, use, res = m.bundle_adjust(p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], **kw)
            full = np.array(p["poses"], dtype=float)
            full[use] = po
            e1, it1, po1, use1, res1 = m.bundle_adjust(p["uvs"], e, it, p["obj"], full, outlier_threshold=1e30, **kw)
        x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"][use])
        ok = np.isfinite(res.x).all() and res.status > 0 and res1.nfev <= 4 and abs(res1.cost - res.cost) <= 1e-9 * max(res.cost, 1e-30)
        bad += not ok
        print("%s C=%2d F=%3d N=%2d: status %d nfev %d cost %.6g (restart: status %d nfev %d cost %.6g) frames %d  %.2fs" % ("ok " if ok else "BAD", C, F, rows * cols, res.status, res.nfev, res.cost, res1.status, res1.nfev, res1.cost, len(use), time.time() - t0), flush=True)
    except Exception as ex:
        bad += 1
        print("BAD C=%d F=%d N=%d: %s: %s" % (C, F, rows * cols, type(ex).__name__, str(ex)[:300]), flush=True)
print("shapes with a problem:", bad)
