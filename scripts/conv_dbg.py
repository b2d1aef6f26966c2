import sys, io, contextlib
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import multicam_calibration_amd as m
from conftest import problem_from_npz
z = np.load("tests/golden/tight_config1.npz")
uvs, ext, intr, obj, poses = problem_from_npz(z)
for kw in (dict(ftol=0.0, xtol=1e-12, gtol=1e-10), dict(ftol=1e-15, xtol=1e-15, gtol=1e-9)):
    with contextlib.redirect_stdout(io.StringIO()):
        e, i, p_, use, res = m.bundle_adjust(uvs, ext, intr, obj, poses, n_frames=None, verbose=0, max_nfev=200, return_jac=False, **kw)
    cam = res.x[:24].reshape(2, 12); camg = z["s0_x"][:24].reshape(2, 12)
    print(kw, "status", res.status, "nfev", res.nfev, "opt %.2e" % res.optimality, "cost %.15g" % res.cost, "gold %.15g" % float(z["s0_cost"]))
    print("  rel intr:", np.abs(cam[:, :6] - camg[:, :6]).max(0) / np.abs(camg[:, :6]).max(0))
    for h in res.lm["history"][-8:]:
        print("   nfev %d dF %.3e ratio %.3f lam %.2e step %.2e" % (h[0], h[1] - h[2], h[4], h[5], h[6]))
