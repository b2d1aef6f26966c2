"""Golden fixture for `bounds=` at a BASELINE size (6 cameras x 1 000 frames x 54 points), FROM THE REFERENCE ITSELF (VERDICT r5 task 4).

Run in the build container only:      python tests/golden/make_golden_bounds_large.py

Same protocol as make_golden_bounds.py (whose dense active-set polish stops at toy sizes), with sparse algebra:
  1. the problem of tight_6x1000.npz (synth.make_problem(6, 1000, seed=0, perturb_seed=1), complete detections) and its unconstrained tight
     optimum place the bounds: k1 / k2 of two cameras and a focal length half-way between the start and that optimum (feasible at the start,
     violated at the unconstrained optimum), the z-translation of 20 board poses a tenth of the way -- more than the 6-dimensional gauge
     freedom can absorb --, plus wide boxes that stay inactive;
  2. the unmodified reference's `bundle_adjust(..., bounds=(lo, hi))` (scipy's trf_bounds + LSMR through its **opt_kwargs, with an analytic
     sparse `jac=` and tight tolerances, as make_golden_tight_large.py does for the unconstrained case);
  3. its point is polished by a sparse active-set Gauss-Newton iteration on the REFERENCE's residual function until the KKT residual is at
     round-off level;
  4. certificate, independent of the analytic Jacobian: scipy's 3-point finite-difference gradient of the reference's robust cost (sparsity
     from the reference's own bundle_adjustment_sparsity) vanishes on the free coordinates and points outward on the active ones.
Writes tight_bounds_6x1000.npz: lo, hi, x, cost, active_mask (the inputs are regenerated from the seed; checksum stored)."""
import contextlib
import io
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

from make_golden import load_reference  # noqa: E402


def main(C=6, F=1000):
    from multicam_calibration_amd import synth
    from oracle import ba_oracle as orc
    from scipy.optimize._lsq.common import find_active_constraints
    from scipy.optimize._numdiff import approx_derivative, group_columns

    geo, ba = load_reference()
    q = synth.make_problem(C, F, seed=0, perturb_seed=1)
    obj = q["obj"]
    with contextlib.redirect_stdout(io.StringIO()):
        use = ba.bundle_adjust(q["uvs"], q["extrinsics"], q["intrinsics"], obj, q["poses"], n_frames=None, max_nfev=1, verbose=0)[3]
    uvs = q["uvs"][:, use]
    x0 = ba.serialize_params(q["extrinsics"], q["intrinsics"], q["poses"][use])
    z = np.load(os.path.join(HERE, f"tight_{C}x{F}.npz"))
    assert np.array_equal(z["s0_use"], use)
    xu = z["s0_x"]
    n, nc = x0.size, 12 * C
    lo, hi = np.full(n, -np.inf), np.full(n, np.inf)
    rng = np.random.default_rng(5)
    pose_frames = np.sort(rng.choice(len(use), 20, replace=False))
    bounded = [12 * 1 + 4, 12 * 1 + 5, 12 * 4 + 4, 12 * 4 + 5, 12 * 2 + 0] + [nc + 6 * int(f) + 5 for f in pose_frames]
    for i in bounded:
        b = x0[i] + (0.5 if i < nc else 0.1) * (xu[i] - x0[i])
        if xu[i] > x0[i]:
            hi[i] = b
        else:
            lo[i] = b
    for c in range(C):   # ... and a wide box on every focal length / principal point, inactive
        for k in range(4):
            i = 12 * c + k
            lo[i] = lo[i] if np.isfinite(lo[i]) else x0[i] - 500.0
            hi[i] = hi[i] if np.isfinite(hi[i]) else x0[i] + 500.0
    assert np.all(lo < hi) and np.all(x0 >= lo) and np.all(x0 <= hi)

    t0 = time.perf_counter()
    jac = lambda x, u, o: orc.jacobian_csr(x, u, o)
    with contextlib.redirect_stdout(io.StringIO()):
        ext, intr, poses, use2, res = ba.bundle_adjust(q["uvs"], q["extrinsics"], q["intrinsics"], obj, q["poses"], n_frames=None, bounds=(lo, hi), jac=jac,
                                                       ftol=1e-13, xtol=1e-13, gtol=1e-10, max_nfev=80, verbose=0)
    assert np.array_equal(use, use2)
    print("reference run: %.1fs cost %.15g nfev %d status %d optimality %.2e active %d" % (time.perf_counter() - t0, res.cost, res.nfev, res.status, res.optimality, int((res.active_mask != 0).sum())), flush=True)

    def lin(xv):
        fv = ba.residuals(xv, uvs, obj)
        js, fs_ = orc.robust_scales(fv)
        Jd = sp.diags(js) @ orc.jacobian_csr(xv, uvs, obj)
        return Jd.tocsc(), fs_, Jd.T @ fs_

    def kkt(xv, g):
        on_lo, on_hi = xv <= lo, xv >= hi
        work = (on_lo & (g > 0)) | (on_hi & (g < 0))
        return work, float(np.abs(g[~work]).max())

    xp = np.clip(res.x, lo, hi)
    snap = find_active_constraints(xp, lo, hi, rtol=1e-9)   # scipy stays strictly inside: put the numerically active ones ON their bounds
    xp[snap == -1], xp[snap == 1] = lo[snap == -1], hi[snap == 1]
    Jd, fs_, g = lin(xp)
    for it in range(40):
        work, r = kkt(xp, g)
        print(f"    polish {it}: KKT residual {r:.3e}, working set {int(work.sum())}", flush=True)
        if r < 2e-9:
            break
        free = np.nonzero(~work)[0]
        Jf = Jd[:, free]
        H = (Jf.T @ Jf).tocsc()
        d = H.diagonal()
        H = H + sp.diags(1e-9 * np.where(d > 0, d, 1.0))   # (only fixes the gauge directions: the stationary point does not depend on it)
        step = np.zeros(n)
        step[free] = spla.splu(H.tocsc()).solve(-g[free])
        for k in range(8):
            xn = np.clip(xp + step * 0.5**k, lo, hi)
            Jn, fn, gn = lin(xn)
            if kkt(xn, gn)[1] < r:
                xp, Jd, fs_, g = xn, Jn, fn, gn
                break
        else:
            break
    work, r = kkt(xp, g)
    cost = orc.robust_cost(ba.residuals(xp, uvs, obj))
    active = find_active_constraints(xp, lo, hi, rtol=1e-8)
    A = ba.bundle_adjustment_sparsity(uvs)
    J3 = approx_derivative(lambda x: ba.residuals(x, uvs, obj), xp, method="3-point", sparsity=(A, group_columns(A))).tocsr()
    f = ba.residuals(xp, uvs, obj)
    g_fd = J3.T @ (f * orc.loss_rho(f**2, "soft_l1")[1])
    outward = bool(np.all(g_fd[active == -1] > 0) and np.all(g_fd[active == 1] < 0))
    print("polished: cost %.15g (reference run %.15g, unconstrained %.15g)  KKT residual %.2e  FD gradient on the free set %.2e  active %d (lower %d, upper %d), multipliers point outward: %s"
          % (cost, res.cost, float(z["s0_cost"]), r, np.abs(g_fd[active == 0]).max(), int((active != 0).sum()), int((active == -1).sum()), int((active == 1).sum()), outward), flush=True)
    print("active indices", np.nonzero(active)[0], "camera block ends at", nc)
    assert outward and np.array_equal(active != 0, work)
    assert (active[:nc] != 0).sum() >= 1 and (active[nc:] != 0).sum() >= 1, "want active bounds on camera parameters AND on board poses"
    np.savez_compressed(os.path.join(HERE, f"tight_bounds_{C}x{F}.npz"), use=use, lo=lo, hi=hi, x=xp, cost=np.array(cost), active_mask=active, kkt_residual=np.array(r),
                        fd_grad_free_inf=np.array(np.abs(g_fd[active == 0]).max()), ref_run_cost=np.array(res.cost), ref_run_nfev=np.array(res.nfev), ref_run_status=np.array(res.status),
                        ref_run_active_mask=res.active_mask, uvs_checksum=np.array(np.nansum(q["uvs"])), shape=np.array([C, F, obj.shape[0]]), bounded=np.array(bounded))
    print(f"tight_bounds_{C}x{F}.npz written")


if __name__ == "__main__":
    main()
