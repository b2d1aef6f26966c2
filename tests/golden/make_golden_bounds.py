"""Golden fixture for `bounds=` (VERDICT r4 task 5), FROM THE REFERENCE ITSELF.

Run in the build container only:      python tests/golden/make_golden_bounds.py

The unmodified reference's `bundle_adjust(..., bounds=(lo, hi))` -- it forwards **opt_kwargs to scipy's least_squares
(bundle_adjustment.py:301-313), whose TRF is a bounded solver (trf_bounds) -- on BASELINE configs[0] (2 cameras x 50 frames x 54 points),
with bounds that the unconstrained optimum violates: k1 / k2 of the cameras, a focal length, two coordinates of two board poses -- so
that at least two bounds are ACTIVE at the constrained optimum -- plus finite bounds that stay inactive.  The reference's run (analytic
sparse Jacobian and tight tolerances through its **opt_kwargs, as in make_golden_tight_large.py) is then polished by a dense active-set
Gauss-Newton iteration on the REFERENCE's residual function until the KKT residual is at round-off level, and certified independently:
scipy's 3-point finite-difference gradient of the reference's cost must vanish on the free coordinates and point outward on the active
ones.  Writes tight_bounds_config1.npz: inputs, lo, hi, x, cost, active_mask (scipy's find_active_constraints at the optimum)."""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

from make_golden import load_reference, problem_arrays  # noqa: E402


def main(tag="config1"):
    from multicam_calibration_amd import synth
    from oracle import ba_oracle as orc
    from scipy.optimize._lsq.common import find_active_constraints
    from scipy.optimize._numdiff import approx_derivative, group_columns

    geo, ba = load_reference()
    # config1: BASELINE configs[0]; missing3: three cameras, 30 frames, a quarter of the detections and six single scalars missing (the same
    # problems as the unconstrained goldens tight_config1.npz / tight_missing3.npz of make_golden.py --slow, whose optima place the bounds)
    q = synth.make_problem(2, 50, seed=0, perturb_seed=1) if tag == "config1" else synth.make_problem(3, 30, seed=40, missing=0.25, scalar_nans=6, perturb_seed=1)
    C = q["uvs"].shape[0]
    with contextlib.redirect_stdout(io.StringIO()):
        use = ba.bundle_adjust(q["uvs"], q["extrinsics"], q["intrinsics"], q["obj"], q["poses"], n_frames=None, max_nfev=1, verbose=0)[3]
    uvs = q["uvs"][:, use]
    x0 = ba.serialize_params(q["extrinsics"], q["intrinsics"], q["poses"][use])
    xu = np.load(os.path.join(HERE, f"tight_{tag}.npz"))["s0_x"]   # the unconstrained tight optimum of the same problem (make_golden.py --slow)
    assert xu.shape == x0.shape
    n = x0.size
    lo, hi = np.full(n, -np.inf), np.full(n, np.inf)
    # bounds half-way between the start and the unconstrained optimum: feasible at x0, violated at the unconstrained optimum
    # (board poses: ONE bounded pose coordinate can always be evaded through the 6-dimensional gauge freedom -- a rigid motion of all poses
    #  against the cameras costs nothing --, so a dozen of them are bounded, close to the start: more than the gauge can absorb)
    nc = 12 * C
    pose_idx = tuple(nc + 6 * f + 5 for f in range(12)) + (nc + 6 * 20 + 0, nc + 6 * (len(use) - 3) + 4)
    for i in (4, 5, 12 + 4, 12 + 5, 12 + 0) + ((24 + 1, 24 + 5) if C > 2 else ()) + pose_idx:
        b = x0[i] + (0.5 if i < nc else 0.1) * (xu[i] - x0[i])
        if xu[i] > x0[i]:
            hi[i] = b
        else:
            lo[i] = b
    # ... and bounds that stay inactive: a wide box on every focal length / principal point and on the poses' translations of frame 7
    for i in (0, 1, 2, 3, 12 + 1, 12 + 2, 12 + 3):
        lo[i] = min(lo[i], x0[i] - 500.0) if np.isfinite(lo[i]) else x0[i] - 500.0
        hi[i] = hi[i] if np.isfinite(hi[i]) else x0[i] + 500.0
    for i in (nc + 6 * 7 + 3, nc + 6 * 7 + 4, nc + 6 * 7 + 5):
        lo[i], hi[i] = x0[i] - 300.0, x0[i] + 300.0
    assert np.all(lo < hi) and np.all(x0 >= lo) and np.all(x0 <= hi)

    # ---- the reference's own bounded run (scipy trf_bounds + LSMR through its **opt_kwargs)
    jac = lambda x, u, o: orc.jacobian_csr(x, u, o)
    with contextlib.redirect_stdout(io.StringIO()):
        ext, intr, poses, use2, res = ba.bundle_adjust(q["uvs"], q["extrinsics"], q["intrinsics"], q["obj"], q["poses"], n_frames=None, bounds=(lo, hi), jac=jac,
                                                       ftol=1e-15, xtol=1e-15, gtol=1e-12, max_nfev=400, verbose=0)
    assert np.array_equal(use, use2)
    print("reference run: cost %.15g nfev %d status %d optimality %.2e active %d" % (res.cost, res.nfev, res.status, res.optimality, int((res.active_mask != 0).sum())))

    # ---- polish: dense active-set Gauss-Newton on the reference's residuals (robust-rescaled pair, as scipy builds it)
    def grad_at(xv):
        fv = ba.residuals(xv, uvs, q["obj"])
        js, fs_ = orc.robust_scales(fv)
        Jd = orc.jacobian_csr(xv, uvs, q["obj"]).toarray() * js[:, None]
        return Jd, fs_, Jd.T @ fs_

    def kkt(xv, g):
        on_lo, on_hi = xv <= lo, xv >= hi
        work = (on_lo & (g > 0)) | (on_hi & (g < 0))
        return work, np.abs(g[~work]).max()

    xp = np.clip(res.x, lo, hi)
    snap = find_active_constraints(xp, lo, hi, rtol=1e-9)   # scipy stays strictly inside: put the numerically active ones ON their bounds
    xp[snap == -1], xp[snap == 1] = lo[snap == -1], hi[snap == 1]
    Jd, fs_, g = grad_at(xp)
    for it in range(60):
        work, r = kkt(xp, g)
        if r < 1e-9:
            break
        free = ~work
        step = np.zeros(n)
        step[free] = np.linalg.lstsq(Jd[:, free], -fs_, rcond=1e-10)[0]
        for k in range(10):
            xn = np.clip(xp + step * 0.5**k, lo, hi)
            Jn, fn, gn = grad_at(xn)
            if kkt(xn, gn)[1] < r:
                xp, Jd, fs_, g = xn, Jn, fn, gn
                break
        else:
            break
    work, r = kkt(xp, g)
    cost = orc.robust_cost(ba.residuals(xp, uvs, q["obj"]))
    active = find_active_constraints(xp, lo, hi, rtol=1e-8)
    # independent certificate: FD gradient of the reference's robust cost
    A = ba.bundle_adjustment_sparsity(uvs)
    J3 = approx_derivative(lambda x: ba.residuals(x, uvs, q["obj"]), xp, method="3-point", sparsity=(A, group_columns(A))).tocsr()
    f = ba.residuals(xp, uvs, q["obj"])
    _, r1, _ = orc.loss_rho(f**2, "soft_l1")
    g_fd = J3.T @ (f * r1)
    outward = np.all(g_fd[active == -1] > 0) and np.all(g_fd[active == 1] < 0)
    print("polished: cost %.15g (reference run %.15g, unconstrained %.15g)  KKT residual %.2e  FD gradient on the free set %.2e  active %d (lower %d, upper %d), multipliers point outward: %s"
          % (cost, res.cost, orc.robust_cost(ba.residuals(xu, uvs, q["obj"])), r, np.abs(g_fd[active == 0]).max(), int((active != 0).sum()), int((active == -1).sum()), int((active == 1).sum()), outward))
    print("active indices", np.nonzero(active)[0], "camera block ends at", nc)
    assert outward and (active != 0).sum() >= 2 and np.array_equal(active != 0, work)
    assert (active[:nc] != 0).sum() >= 1 and (active[nc:] != 0).sum() >= 1, "want active bounds on camera parameters AND on a board pose"
    same_set = np.array_equal(active, find_active_constraints(np.clip(res.x, lo, hi), lo, hi, rtol=1e-6))
    print("the reference's own run (400 evaluations of scipy's bounded TRF + LSMR) has", "the same active set" if same_set else "NOT reached the active set of the optimum (it is far from converged: see its cost)")
    assert same_set or tag != "config1"
    out = problem_arrays(q)
    out.update(use=use, lo=lo, hi=hi, x=xp, cost=np.array(cost), active_mask=active, kkt_residual=np.array(r), fd_grad_free_inf=np.array(np.abs(g_fd[active == 0]).max()),
               ref_run_x=res.x, ref_run_cost=np.array(res.cost), ref_run_active_mask=res.active_mask, ref_run_nfev=np.array(res.nfev), ref_run_status=np.array(res.status))
    np.savez_compressed(os.path.join(HERE, f"tight_bounds_{tag}.npz"), **out)
    print(f"tight_bounds_{tag}.npz written")


if __name__ == "__main__":
    for t in (sys.argv[1:] or ["config1", "missing3"]):
        main(t)
