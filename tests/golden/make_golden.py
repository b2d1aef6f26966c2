"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Run in the build container only (the reference lives at /root/reference and never
travels):      python tests/golden/make_golden.py [--slow]

It loads the reference's own `geometry.py` and `bundle_adjustment.py` unmodified
(recipe of SURVEY.md section 8c: an empty stub for `cv2`, which the hot path never calls)
and records inputs + the reference's outputs as small .npz files.  Nothing here is
reference source -- the fixtures are data.

Files written
  residuals.npz     residuals(x0) for four input variants                     (golden 1)
  jacobian.npz      scipy 2-point / 3-point FD Jacobian of the reference residuals (golden 2)
  sparsity.npz      bundle_adjustment_sparsity indices on a NaN-bearing input (golden 3)
  prefilter.npz     use_frames / printed line for several n_frames, seeds     (golden 4)
  default_run.npz   default bundle_adjust() on config 1 (2 x 50 x 54)         (golden 5)
  robust.npz        scipy's soft_l1/huber/cauchy/arctan rho + rescale         (golden 7)
  tight_*.npz       (--slow) tight-optimum runs, SURVEY.md section 7 protocol (golden 6)
  tight_config1_callable.npz  (--callable) the same recipe with a callable `loss` (tests/losses.py) + the reference's default run with it
  tight_edge_*.npz  (--edge) the same recipe on degenerate inputs (blind camera, 3 frames, 9 / 10 cameras with 40 % missing);
                    (--many) 24 and 27 cameras
"""
import contextlib
import importlib.util
import io
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


def load_reference():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    pkg = types.ModuleType("multicam_calibration")
    pkg.__path__ = ["/root/reference/multicam_calibration"]
    sys.modules["multicam_calibration"] = pkg
    mods = {}
    for name in ("geometry", "bundle_adjustment"):
        spec = importlib.util.spec_from_file_location(f"multicam_calibration.{name}", f"/root/reference/multicam_calibration/{name}.py")
        m = importlib.util.module_from_spec(spec)
        sys.modules[spec.name] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods["geometry"], mods["bundle_adjustment"]


def problem_arrays(p):
    """Flatten a synth problem into npz-storable arrays."""
    K = np.stack([k for k, _ in p["intrinsics"]])
    dist = np.stack([d for _, d in p["intrinsics"]])
    return dict(uvs=p["uvs"], obj=p["obj"], extrinsics=p["extrinsics"], K=K, dist=dist, poses=p["poses"])


def main(slow, edge=False, many=False, callable_loss=False):
    from multicam_calibration_amd import synth
    from scipy.optimize._numdiff import approx_derivative, group_columns
    from scipy.optimize._lsq.least_squares import construct_loss_function
    from scipy.optimize._lsq.common import scale_for_robust_loss_function

    geo, ba = load_reference()

    # ---------------------------------------------------------------- golden 1: residuals
    out = {}
    variants = {
        "complete": dict(n_cameras=2, n_frames=8, seed=10),
        "missing": dict(n_cameras=3, n_frames=10, seed=11, missing=0.3, scalar_nans=7),
        "fourcam": dict(n_cameras=4, n_frames=6, seed=12, rows=5, cols=7),
    }
    for name, kw in variants.items():
        p = synth.make_problem(**kw)
        x0 = ba.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
        for k, v in problem_arrays(p).items():
            out[f"{name}_{k}"] = v
        out[f"{name}_x0"] = x0
        out[f"{name}_res"] = ba.residuals(x0, p["uvs"], p["obj"])
        out[f"{name}_pred"] = ba.predict_calib_uvs(p["extrinsics"], p["intrinsics"], p["obj"], p["poses"])
    # large r^2: push the board towards the image edge (big a, b) with stronger distortion
    p = synth.make_problem(2, 6, seed=13)
    p["poses"][:, 3] += 250.0
    p["intrinsics"] = [(K, d * np.array([3.0, 3.0, 1, 1, 1])) for K, d in p["intrinsics"]]
    x0 = ba.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    for k, v in problem_arrays(p).items():
        out[f"edge_{k}"] = v
    out["edge_x0"] = x0
    out["edge_res"] = ba.residuals(x0, p["uvs"], p["obj"])
    # rodrigues at and near theta = 0, and embed
    rv = np.array([[0.0, 0, 0], [1e-9, -2e-9, 3e-9], [1e-5, 2e-5, -1e-5], [0.3, -0.2, 0.1], [2.0, 1.5, -1.0], [0, 0, np.pi - 1e-3]])
    out["rodrigues_in"] = rv
    out["rodrigues_out"] = geo.rodrigues(rv)
    np.savez_compressed(os.path.join(HERE, "residuals.npz"), **out)
    print("residuals.npz", {k: v.shape for k, v in out.items() if k.endswith("_res")})

    # ---------------------------------------------------------------- golden 2: FD Jacobians of the reference residual
    p = synth.make_problem(3, 6, seed=20, missing=0.2, scalar_nans=3)
    x0 = ba.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    A = ba.bundle_adjustment_sparsity(p["uvs"])
    groups = group_columns(A)
    fun = lambda x: ba.residuals(x, p["uvs"], p["obj"])
    J2 = approx_derivative(fun, x0, method="2-point", sparsity=(A, groups)).tocsr()
    J3 = approx_derivative(fun, x0, method="3-point", sparsity=(A, groups)).tocsr()
    J2.sort_indices(), J3.sort_indices()
    out = problem_arrays(p)
    out.update(x0=x0, J2_data=J2.data, J2_indices=J2.indices, J2_indptr=J2.indptr, J3_data=J3.data, J3_indices=J3.indices, J3_indptr=J3.indptr, n_groups=np.array(groups.max() + 1))
    np.savez_compressed(os.path.join(HERE, "jacobian.npz"), **out)
    print("jacobian.npz", J3.shape, J3.nnz, "groups", groups.max() + 1)

    # ---------------------------------------------------------------- golden 3: sparsity
    Acsr = A.tocsr()
    Acsr.sort_indices()
    np.savez_compressed(os.path.join(HERE, "sparsity.npz"), uvs=p["uvs"], indices=Acsr.indices, indptr=Acsr.indptr, shape=np.array(Acsr.shape))

    # ---------------------------------------------------------------- golden 4: pre-filter (bundle_adjust up to the optimiser)
    p = synth.make_problem(3, 40, seed=30, missing=0.25, outlier_frames=4, scalar_nans=5)
    out = problem_arrays(p)
    cases = []
    for i, (n_frames, seed, thr) in enumerate([(None, 0, None), (1000, 1, None), (10, 2, None), (12, 3, 2.5), (None, 4, 1.0)]):
        np.random.seed(seed)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            # max_nfev=1: we only want the wrapper behaviour (frame choice, printed line, x0)
            r = ba.bundle_adjust(p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=n_frames, outlier_threshold=thr, max_nfev=1, verbose=0)
        use = r[3]
        if n_frames is not None and n_frames == 10 and i == 2:
            pass
        out[f"case{i}_use"] = use
        out[f"case{i}_line"] = np.array(buf.getvalue().splitlines()[0])
        out[f"case{i}_rng_after"] = np.array(np.random.randint(0, 2**31 - 1))
        out[f"case{i}_args"] = np.array([-1 if n_frames is None else n_frames, seed, -1.0 if thr is None else thr], dtype=float)
        cases.append(i)
    # n_frames == len(use_frames): the reference permutes the frames
    np.random.seed(5)
    n_eq = len(out["case0_use"])
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        r = ba.bundle_adjust(p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=n_eq, max_nfev=1, verbose=0)
    out["case5_use"] = r[3]
    out["case5_line"] = np.array(buf.getvalue().splitlines()[0])
    out["case5_rng_after"] = np.array(np.random.randint(0, 2**31 - 1))
    out["case5_args"] = np.array([n_eq, 5, -1.0])
    out["n_cases"] = np.array(6)
    np.savez_compressed(os.path.join(HERE, "prefilter.npz"), **out)
    print("prefilter.npz", [len(out[f"case{i}_use"]) for i in range(6)])

    # ---------------------------------------------------------------- golden 5: default path on config 1
    p = synth.make_problem(2, 50, seed=0, perturb_seed=1)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ext, intr, poses, use, res = ba.bundle_adjust(p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=None)
    out = problem_arrays(p)
    out.update(x=res.x, cost=np.array(res.cost), nfev=np.array(res.nfev), njev=np.array(res.njev), status=np.array(res.status), optimality=np.array(res.optimality), fun=res.fun, grad=res.grad, use=use, log=np.array(buf.getvalue()))
    np.savez_compressed(os.path.join(HERE, "default_run.npz"), **out)
    print("default_run.npz cost", res.cost, "nfev", res.nfev, "njev", res.njev, "status", res.status)

    # ---------------------------------------------------------------- golden 7: robust loss (third-party scipy arithmetic)
    f = np.concatenate([[0.0, 1e-8, 0.5, 1.0, 1.0 + 1e-12, 3.0, 50.0, 1e4], np.random.default_rng(70).normal(0, 2, 40)])
    J = np.random.default_rng(71).normal(size=(f.size, 3))
    out = dict(f=f, J=J)
    for loss in ("soft_l1", "huber", "cauchy", "arctan"):
        for fs in (1.0, 2.5):
            lf = construct_loss_function(f.size, loss, fs)
            rho = lf(f).copy()  # the closure reuses its buffer
            Js, fsc = scale_for_robust_loss_function(J.copy(), f.copy(), rho)
            out[f"{loss}_{fs}_rho"] = rho
            out[f"{loss}_{fs}_cost"] = np.array(lf(f, cost_only=True))
            out[f"{loss}_{fs}_J"] = Js
            out[f"{loss}_{fs}_f"] = fsc
    np.savez_compressed(os.path.join(HERE, "robust.npz"), **out)
    print("robust.npz")

    if not (slow or edge or many or callable_loss):
        return

    # ---------------------------------------------------------------- golden 6: tight optimum (hard part 1)
    # Recipe B of the survey: the REFERENCE's residual function minimised by the same third-party
    # scipy.optimize.least_squares with the reference's settings (trf, soft_l1, x_scale='jac') but the dense
    # exact trust-region solver and tolerances at machine level, from the reference's own x0
    # (serialize_params) on the reference's own frame selection.  (bundle_adjust itself cannot take
    # tr_solver='exact' because it always passes jac_sparsity.)  An analytic dense Jacobian only shortens the
    # path; the optimum is then CERTIFIED independently of it: the gradient of the reference cost by scipy's
    # 3-point finite differences must vanish, and two different starts must agree.
    from oracle import ba_oracle as orc
    from scipy.optimize import least_squares

    def tight(p, tag, pseeds, modify=None, **extra):
        outs = {}
        loss, f_scale = extra.get("loss", "soft_l1"), extra.get("f_scale", 1.0)
        for s, pseed in enumerate(pseeds):
            q = synth.make_problem(perturb_seed=pseed, **p)
            if modify is not None:
                modify(q)
            with contextlib.redirect_stdout(io.StringIO()):
                r0 = ba.bundle_adjust(q["uvs"], q["extrinsics"], q["intrinsics"], q["obj"], q["poses"], n_frames=None, max_nfev=1, verbose=0)
            use = r0[3]
            uvs = q["uvs"][:, use]
            x0 = ba.serialize_params(q["extrinsics"], q["intrinsics"], q["poses"][use])
            jac = lambda x, u, o: orc.jacobian_csr(x, u, o).toarray()
            res = least_squares(ba.residuals, x0, jac=jac, method="trf", tr_solver="exact", x_scale="jac", ftol=1e-15, xtol=1e-15, gtol=1e-11,
                                max_nfev=300, verbose=0, args=(uvs, q["obj"]), loss=loss, f_scale=f_scale)
            # scipy's TRF stalls at |grad| ~ 1e-4..1e-7 here (6-dim gauge null space + running-max x_scale), so the
            # point is polished with dense damped Gauss-Newton steps (min-norm in the gauge directions) on the
            # REFERENCE's residual function until the gradient is at round-off level.
            def grad_at(xv):
                fv = ba.residuals(xv, uvs, q["obj"])
                js, fs_ = orc.robust_scales(fv, loss, f_scale)
                Jd = orc.jacobian_csr(xv, uvs, q["obj"]).toarray() * js[:, None]
                return Jd, fs_, Jd.T @ fs_

            xp = res.x.copy()
            Jd, fs_, g = grad_at(xp)
            for it in range(40):
                if np.abs(g).max() < 1e-9:
                    break
                step = np.linalg.lstsq(Jd, -fs_, rcond=1e-10)[0]
                for k in range(8):  # accept on gradient decrease (cost differences are below round-off here)
                    xn = xp + step * 0.5**k
                    Jn, fn, gn = grad_at(xn)
                    if np.abs(gn).max() < np.abs(g).max():
                        xp, Jd, fs_, g = xn, Jn, fn, gn
                        break
                else:
                    break
            cp = orc.robust_cost(ba.residuals(xp, uvs, q["obj"]), loss, f_scale)
            res.x, res.cost = xp, cp
            res.optimality = np.abs(g).max()
            # independent certificate: FD gradient of the reference's robust cost at the solution
            A = ba.bundle_adjustment_sparsity(uvs)
            J3 = approx_derivative(lambda x: ba.residuals(x, uvs, q["obj"]), res.x, method="3-point", sparsity=(A, group_columns(A))).tocsr()
            f = ba.residuals(res.x, uvs, q["obj"])
            _, r1, _ = orc.loss_rho((f / f_scale) ** 2, loss)
            g_fd = J3.T @ (f * r1)
            print(tag, "start", s, "cost %.15g" % res.cost, "nfev", res.nfev, "opt %.2e" % res.optimality, "FD-grad inf %.2e" % np.abs(g_fd).max(), "status", res.status)
            outs[f"s{s}_x"] = res.x
            outs[f"s{s}_cost"] = np.array(res.cost)
            outs[f"s{s}_use"] = use
            outs[f"s{s}_optimality"] = np.array(res.optimality)
            outs[f"s{s}_fd_grad_inf"] = np.array(np.abs(g_fd).max())
            if s == 0:
                outs.update({k: v for k, v in problem_arrays(q).items()})
        np.savez_compressed(os.path.join(HERE, f"tight_{tag}.npz"), **outs)

    if slow:
        tight(dict(n_cameras=2, n_frames=50, seed=0), "config1", (1, 2))
        tight(dict(n_cameras=3, n_frames=30, seed=40, missing=0.25, scalar_nans=6), "missing3", (1, 2))
        tight(dict(n_cameras=2, n_frames=50, seed=0), "config1_cauchy", (1, 2), loss="cauchy", f_scale=0.5)
    if callable_loss:
        # least_squares' CALLABLE loss (the reference forwards `loss` untouched: bundle_adjustment.py:301-313): the same recipe with a function that is
        # not one of the five names (tests/losses.py), + what the reference's own bundle_adjust(..., loss=function) returns with its default tolerances
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from losses import charbonnier_quarter

        tight(dict(n_cameras=2, n_frames=50, seed=0), "config1_callable", (1, 2), loss=charbonnier_quarter, f_scale=0.7)
        q = synth.make_problem(perturb_seed=1, n_cameras=2, n_frames=50, seed=0)
        with contextlib.redirect_stdout(io.StringIO()):
            r = ba.bundle_adjust(q["uvs"], q["extrinsics"], q["intrinsics"], q["obj"], q["poses"], n_frames=None, loss=charbonnier_quarter, f_scale=0.7, verbose=0)
        z = dict(np.load(os.path.join(HERE, "tight_config1_callable.npz")))
        z.update(ref_default_x=r[4].x, ref_default_cost=np.array(r[4].cost), ref_default_nfev=np.array(r[4].nfev), ref_default_status=np.array(r[4].status))
        np.savez_compressed(os.path.join(HERE, "tight_config1_callable.npz"), **z)
        print("reference bundle_adjust(loss=<function>) default run: cost %.12g nfev %d status %d" % (r[4].cost, r[4].nfev, r[4].status))
    if edge:
        # degenerate inputs (SURVEY 8c edge cases) through the same recipe: a camera that never sees the board (its twelve
        # columns of the Jacobian vanish: the minimum-norm polish leaves its parameters where they started), fewer frames than
        # a wavefront, and the two camera counts either side of the LDS-resident reduced solve with 40 % missing detections
        def blind(q):
            q["uvs"][2] = np.nan

        tight(dict(n_cameras=3, n_frames=40, seed=1), "edge_blind_camera", (1, 2), modify=blind)
        tight(dict(n_cameras=2, n_frames=3, seed=2), "edge_three_frames", (1, 2))
        tight(dict(n_cameras=9, n_frames=20, seed=6, missing=0.4), "edge_nine_cameras", (1, 2))
        tight(dict(n_cameras=10, n_frames=20, seed=7, missing=0.4), "edge_ten_cameras", (1, 2))
    if many:
        # rigs whose reduced system no longer fits LDS: 24 cameras (BASELINE configs[4]'s rig: 16-tile k_syrk wavefronts, 512-thread
        # k_solve_cam) and 27 cameras (two frames per k_syrk stage, the regime a bug was found in)
        tight(dict(n_cameras=24, n_frames=12, seed=8, missing=0.2), "edge_24_cameras", (1, 2))
        tight(dict(n_cameras=27, n_frames=10, seed=9, missing=0.2), "edge_27_cameras", (1, 2))


if __name__ == "__main__":
    main("--slow" in sys.argv, "--edge" in sys.argv, "--many" in sys.argv, "--callable" in sys.argv)
