"""Tight-optimum goldens ABOVE toy size, produced by the UNMODIFIED reference (SURVEY.md section 7 hard part 1,
section 8c-6 / 8c-8).  Build container only:

    python tests/golden/make_golden_tight_large.py 6 1000            # -> tight_6x1000.npz, tight_6x1000_fixed.npz
    python tests/golden/make_golden_tight_large.py 6 10000 --free-only --starts 1   # headline size, time-boxed
    python tests/golden/make_golden_tight_large.py 6 1000 --free-only --missing 0.3 --scalar-nan-frac 0.01 --suffix _missing   # -> tight_6x1000_missing.npz
                                                          # (round 6: SURVEY 8d's correctness variant -- Bernoulli(0.3) per (camera, frame) + 1 % single scalars -- at a BASELINE size)

Recipe (per start = per perturbation seed of the synthetic initial guess):
  1. free intrinsics: the reference's own `bundle_adjust(...)` is called with an ANALYTIC sparse `jac=` callable and
     tight tolerances passed through its `**opt_kwargs` (bundle_adjustment.py:301-313) -- its pre-filter, its x0, its
     residual function, its call of scipy.optimize.least_squares (trf + lsmr, soft_l1, x_scale='jac').
     fixed intrinsics (BASELINE configs[1]; the reference has no such entry point): the thin wrapper
     fun(y) = ba.residuals(scatter(y, frozen intrinsics), uvs, obj) driven by the same least_squares settings.
  2. scipy's TRF/LSMR stalls at |grad| ~ 1e-3..1e-6 (gauge null space, inexact LSMR): the point is polished by
     sparse damped Gauss-Newton steps on the REFERENCE's residual function until the gradient is at round-off level.
  3. certificate, independent of the analytic Jacobian: the gradient of the reference's robust cost by scipy's
     3-point finite differences of the reference's residuals (sparsity from the reference's own
     bundle_adjustment_sparsity); and two different starts must agree after gauge alignment.
Only parameter vectors and scalars are stored; the inputs are regenerated from the seed by synth.make_problem
(checksum stored).  Nothing here is reference source -- the fixtures are data.
"""
import argparse
import contextlib
import io
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from make_golden import load_reference  # noqa: E402
from multicam_calibration_amd import synth  # noqa: E402
from oracle import ba_oracle as orc  # noqa: E402
from scipy.optimize import least_squares  # noqa: E402
from scipy.optimize._numdiff import approx_derivative, group_columns  # noqa: E402

geo, ba = load_reference()


def polish(res_fun, jac_fun, x, tol=2e-9, iters=25, mu=1e-9, log=print):
    """Sparse damped Gauss-Newton on the robust cost 0.5 sum rho(f^2) of `res_fun` (soft_l1, f_scale 1):
    (J~^T J~ + mu diag) step = -J~^T f~, accepted on gradient decrease.  The tiny Marquardt term only fixes the 6 gauge
    directions (J~ is exactly rank deficient there); the stationary point does not depend on it."""
    def lin(xv):
        f = res_fun(xv)
        js, fs = orc.robust_scales(f)
        Jd = sp.diags(js) @ jac_fun(xv)
        return Jd, fs, Jd.T @ fs

    Jd, fs, g = lin(x)
    for it in range(iters):
        gi = np.abs(g).max()
        log(f"    polish {it}: |grad|inf {gi:.3e}")
        if gi < tol:
            break
        H = (Jd.T @ Jd).tocsc()
        d = H.diagonal()
        H = H + sp.diags(mu * np.where(d > 0, d, 1.0))
        step = spla.splu(H.tocsc()).solve(-g)
        for k in range(6):
            xn = x + step * 0.5**k
            Jn, fn, gn = lin(xn)
            if np.abs(gn).max() < gi:
                x, Jd, fs, g = xn, Jn, fn, gn
                break
        else:
            break
    return x, float(np.abs(g).max())


def fd_gradient_inf(x, uvs, obj, cols=None):
    """|gradient|inf of the REFERENCE's robust cost, 3-point finite differences of the reference's residuals."""
    A = ba.bundle_adjustment_sparsity(uvs)
    J3 = approx_derivative(lambda v: ba.residuals(v, uvs, obj), x, method="3-point", sparsity=(A, group_columns(A))).tocsr()
    f = ba.residuals(x, uvs, obj)
    g = J3.T @ (f * orc.loss_rho(f**2, "soft_l1")[1])
    if cols is not None:
        g = g[cols]
    return float(np.abs(g).max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("C", type=int)
    ap.add_argument("F", type=int)
    ap.add_argument("--starts", type=int, default=2)
    ap.add_argument("--free-only", action="store_true")
    ap.add_argument("--fixed-only", action="store_true")
    ap.add_argument("--max-nfev", type=int, default=60)
    ap.add_argument("--no-fd", action="store_true", help="skip the finite-difference certificate (large sizes: hours)")
    ap.add_argument("--only-start", type=int, default=None, help="compute this start only and park it in --part-dir (large sizes: one process per start)")
    ap.add_argument("--merge", action="store_true", help="combine the parked starts of --part-dir into the fixture")
    ap.add_argument("--part-dir", default="/tmp/gold")
    ap.add_argument("--missing", type=float, default=0.0, help="Bernoulli probability that a whole (camera, frame) detection is NaN (SURVEY 8d's correctness variant: 0.3)")
    ap.add_argument("--scalar-nan-frac", type=float, default=0.0, help="this fraction of the single (u or v) scalars is NaN as well")
    ap.add_argument("--suffix", default="", help="appended to the fixture's name (tight_CxF<suffix>.npz)")
    args = ap.parse_args()
    C, F = args.C, args.F
    tag = f"{C}x{F}{args.suffix}"
    gen = dict(missing=args.missing, scalar_nans=int(round(args.scalar_nan_frac * 2 * C * F * 54)))

    for mode in ("free", "fixed"):
        if (mode == "free" and args.fixed_only) or (mode == "fixed" and args.free_only):
            continue
        outs = {}
        part = lambda k: os.path.join(args.part_dir, f"tight_{tag}_{mode}.s{k}.part.npz")
        for s in range(args.starts):
            if args.merge:
                outs.update(np.load(part(s)))
                continue
            if args.only_start is not None and s != args.only_start:
                continue
            pseed = s + 1
            q = synth.make_problem(C, F, seed=0, perturb_seed=pseed, **gen)
            if mode == "fixed" and s > 0:  # same frozen intrinsics as start 0; only the extrinsics / poses start elsewhere
                q["intrinsics"] = synth.make_problem(C, F, seed=0, perturb_seed=1, **gen)["intrinsics"]
            obj = q["obj"]
            t0 = time.perf_counter()
            if mode == "free":
                jac = lambda x, u, o: orc.jacobian_csr(x, u, o)
                # the second start must solve the SAME frames: its pre-filter threshold (5 x the median error of ITS initial guess)
                # is switched off -- the first start, run with the reference's default, excludes nothing on this data either
                thr = None if s == 0 else 1e30
                with contextlib.redirect_stdout(io.StringIO()):
                    ext, intr, poses, use, res = ba.bundle_adjust(q["uvs"], q["extrinsics"], q["intrinsics"], obj, q["poses"], n_frames=None, outlier_threshold=thr, jac=jac,
                                                                  ftol=1e-13, xtol=1e-13, gtol=1e-10, max_nfev=args.max_nfev, verbose=0)
                uvs = q["uvs"][:, use]
                x = res.x
                cols = None
                res_fun = lambda v: ba.residuals(v, uvs, obj)
                jac_fun = lambda v: orc.jacobian_csr(v, uvs, obj)
            else:
                with contextlib.redirect_stdout(io.StringIO()):
                    use = ba.bundle_adjust(q["uvs"], q["extrinsics"], q["intrinsics"], obj, q["poses"], n_frames=None, max_nfev=1, verbose=0)[3]
                uvs = q["uvs"][:, use]
                xfull = ba.serialize_params(q["extrinsics"], q["intrinsics"], q["poses"][use])
                free = np.ones(xfull.size, bool)
                free[: 12 * C] = np.tile(np.r_[np.zeros(6, bool), np.ones(6, bool)], C)
                cols = np.nonzero(free)[0]

                def scatter(y):
                    v = xfull.copy()
                    v[cols] = y
                    return v

                fun = lambda y: ba.residuals(scatter(y), uvs, obj)
                jacy = lambda y: orc.jacobian_csr(scatter(y), uvs, obj)[:, cols]
                res = least_squares(fun, xfull[cols], jac=jacy, method="trf", loss="soft_l1", x_scale="jac", ftol=1e-13, xtol=1e-13, gtol=1e-10,
                                    max_nfev=args.max_nfev, verbose=0)
                x = res.x
                res_fun, jac_fun = fun, jacy
            t_trf = time.perf_counter() - t0
            print(f"{tag} {mode} start {s}: scipy TRF {t_trf:.1f}s nfev {res.nfev} njev {res.njev} status {res.status} cost {res.cost:.12g} opt {res.optimality:.2e}", flush=True)
            x, gopt = polish(res_fun, jac_fun, x)
            xf = x if mode == "free" else scatter(x)
            cost = orc.robust_cost(ba.residuals(xf, uvs, obj))
            fd = float("nan") if args.no_fd else fd_gradient_inf(xf, uvs, obj, cols)
            print(f"{tag} {mode} start {s}: cost {cost:.15g} |grad|inf {gopt:.2e} FD-grad inf {fd:.2e} total {time.perf_counter() - t0:.1f}s", flush=True)
            outs[f"s{s}_x"] = xf
            outs[f"s{s}_cost"] = np.array(cost)
            outs[f"s{s}_use"] = use
            outs[f"s{s}_optimality"] = np.array(gopt)
            outs[f"s{s}_fd_grad_inf"] = np.array(fd)
            outs[f"s{s}_trf"] = np.array([res.nfev, res.njev, res.status, t_trf])
            if s == 0:
                outs["uvs_checksum"] = np.array(np.nansum(q["uvs"]))
                outs["shape"] = np.array([C, F, obj.shape[0]])
                outs["generator"] = np.array([args.missing, gen["scalar_nans"]], dtype=np.float64)   # synth.make_problem(C, F, seed=0, perturb_seed=s + 1, missing=, scalar_nans=)
            if args.only_start is not None:
                os.makedirs(args.part_dir, exist_ok=True)
                np.savez_compressed(part(s), **{k: v for k, v in outs.items() if k.startswith(f"s{s}_") or k in ("uvs_checksum", "shape", "generator")})
                print("parked", part(s), flush=True)
        if args.only_start is not None:
            continue
        if args.starts > 1:
            assert np.array_equal(outs["s0_use"], outs["s1_use"]), "the two starts selected different frames"
            e0, i0, p0 = orc.deserialize_params(outs["s0_x"], C)
            e1, i1, p1 = orc.deserialize_params(outs["s1_x"], C)
            c0, c1 = outs["s0_x"][: 12 * C].reshape(C, 12)[:, :6], outs["s1_x"][: 12 * C].reshape(C, 12)[:, :6]
            e1a, p1a = orc.gauge_align(e1, p1, e0[0])
            # poses are compared as 4x4 transforms (a rotation vector near |r| = pi has two representations), translations
            # relative to the largest one
            Ta, Tb = orc.to_matrix(p1a), orc.to_matrix(p0)
            dpose = max(np.abs(Ta - Tb)[..., :3, :3].max(), (np.abs(Ta - Tb)[..., :3, 3] / np.abs(Tb[..., :3, 3]).max()).max())
            print(f"{tag} {mode}: two-start agreement intrinsics {np.abs(c0 - c1).max() / 1:.2e} abs, rel {(np.abs(c0 - c1) / np.abs(c0)).max():.2e}; "
                  f"extrinsics {np.abs(e1a - e0).max():.2e}; poses (as transforms) {dpose:.2e}", flush=True)
            # the second start only certifies the first: keep its camera block and the agreement, not its poses
            outs["s1_cam"] = outs["s1_x"][: 12 * C].copy()
            outs["agree_ext"] = np.array(np.abs(e1a - e0).max())
            outs["agree_poses"] = np.array(dpose)
            if F > 2000:
                del outs["s1_x"]
        name = f"tight_{tag}.npz" if mode == "free" else f"tight_{tag}_fixed.npz"
        np.savez_compressed(os.path.join(HERE, name), **outs)
        print("wrote", name, flush=True)


if __name__ == "__main__":
    main()
