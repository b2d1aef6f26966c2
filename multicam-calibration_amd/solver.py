"""Host side of the GPU solver: Levenberg-Marquardt on the Schur-reduced camera system.

Replaces the optimisation loop the reference delegates to scipy (`trf_no_bounds`,
scipy/optimize/_lsq/trf.py:401-560, + LSMR): same objective (robust cost 0.5*sum rho(f^2), Triggs
rescaling as scipy/optimize/_lsq/common.py:720-731), same termination tests and status codes
(common.py:705-717), but the step comes from an exact damped Gauss-Newton solve:

    linearise (GPU)  ->  Schur complement of the 6x6 frame blocks (GPU)  ->  [all-reduce over frame shards]
    ->  (12C)^2 reduced camera system solved HERE with LAPACK Cholesky  ->  back-substitution + trial cost (GPU)

`problem` is an `ops.Problem` (libmcba.so).  Everything per-observation happens on the GPU; this file
only sees the (12C)^2 + 3*12C + 16 doubles of the reduced system and 8 trial scalars per step.
"""
import numpy as np
import scipy.linalg as sla
from scipy.optimize import OptimizeResult

TERMINATION_MESSAGES = {
    -1: "Improper input parameters status returned from `leastsq`",
    0: "The maximum number of function evaluations is exceeded.",
    1: "`gtol` termination condition is satisfied.",
    2: "`ftol` termination condition is satisfied.",
    3: "`xtol` termination condition is satisfied.",
    4: "Both `ftol` and `xtol` termination conditions are satisfied.",
}


def _print_header():
    print("{:^15}{:^15}{:^15}{:^15}{:^15}{:^15}".format("Iteration", "Total nfev", "Cost", "Cost reduction", "Step norm", "Optimality"))


def _print_iteration(iteration, nfev, cost, cost_reduction, step_norm, optimality):
    cr = "{:^15}".format("") if cost_reduction is None else "{:^15.2e}".format(cost_reduction)
    sn = "{:^15}".format("") if step_norm is None else "{:^15.2e}".format(step_norm)
    print("{:^15}{:^15}{:^15.4e}{}{}{:^15.2e}".format(iteration, nfev, cost, cr, sn, optimality))


class SingleProcess:
    """No-op collective for one GPU."""

    rank = 0
    world = 1

    def all_reduce_system(self, problem):
        pass

    def all_reduce_trial(self, problem):
        pass


class TorchDistributed:
    """Frame shards on several GPUs: ONE all-reduce (SUM, f64) of the reduced camera system per
    linear solve and one of the 8 trial scalars per trial step, via torch.distributed
    (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests)."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def all_reduce_system(self, problem):
        t = problem.reduce_tensor
        self.dist.all_reduce(t[: problem.nsys], group=self.group)

    def all_reduce_trial(self, problem):
        t = problem.reduce_tensor
        self.dist.all_reduce(t[problem.nsys : problem.nsys + 8], group=self.group)


def _solve_reduced(S, rhs):
    """(S symmetric positive definite) -> step, or None if the factorisation fails."""
    try:
        c, low = sla.cho_factor(S, lower=True, check_finite=False)
    except (sla.LinAlgError, ValueError):
        return None
    d = sla.cho_solve((c, low), rhs, check_finite=False)
    return d if np.all(np.isfinite(d)) else None


def lm_solve(problem, x0, ftol=1e-8, xtol=1e-8, gtol=1e-8, max_nfev=None, verbose=0, comm=None, free_cam_mask=None,
             lam0=1e-4, lam_min=1e-12, lam_max=1e12, max_iterations=None, callback=None):
    """Minimise the robust reprojection cost starting from x0 (this shard's flat vector, a7 layout).

    Returns an OptimizeResult with x, cost, grad-related scalars, nfev, njev, status, message, success and
    `lm` diagnostics.  `fun` / `jac` / `grad` are attached by the caller (api.bundle_adjust) on request.
    `max_iterations` stops after that many ACCEPTED-or-REJECTED LM iterations regardless of tolerances
    (used by bench.py to time a fixed number of steps)."""
    comm = comm or SingleProcess()
    n = problem.n
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    n_total = problem.nx if comm.world == 1 else None
    if max_nfev is None:
        max_nfev = 100 * (x0.size if n_total is None else n_total)
    free = np.ones(n, dtype=bool) if free_cam_mask is None else np.asarray(free_cam_mask, dtype=bool)
    all_free = bool(free.all())

    cur = 0
    problem.set_params(cur, x0)
    x_cam = x0[:n].copy()
    problem.linearize(cur)
    nfev, njev = 1, 1
    lam, nu = float(lam0), 2.0
    iteration, n_steps = 0, 0
    status = None
    cost = None
    step_norm = None
    actual_reduction = None
    g_inf = np.inf
    need_system = True
    history = []

    while True:
        if need_system:
            problem.build_reduced(lam, comm.rank % 12)
            comm.all_reduce_system(problem)
            red = problem.get_reduced()
            scal = red["scal"]
            cost = float(scal[0])
            if not np.isfinite(cost):
                if nfev == 1:
                    raise ValueError("Residuals are not finite in the initial point.")
                raise FloatingPointError("non-finite cost at an accepted point")
            gc = red["gc"]
            g_inf = max(float(np.abs(gc[free]).max()) if free.any() else 0.0, float(scal[4:16].max()))
            if verbose == 2:
                if iteration == 0 and n_steps == 0:
                    _print_header()
                if actual_reduction is None or accepted_last:
                    _print_iteration(iteration, nfev, cost, actual_reduction, step_norm, g_inf)
            if g_inf < gtol:
                status = 1
                break
        accepted_last = False
        if max_iterations is not None and n_steps >= max_iterations:
            status = 0
            break
        if nfev >= max_nfev:
            status = 0
            break

        Dc = np.where(red["diagU"] > 0, red["diagU"], 1.0)
        S = red["S0"] + np.diag(lam * Dc)
        rhs = red["rhs"]
        if all_free:
            dc = _solve_reduced(S, rhs)
        else:
            dfree = _solve_reduced(S[np.ix_(free, free)], rhs[free])
            dc = None
            if dfree is not None:
                dc = np.zeros(n)
                dc[free] = dfree
        n_steps += 1
        if dc is None or scal[2] > 0:  # reduced system or a frame block not positive definite: more damping
            lam = min(lam * nu, lam_max)
            nu *= 2
            need_system = True
            if lam >= lam_max:
                status = 3
                break
            continue

        problem.step(dc, lam, cur, 1 - cur)
        comm.all_reduce_trial(problem)
        t = problem.get_trial()
        nfev += 1
        cost_new = float(t[0])
        pred = 0.5 * (float(t[1]) + float(dc @ (lam * Dc * dc - gc)))
        step_norm = float(np.sqrt(t[2] + dc @ dc))
        x_norm = float(np.sqrt(t[3] + x_cam @ x_cam))
        if np.isfinite(cost_new) and pred > 0:
            ratio = (cost - cost_new) / pred
        else:
            ratio = -1.0
        history.append((nfev, cost, cost_new, pred, ratio, lam, step_norm))

        dF = cost - cost_new
        ftol_ok = dF < ftol * cost and ratio > 0.25
        xtol_ok = step_norm < xtol * (xtol + x_norm)
        term = 4 if (ftol_ok and xtol_ok) else 2 if ftol_ok else 3 if xtol_ok else None

        if ratio > 0 and dF >= 0:
            cur = 1 - cur
            x_cam = x_cam + dc
            lam = max(lam * max(1.0 / 3.0, 1.0 - (2.0 * ratio - 1.0) ** 3), lam_min)
            nu = 2.0
            actual_reduction = dF
            iteration += 1
            accepted_last = True
            problem.linearize(cur)
            njev += 1
            need_system = True
            if callback is not None:
                callback(iteration, cost_new)
            if term is not None:
                status = term
                # one more system build so cost/optimality describe the returned point
                problem.build_reduced(lam, comm.rank % 12)
                comm.all_reduce_system(problem)
                red = problem.get_reduced()
                cost = float(red["scal"][0])
                g_inf = max(float(np.abs(red["gc"][free]).max()) if free.any() else 0.0, float(red["scal"][4:16].max()))
                if verbose == 2:
                    _print_iteration(iteration, nfev, cost, actual_reduction, step_norm, g_inf)
                break
        else:
            lam = min(lam * nu, lam_max)
            nu *= 2
            need_system = True
            if term == 3 or term == 4:
                status = 3
                break
            if lam >= lam_max:
                status = 3
                break

    x = problem.get_params(cur)
    res = OptimizeResult(
        x=x, cost=cost, optimality=g_inf, nfev=nfev, njev=njev, status=status, message=TERMINATION_MESSAGES[status],
        success=status > 0, active_mask=np.zeros_like(x), lm=dict(iterations=iteration, steps=n_steps, lam=lam, slot=cur, history=history),
    )
    if verbose >= 1:
        print(TERMINATION_MESSAGES[status])
        print("Function evaluations {}, initial cost {:.4e}, final cost {:.4e}, first-order optimality {:.2e}.".format(
            nfev, history[0][1] if history else cost, cost, g_inf))
    return res
