"""Host side of the GPU solver: Levenberg-Marquardt on the Schur-reduced camera system.

Replaces the optimisation loop the reference delegates to scipy (`trf_no_bounds`,
scipy/optimize/_lsq/trf.py:401-560, + LSMR): same objective (robust cost 0.5*sum rho(f^2); curvature: the IRLS weight rho' far from the
optimum, scipy's Triggs rescaling -- common.py:720-731 -- at it: CURV_SWITCH below), same termination tests and status codes
(common.py:705-717), but the step comes from an exact damped Gauss-Newton solve:

    linearise (GPU)  ->  Schur complement of the 6x6 frame blocks (GPU)  ->  [all-reduce over frame shards]
    ->  (12C)^2 reduced camera system: Cholesky on the GPU (k_solve_cam; default) or HERE with LAPACK (reduced_solver="host")
    ->  back-substitution + trial cost (GPU)

`problem` is an `ops.Problem` (libmcba.so).  Everything per-observation happens on the GPU.  In the default,
device-resident mode this file only enqueues iterations and reads 32 doubles of LM state per iteration; with the host
solve it sees the (12C)^2 + 3*12C + 16 doubles of the reduced system and 8 trial scalars per step.
"""
import os

import numpy as np
from scipy.linalg import lapack
from scipy.optimize import OptimizeResult

EPS = np.finfo(float).eps
NEUTRAL_BAND = 32 * EPS   # csrc/mcba_math.h MCBA_NEUTRAL_BAND: a gain inside this fraction of the cost = a neutral step
GREY_LEVEL = 1e-9         # csrc/mcba_math.h MCBA_GREY_LEVEL: a rejection whose cost rose by less than this fraction doubles the damping without escalating, and keeps the curvature model

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
    (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).  `problem.reduce_tensor` is the
    torch tensor that aliases the library's reduce buffer (ops.Problem.enable_collective)."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def all_reduce_system(self, problem):
        self.dist.all_reduce(problem.reduce_tensor[: problem.nsys], group=self.group)

    def all_reduce_trial(self, problem):
        self.dist.all_reduce(problem.reduce_tensor[problem.nsys : problem.nsys + 8], group=self.group)

    def all_reduce_tick(self, problem):
        """[system | trial scalars] in one collective (speculative ticks of the device-resident loop)."""
        self.dist.all_reduce(problem.reduce_tensor[: problem.nsys + 8], group=self.group)


class DirectRCCL:
    """Same two collectives as TorchDistributed, but enqueued by libmcba itself with ncclAllReduce on its own reduce
    buffer (ops.Problem.comm_init_from_torch): no torch tensor, no Python collective dispatch (~25 us each)."""

    def __init__(self, problem, group=None):
        self.rank, self.world = problem.comm_init_from_torch(group)

    def all_reduce_system(self, problem):
        problem.comm_allreduce(0, problem.nsys)

    def all_reduce_trial(self, problem):
        problem.comm_allreduce(problem.nsys, 8)


class HostStagedGloo(TorchDistributed):
    """The same collectives staged through the host over a `gloo` group: for REHEARSALS of the frame-sharded loop with
    several ranks sharing one GPU (RCCL refuses two ranks on one device) -- tests and `bench.py --gpus N` on a one-GPU box.
    `.cpu()` synchronises the stream the library launches on (torch's current stream: bundle_adjust / bench pass it in)."""

    def _ar(self, t):
        c = t.cpu()
        self.dist.all_reduce(c, group=self.group)
        t.copy_(c)

    def all_reduce_system(self, problem):
        self._ar(problem.reduce_tensor[: problem.nsys])

    def all_reduce_trial(self, problem):
        self._ar(problem.reduce_tensor[problem.nsys : problem.nsys + 8])

    def all_reduce_tick(self, problem):
        self._ar(problem.reduce_tensor[: problem.nsys + 8])


class InProcessShards:
    """REHEARSAL of north_star's partition inside ONE process: `world` frame shards, one ops.Problem and one LevenbergMarquardt per
    shard, each driven by its own Python thread (the C-ABI calls release the GIL); `comm(rank)` is the shard's collective: the
    slices of the reduce buffers meet in host memory behind a barrier and every shard adds them IN RANK ORDER (so all get the same
    bits, as after an all-reduce).  For boxes that show one GPU and admit at most six processes on it: eight ranks as eight
    processes (tests/test_gpu_multirank.py runs up to six) are refused there.  Same kernels, same tick sequence as a frame-sharded
    run with torch.distributed issuing the collective; no fabric, no RCCL -- the arithmetic and the control flow only."""

    def __init__(self, world):
        import threading

        self.world = int(world)
        self.barrier = threading.Barrier(self.world)
        self.parts = [None] * self.world

    def comm(self, rank, problem, device):
        problem.enable_collective(device)
        return _ShardComm(self, rank)

    def run(self, fn):
        """fn(rank) on one thread per shard; returns the results in rank order.  A shard that raises breaks the barrier, so that
        the others do not wait for it, and its exception is re-raised here."""
        import threading

        out, err = [None] * self.world, []

        def body(rank):
            try:
                out[rank] = fn(rank)
            except BaseException as e:  # noqa: BLE001
                err.append(e)
                self.barrier.abort()

        threads = [threading.Thread(target=body, args=(r,)) for r in range(self.world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if err:
            first = [e for e in err if not isinstance(e, threading.BrokenBarrierError)] or err
            raise first[0]
        return out


class _ShardComm(TorchDistributed):
    def __init__(self, owner, rank):  # (no process group: TorchDistributed's constructor is not called)
        self.owner, self.rank, self.world = owner, int(rank), owner.world

    def _ar(self, t):
        import torch

        o = self.owner
        o.parts[self.rank] = t.cpu().numpy().copy()   # (.cpu() synchronises the stream the library launches on: the null stream)
        o.barrier.wait()
        total = o.parts[0].copy()
        for r in range(1, self.world):
            total += o.parts[r]
        o.barrier.wait()                               # nobody overwrites its part before everybody has read it
        t.copy_(torch.from_numpy(total))

    def all_reduce_system(self, problem):
        self._ar(problem.reduce_tensor[: problem.nsys])

    def all_reduce_trial(self, problem):
        self._ar(problem.reduce_tensor[problem.nsys : problem.nsys + 8])

    def all_reduce_tick(self, problem):
        self._ar(problem.reduce_tensor[: problem.nsys + 8])


# Damping schedule of the Levenberg-Marquardt loop (Nielsen's rule: lambda *= max(DEC_FLOOR, 1 - (2 ratio - 1)^3) on an accepted step,
# lambda *= nu, nu *= 2 on a rejected one).  Nielsen's floor of 1/3 and a start at 1e-4 cost the bench problem (6 x 10 000 x 54 from a
# perturbed start) 24 evaluations to the reference's ftol = 1e-4, six of them rejected first steps and twelve just to bring the damping
# down again; 1/10 and 1e-2 reach a lower cost in 16 (profiles/round3/NOTES_round3.md section 5).  Both are keyword arguments.
# LAM_MIN: the bundle-adjustment Hessian has a 6-DoF gauge null space whose computed curvature is round-off; the damping is what keeps
# the step finite there.  With the faster decrease a floor of 1e-12 is reached before scipy's xtol test can fire and the loop then
# wanders at |step| ~ 1e-6 on neutral steps (4 of 10 problem / tolerance pairs ran into max_nfev); at 1e-9 all of them terminate,
# in 19-26 evaluations instead of 29-33 with the round-2 schedule (1e-4, 1/3, 1e-12).  The minimiser does not depend on it.
# Round 4: those rejected first steps were the curvature model's (CURV_SWITCH below); with the IRLS weight the start can be lower again --
# 1e-3: 64 random problems 1 087 evaluations (1e-2: 1 186, 1e-4: 1 135), the bench problem 5 to the default tolerance (6 / 4).
LAM0 = 1e-3
DEC_FLOOR = 0.1
LAM_MIN = 1e-9

# Curvature model of the linearisations (csrc/mcba_math.h: lm_weight = max(Triggs weight, floor * rho')).  The loop starts on the
# IRLS weight rho' (floor 1: the majorising quadratic -- monotone, the model to be far from the optimum with) and moves to Triggs'
# second-order term (floor 0.1: fast AT the optimum) once an accepted step gained less than CURV_SWITCH of the cost; a rejected step
# whose cost rose by more than GREY_LEVEL (not a round-off rejection in the tail of a converged run) sends it back.  Rounds 1-3 ran on Triggs alone: 16 evaluations to the reference's default tolerance on the bench problem (IRLS /
# this rule: 6), and redescending losses started far from the optimum ran into max_nfev (profiles/round4/NOTES_round4.md section 13).
# The minimiser does not depend on it.  `curvature=` "auto" (this rule), "irls", "triggs"; MCBA_CURVATURE overrides the default.
CURV_IRLS, CURV_TRIGGS = 1.0, 0.1
CURV_SWITCH = 1e-2

MAX_RANKS = 12  # per-rank max |g_f| slots in the reduce buffer's scalar block (include/mcba.h: scal[4..15])


def make_comm(problem, device, group=None, direct=None):
    """Collective backend for a frame-sharded solve: direct RCCL if it can be set up (MCBA_DIRECT_RCCL=0 disables),
    otherwise torch.distributed on a tensor that aliases the reduce buffer."""
    import torch.distributed as dist

    if dist.get_world_size(group) > MAX_RANKS:
        raise ValueError(f"frame-sharded solves support at most {MAX_RANKS} ranks (one max|g_f| slot per rank travels in the SUM all-reduce); got {dist.get_world_size(group)}")
    if dist.get_backend(group) == "gloo" and str(device) != "cpu":  # rehearsal: ranks share a GPU, collectives via the host
        problem.enable_collective(device)
        return HostStagedGloo(group)
    if direct is None:
        direct = os.environ.get("MCBA_DIRECT_RCCL", "1") != "0"
    why = None
    if direct and hasattr(problem, "comm_init_from_torch"):
        try:
            return DirectRCCL(problem, group)
        except Exception as e:  # noqa: BLE001 -- any failure: use the torch path, and say so in result.lm["collectives"]
            why = str(e)
            print(f"[mcba] direct RCCL unavailable ({e}); using torch.distributed collectives", flush=True)
    problem.enable_collective(device)
    comm = TorchDistributed(group)
    comm.fallback_reason = why
    return comm


def _solve_spd(S, rhs):
    """S symmetric positive definite -> step, or None if the Cholesky factorisation fails.
    LAPACK dposv straight on the buffer (S is symmetric, so its C-order memory is a valid Fortran matrix)."""
    if S.size == 0:   # (every camera parameter is held: fixed by the caller, or in the working set of the bounded loop)
        return np.zeros(0)
    _, d, info = lapack.dposv(S.T, rhs, lower=0, overwrite_a=1, overwrite_b=0)
    if info != 0 or not np.isfinite(d).all():
        return None
    return d


class LevenbergMarquardt:
    """State machine: `start(x0)` then `iterate()` until it returns a status.

    One `iterate()` = one LM iteration (trial step, accept / reject, damping update, reduced system at the accepted point,
    its solution).  Three drivers, chosen by what the problem backend offers:
      device-resident (libmcba, default): everything incl. the reduced solve and the termination tests on the GPU; this
          class enqueues ticks `depth` ahead and reads the state each tick posts (`_iterate_auto`);
      host solve (`reduced_solver="host"`): LAPACK Cholesky here, decision on the GPU, one synchronisation per iteration
          (`_iterate_device`);
      host-driven (`iterate` itself): decision here as well -- what the CPU test double (tests/fake_problem.py) runs."""

    def __init__(self, problem, comm=None, free_cam_mask=None, ftol=1e-8, xtol=1e-8, gtol=1e-8, lam0=LAM0, lam_min=LAM_MIN, lam_max=1e12, speculative=True,
                 reduced_solver=None, depth=2, x_scale=None, dec_floor=DEC_FLOOR, curvature=None):
        self.p = problem
        if curvature is None:
            curvature = os.environ.get("MCBA_CURVATURE", "auto")
        if curvature not in ("auto", "irls", "triggs"):
            raise ValueError("curvature must be 'auto', 'irls' or 'triggs'")
        self.curvature = curvature
        self.curv_floor = None
        self.comm = comm or SingleProcess()
        n = problem.n   # size of the camera system: 12 C, or 6 C when the problem holds every camera's intrinsics fixed (ops.Problem.set_camera_block)
        # where the camera system's variables sit in the parameter vector
        self.cam_index = np.asarray(problem.cam_index) if hasattr(problem, "cam_index") else np.arange(n)
        # least_squares' numeric x_scale (this shard's vector, cameras first): the fixed damping matrix D = 1 / x_scale^2 instead
        # of Marquardt's diag(J^T J).  The backend applies it in the frame blocks and the device solve; the host solve needs D_c here.
        self.Dc_fixed = None
        if x_scale is not None:
            problem.set_x_scale(x_scale)
            self.Dc_fixed = 1.0 / np.asarray(x_scale, dtype=np.float64)[self.cam_index] ** 2
        elif hasattr(problem, "set_x_scale"):
            problem.set_x_scale(None)
        self.free = np.ones(n, dtype=bool) if free_cam_mask is None else np.asarray(free_cam_mask, dtype=bool)
        self.all_free = bool(self.free.all())
        self.ftol, self.xtol, self.gtol = ftol, xtol, gtol
        self.lam0, self.lam_min, self.lam_max = float(lam0), lam_min, lam_max
        self.dec_floor = float(dec_floor)
        if hasattr(problem, "lm_set_decrease_floor"):
            problem.lm_set_decrease_floor(self.dec_floor)
        # speculative: every trial point is linearised right away (k_gram also yields its cost), so an accepted
        # step needs one pass over the observations instead of two; a rejected step wastes the extra arithmetic.
        self.speculative = bool(speculative)
        if getattr(problem, "loss_is_callable", False):
            # least_squares' callable `loss`: the function runs here, between a step and the next linearisation -- nothing is linearised before
            # its trial cost is known, and the decision stays on the host (the device-resident loops cannot call back)
            self.speculative = False   # (frame-sharded: every shard evaluates the function on its own residuals, the costs meet in the all-reduce of the trial scalars)
        # device_decide: accept/reject and the damping update run on the GPU (k_decide) so that one LM iteration is a
        # single stream-ordered chain with ONE host synchronisation; needs a backend with lm_iterate (libmcba).
        self.device_decide = self.speculative and hasattr(problem, "lm_iterate")
        # device_solve: the reduced camera system is factorised on the GPU as well (k_solve_cam) and the termination
        # tests run there; the host enqueues `depth` iterations ahead and never synchronises inside the loop.
        # reduced_solver = "host" keeps the LAPACK solve above (one synchronisation per iteration); MCBA_REDUCED_SOLVER
        # overrides the default.
        if reduced_solver is None:
            reduced_solver = os.environ.get("MCBA_REDUCED_SOLVER", "device")
        if reduced_solver not in ("device", "host"):
            raise ValueError("reduced_solver must be 'device' or 'host'")
        self.device_solve = self.device_decide and reduced_solver == "device" and hasattr(problem, "lm_auto_tick")
        self.depth = max(1, min(int(depth), 12))
        self.speculate = os.environ.get("MCBA_SPECULATE", "1") != "0"  # frame-sharded ticks: one collective instead of two
        self.max_nfev = None
        self.max_steps = None

    # ------------------------------------------------------------------ set-up
    def start(self, x0):
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        self.cur = 0
        self._last_state = None
        self.p.set_params(0, x0)
        self.x_cam = x0[self.cam_index].copy()
        self._set_curvature(CURV_TRIGGS if self.curvature == "triggs" else CURV_IRLS)
        self.p.linearize(0)
        self.nfev, self.njev = 1, 1
        self.lam, self.nu = self.lam0, 2.0
        self.iteration, self.steps = 0, 0
        self.rebuilds = 0
        self._terminated = False
        self.cost = None
        self.step_norm = None
        self.actual_reduction = None
        self.g_inf = np.inf
        self.red = None
        self.history = []
        self._refresh_system()
        if not np.isfinite(self.cost):
            raise ValueError("Residuals are not finite in the initial point.")
        if self.device_decide:
            self.p.lm_set_state(self.cost, self.lam, self.nu, self.cur, self.curv_floor, CURV_SWITCH if self.curvature == "auto" else 0.0)
        if self.device_solve:
            p = self.p
            p.synchronize()  # no tick of an earlier run may still be posting into the ring
            p.lm_auto_config(self.ftol, self.xtol, self.gtol, self.lam_min, self.lam_max, None if self.all_free else ~self.free)
            self.issued = self.retired = 1  # sequence number 1 = the solve of the initial system
            p.lm_auto_solve(1)
            st = p.lm_auto_wait(1)
            self.g_inf = float(st[16])
            self._status0 = int(st[15]) or None

    # ------------------------------------------------------------------ the whole loop in one C-ABI crossing
    def can_run_on_device(self):
        """The device-resident loop can be handed to the library as a whole (ops.Problem.lm_run): single GPU, or direct RCCL (the library
        issues the collectives itself).  MCBA_HOST_LOOP=1 keeps the per-tick Python loop (same ticks, same decisions: the A/B tests)."""
        return self.device_solve and hasattr(self.p, "lm_run") and isinstance(self.comm, (SingleProcess, DirectRCCL)) and os.environ.get("MCBA_HOST_LOOP", "0") == "0"

    def run_device(self, x0, verbose=0):
        """start() + the iterate() loop + finalize()'s drain, enqueued and polled by libmcba (mcba_lm_run): one crossing, no host
        synchronisation before the first tick.  x0 None: parameter slot 0 already holds the start point (ops.Problem.subset gathers it on
        the device).  The book-keeping below replays the states the ticks posted through the same _retire as the Python loop."""
        p = self.p
        self.cur = 0
        self.curv_floor = CURV_TRIGGS if self.curvature == "triggs" else CURV_IRLS
        self.nfev, self.njev = 1, 1
        self.lam, self.nu = self.lam0, 2.0
        self.iteration, self.steps, self.rebuilds = 0, 0, 0
        self._terminated = False
        self.step_norm = self.actual_reduction = None
        self.history = []
        status, rows, n_main, steps = p.lm_run(x0, self.ftol, self.xtol, self.gtol, self.lam0, self.lam_min, self.lam_max, self.dec_floor, self.curv_floor,
                                               CURV_SWITCH if self.curvature == "auto" else 0.0, self.max_nfev if self.max_nfev is not None else 2 ** 62, self.max_steps, self.depth,
                                               self.comm.rank % 12, None if self.all_free else ~self.free)
        st0 = rows[0]
        self.cost = self.cost0 = float(st0[0])
        self.g_inf = float(st0[16])
        self._status0 = int(st0[15]) or None
        if verbose == 2:
            _print_header()
            _print_iteration(0, self.nfev, self.cost, None, None, self.g_inf)
        for st in rows[1:n_main]:
            self._retire(st)
            if verbose == 2 and self.accepted:
                _print_iteration(self.iteration, self.nfev, self.cost, self.actual_reduction, self.step_norm, self.g_inf)
        for st in rows[n_main:]:   # retired by the final drain (finalize): their accepted steps count unless the loop had terminated
            if not self._terminated:
                self._retire(st)
        self.steps = steps
        self.issued = self.retired = len(rows)
        self._last_state = rows[-1]
        return status

    def _set_curvature(self, floor):
        """The model of the linearisations ENQUEUED from here on (a kernel argument of k_gram: nothing is recomputed)."""
        if floor != self.curv_floor:
            self.curv_floor = floor
            if hasattr(self.p, "set_curvature_floor"):
                self.p.set_curvature_floor(floor)

    def _curvature_after(self, accepted, dF, cost_before):
        """The switching rule (see CURV_SWITCH above) of the HOST-DRIVEN loop, applied with every decision.  The loops whose decision
        runs on the GPU have it inside lm_decide (csrc/mcba_lm.h: the same rule on the same numbers; the model travels in the LM
        state, slots 25 / 26), so that every driver -- and every rank of a frame-sharded run -- changes model at the same step."""
        if self.curvature != "auto":
            return
        if not accepted:
            if not (-dF <= GREY_LEVEL * cost_before):   # (a rejection at round-off level is no reason to leave Triggs)
                self._set_curvature(CURV_IRLS)
        elif 0.0 <= dF < CURV_SWITCH * cost_before or abs(dF) <= NEUTRAL_BAND * abs(cost_before):
            self._set_curvature(CURV_TRIGGS)

    def _issue_tick(self):
        p, comm = self.p, self.comm
        self.issued += 1
        slot = comm.rank % 12
        if isinstance(comm, TorchDistributed):  # torch issues the collectives, on the stream the library launches on
            p.lm_auto_trial(-1 if self.speculate else 0)  # -1: the speculative reduction sums the trial scalars itself
            if self.speculate:  # ONE collective per tick: speculative Schur reduction, decision inside k_solve_cam
                p.lm_auto_reduce(2, slot)
                comm.all_reduce_tick(p)
                p.lm_auto_solve(self.issued, 1)
            else:
                comm.all_reduce_trial(p)
                p.lm_auto_reduce(1, slot)
                comm.all_reduce_system(p)
                p.lm_auto_solve(self.issued, 0)
        else:  # single GPU, or direct RCCL: the library enqueues the whole tick
            p.lm_auto_tick(self.issued, slot)

    def _iterate_auto(self):
        """One LM iteration of the device-resident loop: top up the ticks in flight, then retire the oldest one."""
        if self._status0 is not None:
            return self._status0
        while self.issued - self.retired < self.depth:
            inflight = self.issued - self.retired
            if self.max_nfev is not None and self.nfev + inflight >= self.max_nfev:
                break
            if self.max_steps is not None and self.issued - 1 >= self.max_steps:
                break
            self._issue_tick()
        if self.issued == self.retired:
            return 0
        self.retired += 1
        return self._retire(self.p.lm_auto_wait(self.retired))

    def _retire(self, st):
        """Book-keeping for one finished tick from the state it posted."""
        done = int(st[15])
        self._terminated = self._terminated or done != 0  # (ticks enqueued behind this one return at once and post the same state)
        if st[24] != 0:  # a rebuild-only tick (the reduced solve had failed, or a speculative reduction was mispredicted)
            self.rebuilds += 1
            self.lam, self.nu = float(st[1]), float(st[2])
            self.g_inf = float(st[16])
            self.accepted = False
            return done or None
        cost_before, lam_used = float(st[21]), float(st[20])
        accepted = st[4] > 0
        cost_new, pred, ratio, step_norm, dF = float(st[5]), float(st[6]), float(st[7]), float(st[8]), float(st[10])
        self.nfev = 1 + int(st[17])
        self.njev = self.nfev
        self.history.append((self.nfev, cost_before, cost_new, pred, ratio, lam_used, step_norm))
        self.cost, self.lam, self.nu, self.cur = float(st[0]), float(st[1]), float(st[2]), int(st[3])
        self.g_inf = float(st[16])
        self.iteration = int(st[18])
        if accepted:
            self.actual_reduction, self.step_norm = dF, step_norm
        self.accepted = bool(accepted)
        self.curv_floor = float(st[25])   # (the model the device decision left for the next linearisations: csrc/mcba_lm.h)
        return done or None

    def _refresh_system(self):
        p = self.p
        if isinstance(self.comm, SingleProcess) and hasattr(p, "reduce_fetch"):
            red = p.reduce_fetch(self.lam, 0)
        else:
            p.build_reduced(self.lam, self.comm.rank % 12)
            self.comm.all_reduce_system(p)
            red = p.get_reduced()
        self.red = red
        self.cost = float(red["scal"][0])
        gc = red["gc"]
        self.g_inf = max(float(np.abs(gc[self.free]).max()) if self.free.any() else 0.0, float(red["scal"][4:16].max()))

    def _adopt(self, red):
        self.red = red
        self.cost = float(red["scal"][0])
        self.g_inf = max(float(np.abs(red["gc"][self.free]).max()) if self.free.any() else 0.0, float(red["scal"][4:16].max()))

    def _solve_cameras(self, red, lam):
        diagU = red["diagU"]
        Dc = self.Dc_fixed if self.Dc_fixed is not None else np.where(diagU > 0, diagU, 1.0)
        S = red["S0"]  # damped in place: the buffer is rebuilt before it is used again
        S.flat[:: self.p.n + 1] += lam * Dc
        if red["scal"][2] != 0:  # a frame block failed to factorise
            return None, Dc
        if self.all_free:
            return _solve_spd(S, red["rhs"]), Dc
        dfree = _solve_spd(np.ascontiguousarray(S[np.ix_(self.free, self.free)]), red["rhs"][self.free])
        if dfree is None:
            return None, Dc
        dc = np.zeros(self.p.n)
        dc[self.free] = dfree
        return dc, Dc

    def _iterate_device(self):
        """One LM iteration with the decision on the GPU: a single host synchronisation (the fetch)."""
        p, red, comm = self.p, self.red, self.comm
        lam = self.lam
        dc, Dc = self._solve_cameras(red, lam)
        distributed = not isinstance(comm, SingleProcess)
        if dc is None:  # more damping, rebuild the reduced system on the GPU
            self.lam = min(lam * self.nu, self.lam_max)
            self.nu *= 2
            p.lm_set_state(self.cost, self.lam, self.nu, self.cur, self.curv_floor, CURV_SWITCH if self.curvature == "auto" else 0.0)
            p.lm_rebuild(comm.rank % 12)
            if distributed:
                comm.all_reduce_system(p)
            new_red, _, _ = p.lm_fetch()
            self._adopt(new_red)
            self.accepted = False
            return 3 if self.lam >= self.lam_max else None
        pred_cam = float(dc @ (lam * Dc * dc - red["gc"]))
        dcn2, xcn2 = float(dc @ dc), float(self.x_cam @ self.x_cam)
        cost_before = self.cost
        if distributed:
            p.lm_trial(dc)
            comm.all_reduce_trial(p)
            p.lm_decide_reduce(pred_cam, dcn2, xcn2, self.lam_min, self.lam_max, comm.rank % 12)
            comm.all_reduce_system(p)
            new_red, t, st = p.lm_fetch()
        else:
            new_red, t, st = p.lm_iterate(dc, pred_cam, dcn2, xcn2, self.lam_min, self.lam_max)
        self.nfev += 1
        self.njev += 1
        accepted = st[4] > 0
        cost_new, pred, ratio, step_norm, x_norm, dF = float(st[5]), float(st[6]), float(st[7]), float(st[8]), float(st[9]), float(st[10])
        self.history.append((self.nfev, cost_before, cost_new, pred, ratio, lam, step_norm))
        ftol_ok = max(dF, 0.0) < self.ftol * cost_before and ratio > 0.25  # (a neutral step has dF ~ -EPS F: counts as 0)
        xtol_ok = step_norm < self.xtol * (self.xtol + x_norm)
        status = 4 if (ftol_ok and xtol_ok) else 2 if ftol_ok else 3 if xtol_ok else None
        if accepted:
            self.cur = int(st[3])
            self.x_cam = self.x_cam + dc
            self.actual_reduction, self.step_norm = dF, step_norm
            self.iteration += 1
        elif status == 2:
            status = None
        self.lam, self.nu = float(st[1]), float(st[2])
        if not accepted and self.lam >= self.lam_max and status is None:
            status = 3
        self._adopt(new_red)
        self.accepted = bool(accepted)
        self.curv_floor = float(st[25])   # (switched by the device decision: csrc/mcba_lm.h)
        return status

    # ------------------------------------------------------------------ one iteration
    def iterate(self, always_linearize=False):
        """Returns None to continue or a scipy-style status (1 gtol, 2 ftol, 3 xtol, 4 both).
        always_linearize: (non-speculative mode) re-linearise even after a rejected step, so that every step
        does identical work; in speculative mode every step linearises its trial point anyway."""
        p, red = self.p, self.red
        self.steps += 1
        if self.device_solve:
            return self._iterate_auto()
        if self.g_inf < self.gtol:
            return 1
        if self.device_decide:
            return self._iterate_device()
        lam = self.lam
        diagU = red["diagU"]
        Dc = self.Dc_fixed if self.Dc_fixed is not None else np.where(diagU > 0, diagU, 1.0)
        S = red["S0"]  # damped in place: the buffer is rebuilt by the next _refresh_system anyway
        S.flat[:: p.n + 1] += lam * Dc
        rhs = red["rhs"]
        dc = None
        if red["scal"][2] == 0:  # every frame block factorised
            if self.all_free:
                dc = _solve_spd(S, rhs)
            else:
                dfree = _solve_spd(np.ascontiguousarray(S[np.ix_(self.free, self.free)]), rhs[self.free])
                if dfree is not None:
                    dc = np.zeros(p.n)
                    dc[self.free] = dfree
        status = None
        accepted = False
        if dc is not None:
            if self.speculative:
                self.njev += 1
            if isinstance(self.comm, SingleProcess) and hasattr(p, "step_fetch"):
                t = p.step_fetch(dc, lam, self.cur, 1 - self.cur, self.speculative)
            else:
                if self.speculative:
                    p.step_linearize(dc, lam, self.cur, 1 - self.cur)
                else:
                    p.step(dc, lam, self.cur, 1 - self.cur)
                self.comm.all_reduce_trial(p)
                t = p.get_trial()
            self.nfev += 1
            cost_new = float(t[0])
            pred = 0.5 * (float(t[1]) + float(dc @ (lam * Dc * dc - red["gc"])))
            step_norm = float(np.sqrt(t[2] + dc @ dc))
            x_norm = float(np.sqrt(t[3] + self.x_cam @ self.x_cam))
            ratio = (self.cost - cost_new) / pred if (np.isfinite(cost_new) and pred > 0) else -1.0
            dF = self.cost - cost_new
            self.history.append((self.nfev, self.cost, cost_new, pred, ratio, lam, step_norm))
            # round-off guard (same rule, same ORDER as lm_decide in csrc/mcba_lm.h): |dF| below FP64 resolution of the cost
            # = neutral step, accepted with ratio := 0.5 BEFORE the termination tests look at the ratio
            neutral = bool(np.isfinite(cost_new) and pred >= 0 and abs(dF) <= NEUTRAL_BAND * abs(self.cost))
            accepted = (ratio > 0 and dF >= 0) or neutral
            if accepted and not (ratio > 0 and dF >= 0):
                ratio = 0.5
            ftol_ok = max(dF, 0.0) < self.ftol * self.cost and ratio > 0.25  # (a neutral step has dF ~ -EPS F: counts as 0)
            xtol_ok = step_norm < self.xtol * (self.xtol + x_norm)
            status = 4 if (ftol_ok and xtol_ok) else 2 if ftol_ok else 3 if xtol_ok else None
            if accepted:
                self.cur = 1 - self.cur
                self.x_cam = self.x_cam + dc
                self.lam = max(lam * max(self.dec_floor, 1.0 - (2.0 * ratio - 1.0) ** 3), self.lam_min)
                self.nu = 2.0
                self.actual_reduction, self.step_norm = dF, step_norm
                self.iteration += 1
            elif status == 2:
                status = None  # ftol needs an accepted step (ratio > 0.25)
        if not accepted:
            grey = dc is not None and bool(cost_new - self.cost <= GREY_LEVEL * self.cost)   # (same rule as lm_decide, csrc/mcba_lm.h)
            self.lam = min(lam * (2.0 if grey else self.nu), self.lam_max)
            self.nu = 2.0 if grey else self.nu * 2
            if self.lam >= self.lam_max and status is None:
                status = 3
        if accepted and self.speculative:
            p.accept_linearization()
        elif accepted or (always_linearize and not self.speculative):
            # (non-speculative mode: the accepted point is linearised HERE, still under the model its step was decided with -- the speculative
            #  host path and the device loops keep the trial linearisation, which was built under that model too: every driver changes model
            #  at the same step and produces the same iterates.  ADVICE r4)
            p.linearize(self.cur)
            self.njev += 1
        self._curvature_after(accepted, (self.cost - cost_new) if dc is not None else 0.0, self.cost)  # the model of the linearisations enqueued from here on
        self._refresh_system()
        self.accepted = accepted
        return status

    def finalize(self):
        """Device-resident loop: retire what is still in flight and make sure the reduce buffer holds the system of the
        CURRENT point (api.bundle_adjust reads g_c and the frame gradients from it).  It does not when the loop was stopped
        from outside (max_nfev) right after a tick whose speculative reduction was mispredicted, or whose solve failed."""
        if not self.device_solve:
            return
        st = None
        while self.retired < self.issued:  # (stopped from outside with ticks in flight: their accepted steps count)
            self.retired += 1
            st = self.p.lm_auto_wait(self.retired)
            if not self._terminated:
                self._retire(st)
        if st is None:
            st = self._last_state if getattr(self, "_last_state", None) is not None else self.p.lm_auto_wait(self.retired)
        if st[15] == 0 and (st[14] != 0 or st[23] != 0):  # not terminated on the device, and the next tick would have rebuilt
            self.p.lm_rebuild(self.comm.rank % 12)
            self.comm.all_reduce_system(self.p)
            red = self.p.get_reduced()
            self.g_inf = max(float(np.abs(red["gc"][self.free]).max()) if self.free.any() else 0.0, float(red["scal"][4:16].max()))

    def result(self, status, lazy_grad=False):
        self.finalize()
        grad = None
        if self.device_solve and hasattr(self.p, "lm_result"):
            x, grad = self.p.lm_result(self.cur, lazy_grad)   # solution + gradient packed on the GPU (api.bundle_adjust attaches the gradient; lazy: a DeviceArray)
        else:
            x = self.p.get_params(self.cur)
        res = self._result(status, x)
        if grad is not None:
            res.lm["grad"] = grad
        return res

    def _result(self, status, x):
        return OptimizeResult(
            x=x, cost=self.cost, optimality=self.g_inf, nfev=self.nfev, njev=self.njev, status=status, message=TERMINATION_MESSAGES[status],
            success=status > 0, active_mask=np.zeros_like(x),
            lm=dict(iterations=self.iteration, steps=self.steps, lam=self.lam, slot=self.cur, history=self.history, rebuilds=getattr(self, "rebuilds", 0), curvature=self.curvature, curvature_floor=self.curv_floor,
                    # which collective backend the frame-sharded loop ran on (DirectRCCL, TorchDistributed [+ why direct RCCL was not
                    # used], HostStagedGloo = rehearsal, SingleProcess) and where the reduced system was solved
                    collectives=type(self.comm).__name__, collectives_fallback_reason=getattr(self.comm, "fallback_reason", None), world=self.comm.world,
                    reduced_solver="device" if self.device_solve else "host",
                    fuse_timeout_tick=(self.p.fuse_status()[0] if self.device_solve and hasattr(self.p, "fuse_status") else 0.0)),
        )


def find_active_constraints(x, lb, ub, rtol=1e-10):
    """scipy's rule for `OptimizeResult.active_mask` (scipy/optimize/_lsq/common.py: find_active_constraints; trf_bounds calls it with
    rtol = xtol): -1 a lower bound is active, +1 an upper bound, 0 neither."""
    active = np.zeros(x.shape, dtype=int)
    if rtol == 0:
        active[x <= lb] = -1
        active[x >= ub] = 1
        return active
    lower_dist, upper_dist = x - lb, ub - x
    lower_threshold, upper_threshold = rtol * np.maximum(1, np.abs(lb)), rtol * np.maximum(1, np.abs(ub))
    active[np.isfinite(lb) & (lower_dist <= np.minimum(upper_dist, lower_threshold))] = -1
    active[np.isfinite(ub) & (upper_dist <= np.minimum(lower_dist, upper_threshold))] = 1
    return active


class BoundedLevenbergMarquardt(LevenbergMarquardt):
    """Box constraints lo <= x <= hi (the reference forwards `bounds` to scipy's least_squares, whose TRF is a bounded solver:
    bundle_adjustment.py:301-313, scipy trf.py: trf_bounds).  Here: an ACTIVE-SET Levenberg-Marquardt on the same Schur-reduced system.
      * working set = the coordinates that sit on a bound with the gradient pushing them outward (x_i = lo_i and g_i > 0, or x_i = hi_i and
        g_i < 0): they leave the linear solve -- camera parameters as rows / columns the reduced solve leaves out, frame coordinates frozen
        inside the Schur reduction and the back-substitution (ops.Problem.set_frozen: their step is exactly 0) -- and are released as soon
        as their gradient points inward;
      * the trial point of every step is PROJECTED onto the box on the GPU (ops.Problem.set_bounds) before its cost is taken, so every
        iterate is feasible; accept / reject, damping and curvature rules are those of the unbounded host-driven loop;
      * optimality = the largest gradient entry outside the working set (the KKT residual), the termination tests scipy's.
    It minimises the same cost over the same box, so it ends at the same constrained minimiser as scipy's trust-region-reflective
    iteration (the iterates differ, as in the unbounded case).  The decision runs on the host (one synchronisation per iteration): this is
    an off-default path, built for exactness, not for the bench.  Frame-sharded runs: lo / hi are this shard's (cameras first, then its own
    frames); every quantity a decision looks at is all-reduced (trial scalars, camera system, the per-rank gradient maxima), so every shard
    takes the same decisions; the working set of a shard's frame coordinates is its own business."""

    def __init__(self, problem, lo, hi, **kw):
        kw["reduced_solver"] = "host"
        super().__init__(problem, **kw)
        self.device_decide = self.device_solve = False
        self.lo, self.hi = np.ascontiguousarray(lo, dtype=np.float64), np.ascontiguousarray(hi, dtype=np.float64)
        self.user_free = self.free.copy()
        self.frozen = np.zeros(problem.nx, dtype=bool)
        self.forced = np.zeros(problem.nx, dtype=bool)   # on a bound, gradient inward, but the coupled LM step bent outward: held until the next accepted step
        self.releases = 0
        self.ncam = 12 * problem.C

    def start(self, x0):
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        if np.any(x0 < self.lo) or np.any(x0 > self.hi):
            raise ValueError("Initial guess is outside of provided bounds")
        self.p.set_bounds(self.lo, self.hi)
        self.p.set_frozen(None)
        self.x_host = x0.copy()
        super().start(x0)
        self._update_working_set()

    def _update_working_set(self):
        """After the reduced system of the current point has been built: who is in the working set now?  (Only coordinates that sit ON a
        bound can be: when none does, nothing is fetched.)"""
        x, ncam = self.x_host, self.ncam
        on_lo, on_hi = x <= self.lo, x >= self.hi
        frozen = np.zeros(x.size, dtype=bool)
        if on_lo.any() or on_hi.any():
            gcam = np.zeros(ncam)
            gcam[self.cam_index] = self.red["gc"]
            g = np.concatenate([gcam, self.p.frame_gradient().ravel()])
            natural = (on_lo & (g > 0)) | (on_hi & (g < 0))
            self.forced &= (on_lo | on_hi) & ~natural   # (a held coordinate that left its bound, or whose gradient now holds it there by itself, is no longer held for a bend)
            frozen = natural | self.forced
        else:
            self.forced[:] = False
        frames_changed = bool((frozen[ncam:] != self.frozen[ncam:]).any())
        self.frozen = frozen
        self.free = self.user_free & ~frozen[:ncam][self.cam_index]   # (camera coordinates: x and g_c are the same on every shard, so is this)
        self.all_free = bool(self.free.all())
        # the frame blocks of the reduced system change with the set: build it again (the linearisation stays).  Frame-sharded: whether a
        # frame coordinate changed sides is known to its own shard only, and the rebuild is a collective -- every shard rebuilds after every
        # iteration (one more all-reduce of the camera system per iteration: an off-default path)
        if frames_changed or self.comm.world > 1:
            if frames_changed:
                self.p.set_frozen(frozen if frozen[ncam:].any() else None)
            self._refresh_system()
        else:
            self.g_inf = max(float(np.abs(self.red["gc"][self.free]).max()) if self.free.any() else 0.0, float(self.red["scal"][4:16].max()))

    def iterate(self, always_linearize=False):
        lam_b, nu_b, tried = self.lam, self.nu, len(self.history)
        status = super().iterate(always_linearize)
        if getattr(self, "accepted", False):
            self.x_host = self.p.get_params(self.cur)   # (the projected point: what the GPU holds)
            self.x_cam = self.x_host[self.cam_index].copy()
        elif len(self.history) > tried and self.comm.world == 1:
            # A rejected step.  Was it BENT?  A coordinate that sits on a bound with the gradient pointing inward is free, but the coupled LM
            # step may still push it outward (the step is -(H + lam D)^-1 g, not -g): the projection then cuts that one component out of a
            # step whose other components were computed with it -- for strongly coupled coordinates (a near-rigid motion of the rig) not a
            # descent step at any length, and raising the damping does not help.  Such coordinates join the working set (the others' steps
            # are then computed with them held), and the damping stays where it was: the step failed for its bend, not for its length.  They
            # STAY in it until the problem with them held has converged (below) -- released after every accepted step, the same
            # coordinates bent again at the next one: three evaluations per accepted step of 2e-6 gain, found by a soak of the
            # randomised sweep.  (Frame-sharded runs keep the plain projection: the bend of a shard's frames is known to it alone.)
            xt = self.p.get_params(1 - self.cur)
            on_lo, on_hi = self.x_host <= self.lo, self.x_host >= self.hi
            bent = ~self.frozen & ((on_lo & (xt <= self.lo)) | (on_hi & (xt >= self.hi)))
            bent[: self.ncam][self.cam_index[~self.user_free]] = False
            if bent.any():
                self.forced |= bent
                self.lam, self.nu = lam_b, nu_b
                if status == 3:
                    status = None
        self._update_working_set()
        if status is None and self.g_inf < self.gtol:
            status = 1
        if status is not None and status > 0 and self.forced.any() and self.releases < 50:
            # converged with coordinates held for a bend: release them and look again (at a minimiser of the held problem the step of a released
            # coordinate points inward: the diagonal of an SPD inverse is positive)
            self.releases += 1
            self.forced[:] = False
            self._update_working_set()
            status = 1 if self.g_inf < self.gtol else None
        return status

    def result(self, status, lazy_grad=False):
        res = super().result(status, False)
        res.active_mask = find_active_constraints(res.x, self.lo, self.hi, rtol=self.xtol)
        res.lm["working_set"] = int(self.frozen.sum())
        return res


def lm_solve(problem, x0, ftol=1e-8, xtol=1e-8, gtol=1e-8, max_nfev=None, verbose=0, comm=None, free_cam_mask=None, lam0=LAM0, max_iterations=None, speculative=True,
             reduced_solver=None, x_scale=None, dec_floor=DEC_FLOOR, curvature=None, x0_on_device=False, lazy_grad=False, bounds=None):
    """Minimise the robust reprojection cost from x0 (this shard's flat vector, a7 layout of SURVEY.md).
    `fun` / `jac` / `grad` of the OptimizeResult are attached by api.bundle_adjust.
    x0_on_device: parameter slot 0 of the problem already holds x0 (ops.Problem.subset gathered it on the GPU): nothing is uploaded."""
    if bounds is not None:   # (lo, hi) in the layout of x: the active-set loop
        lm = BoundedLevenbergMarquardt(problem, bounds[0], bounds[1], comm=comm, free_cam_mask=free_cam_mask, ftol=ftol, xtol=xtol, gtol=gtol, lam0=lam0, speculative=speculative, x_scale=x_scale,
                                       dec_floor=dec_floor, curvature=curvature)
    else:
        lm = LevenbergMarquardt(problem, comm, free_cam_mask, ftol, xtol, gtol, lam0, speculative=speculative, reduced_solver=reduced_solver, x_scale=x_scale, dec_floor=dec_floor, curvature=curvature)
    if max_nfev is None:
        max_nfev = 100 * (np.size(x0) if x0 is not None else problem.nx)  # trf.py:437-438
    lm.max_nfev, lm.max_steps = max_nfev, max_iterations
    if x0 is None and not (bounds is None and lm.can_run_on_device()):
        x0 = problem.get_params(0)   # (the caller left the start point on the device, and this driver wants it on the host)
    if bounds is None and lm.can_run_on_device():   # the loop as a whole inside libmcba (the verbose table is printed from the states the ticks posted)
        status = lm.run_device(None if (x0_on_device or x0 is None) else x0, verbose)
        cost0 = lm.cost0
    else:
        lm.start(x0)
        cost0 = lm.cost
        if verbose == 2:
            _print_header()
            _print_iteration(0, lm.nfev, lm.cost, None, None, lm.g_inf)
        status = None
        while status is None:
            if lm.nfev >= max_nfev or (max_iterations is not None and lm.steps >= max_iterations):
                status = 0
                break
            status = lm.iterate()
            if verbose == 2 and getattr(lm, "accepted", False):
                _print_iteration(lm.iteration, lm.nfev, lm.cost, lm.actual_reduction, lm.step_norm, lm.g_inf)
    res = lm.result(status, lazy_grad)
    if verbose >= 1:
        print(TERMINATION_MESSAGES[status])
        print("Function evaluations {}, initial cost {:.4e}, final cost {:.4e}, first-order optimality {:.2e}.".format(res.nfev, cost0, res.cost, res.optimality))
    return res
