"""Two ranks on ONE MI355X (both on cuda:0, gloo rendezvous): the frame-sharded device-resident LM loop end to end
through libmcba.so -- trial kernels -> all-reduce of the trial scalars -> decision kernel -> Schur reduction ->
all-reduce of the reduced system -> k_solve_cam -- against the single-process run on the same data.
RCCL refuses two ranks on one device, so the collectives here are host-staged gloo all-reduces of the very same
buffers (solver.HostStagedGloo, chosen by solver.make_comm for a gloo group on a CUDA device); the RCCL plumbing itself is exercised by the single-rank `nccl` runs of bench.py (MCBA_BENCH_FORCE_DIST=1)."""
import datetime
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(420)]   # (a rank that hangs must fail the test, not the run: pytest-timeout here, faulthandler in the full-size workers)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem(m):
    return m.synth.make_problem(4, 150, seed=31, missing=0.1, scalar_nans=5)   # 150 frames: ragged 64-frame blocks per shard


def _curvature(tag):
    """The speculative one-collective ticks are tested on their hard case -- mispredicted ticks -- and those need rejected steps: with the
    default curvature model (IRLS first) this problem's steps are all accepted, on Triggs alone (rounds 1-3) the first ones are rejected.
    The other modes run the default."""
    return "triggs" if tag in ("device", "device_nofuse") else "auto"


def _worker(rank, world, port, out_dir, mode):
    sys.path.insert(0, ROOT)
    if mode == "device2":  # two collectives per tick instead of the speculative single one
        os.environ["MCBA_SPECULATE"] = "0"
        mode = "device"
        tag = "device2"
    elif mode == "device_nofuse":  # the solve and the next trial step's back-substitution as two launches
        os.environ["MCBA_FUSE_BACKSUB"] = "0"
        mode = "device"
        tag = "device_nofuse"
    else:
        tag = mode
    import contextlib
    import io

    import torch
    import torch.distributed as dist

    import multicam_calibration_amd as m

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))   # a lost rank fails the collectives instead of blocking them for 30 minutes

    # gloo group + CUDA device: solver.make_comm picks HostStagedGloo (host-staged all-reduces of the library's reduce buffer)
    p = _problem(m)
    with contextlib.redirect_stdout(io.StringIO()):
        e, it, ps, use, res = m.bundle_adjust(p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=None, device=0,
                                              ftol=0.0, xtol=1e-12, gtol=1e-10, max_nfev=int(os.environ.get("MCBA_TEST_MAX_NFEV", "100")), verbose=0, distributed=True,
                                              return_jac=False, reduced_solver=mode, curvature=_curvature(tag))
    np.savez(os.path.join(out_dir, f"{tag}{rank}.npz"), x=res.x, cost=res.cost, nfev=res.nfev, status=res.status, use=use, grad=res.grad,
             rebuilds=res.lm["rebuilds"], steps=res.lm["steps"])
    dist.destroy_process_group()


def test_four_ranks_one_gpu_speculative_ticks(tmp_path):
    """Four frame shards (ragged: 150 frames -> 38 / 38 / 37 / 37), four per-rank gradient slots, one collective per tick."""
    import contextlib
    import io

    import torch.multiprocessing as mp

    mp.spawn(_worker, args=(4, _free_port(), str(tmp_path), "device"), nprocs=4, join=True)
    rs = [np.load(tmp_path / f"device{r}.npz") for r in range(4)]
    for r in rs[1:]:
        np.testing.assert_array_equal(r["x"], rs[0]["x"])
        assert float(r["cost"]) == float(rs[0]["cost"]) and int(r["nfev"]) == int(rs[0]["nfev"])
    import multicam_calibration_amd as m

    p = _problem(m)
    with contextlib.redirect_stdout(io.StringIO()):
        e, it, ps, use, res = m.bundle_adjust(p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=None,
                                              ftol=0.0, xtol=1e-12, gtol=1e-10, max_nfev=100, verbose=0, return_jac=False)
    assert abs(res.cost - float(rs[0]["cost"])) <= 1e-10 * res.cost
    cam_a, cam_b = rs[0]["x"][:48].reshape(4, 12), res.x[:48].reshape(4, 12)
    assert (np.abs(cam_a[:, :6] - cam_b[:, :6]) / np.abs(cam_b[:, :6])).max() < 1e-6


@pytest.mark.parametrize("mode", ["device", "device2", "host"])
def test_two_ranks_one_gpu_match_single_process(tmp_path, mode):
    import contextlib
    import io

    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), mode), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / f"{mode}0.npz"), np.load(tmp_path / f"{mode}1.npz")
    if mode == "device":  # speculative ticks: mispredictions cost rebuild ticks (here: the rejected first steps), never correctness
        assert int(r0["rebuilds"]) >= 1 and int(r0["steps"]) == int(r0["nfev"]) - 1 + int(r0["rebuilds"])
    else:
        assert int(r0["rebuilds"]) == 0
    mode = "device" if mode == "device2" else mode
    # every rank took the same decisions and returns the same full result
    assert int(r0["nfev"]) == int(r1["nfev"]) and int(r0["status"]) == int(r1["status"]) and int(r0["status"]) > 0
    assert float(r0["cost"]) == float(r1["cost"])
    np.testing.assert_array_equal(r0["x"], r1["x"])
    np.testing.assert_array_equal(r0["use"], r1["use"])

    import multicam_calibration_amd as m

    p = _problem(m)
    with contextlib.redirect_stdout(io.StringIO()):
        e, it, ps, use, res = m.bundle_adjust(p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=None,
                                              ftol=0.0, xtol=1e-12, gtol=1e-10, max_nfev=100, verbose=0, return_jac=False, reduced_solver=mode, curvature=_curvature(mode))
    np.testing.assert_array_equal(use, r0["use"])
    assert abs(res.cost - float(r0["cost"])) <= 1e-10 * res.cost
    C = 4
    cam_a, cam_b = r0["x"][: 12 * C].reshape(C, 12), res.x[: 12 * C].reshape(C, 12)
    assert (np.abs(cam_a[:, :6] - cam_b[:, :6]) / np.abs(cam_b[:, :6])).max() < 1e-6   # intrinsics + distortion: gauge-free
    assert np.abs(r0["grad"]).max() <= 10 * max(res.optimality, 1e-6)   # the sharded run is as stationary as the single-process one


def test_two_ranks_fused_backsub_is_bit_identical(tmp_path):
    """Speculative frame-sharded ticks with the next trial step's back-substitution inside the solve's launch (its loads
    addressed on the reduction's prediction, its result dropped when the solve finds the prediction wrong) against the same
    ticks with a k_backsub launch of their own: the same iterates to the last bit, mispredicted ticks included."""
    import torch.multiprocessing as mp

    for mode in ("device", "device_nofuse"):
        mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), mode), nprocs=2, join=True)
    a, b = np.load(tmp_path / "device0.npz"), np.load(tmp_path / "device_nofuse0.npz")
    assert int(a["rebuilds"]) >= 1 and int(a["rebuilds"]) == int(b["rebuilds"])
    assert int(a["nfev"]) == int(b["nfev"]) and int(a["steps"]) == int(b["steps"]) and float(a["cost"]) == float(b["cost"])
    np.testing.assert_array_equal(a["x"], b["x"])
    np.testing.assert_array_equal(a["grad"], b["grad"])


# ------------------------------------------------------------------ the direct RCCL path (one rank: RCCL needs one device per rank)
def _rccl_worker(rank, world, port, out_dir, speculate):
    sys.path.insert(0, ROOT)
    os.environ["MCBA_SPECULATE"] = "1" if speculate else "0"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import contextlib
    import io

    import torch
    import torch.distributed as dist

    import multicam_calibration_amd as m

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    p = _problem(m)
    made = []
    real_make_comm = m.solver.make_comm

    def make_comm(problem, device, group=None, direct=None):
        c = real_make_comm(problem, device, group, direct)
        made.append(type(c).__name__)
        return c

    m.solver.make_comm = make_comm
    with contextlib.redirect_stdout(io.StringIO()):
        e, it, ps, use, res = m.bundle_adjust(p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=None, device=0,
                                              ftol=0.0, xtol=1e-12, gtol=1e-10, max_nfev=100, verbose=0, distributed=True, return_jac=False)
    np.savez(os.path.join(out_dir, f"rccl{int(speculate)}.npz"), x=res.x, cost=res.cost, nfev=res.nfev, status=res.status, rebuilds=res.lm["rebuilds"], comm=np.array(made[0]))
    dist.destroy_process_group()


@pytest.mark.parametrize("speculate", [True, False])
def test_single_rank_direct_rccl_matches_single_process(tmp_path, speculate):
    """bundle_adjust(distributed=True) over the `nccl` backend with the library's own RCCL communicator (ncclAllReduce
    enqueued by mcba_lm_auto_tick): one collective per tick (speculative) and two, against the plain single-GPU solve."""
    import contextlib
    import io

    import torch.multiprocessing as mp

    mp.spawn(_rccl_worker, args=(1, _free_port(), str(tmp_path), speculate), nprocs=1, join=True)
    r = np.load(tmp_path / f"rccl{int(speculate)}.npz")
    assert str(r["comm"]) == "DirectRCCL"
    assert speculate or int(r["rebuilds"]) == 0   # (two collectives per tick: nothing is predicted, nothing rebuilt; the speculative tick's mispredictions are test_two_ranks_one_gpu_match_single_process[device]'s subject)

    import multicam_calibration_amd as m

    p = _problem(m)
    with contextlib.redirect_stdout(io.StringIO()):
        e, it, ps, use, res = m.bundle_adjust(p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=None,
                                              ftol=0.0, xtol=1e-12, gtol=1e-10, max_nfev=100, verbose=0, return_jac=False)
    assert int(r["status"]) == res.status and int(r["nfev"]) == res.nfev
    assert abs(float(r["cost"]) - res.cost) <= 1e-12 * res.cost
    np.testing.assert_allclose(r["x"], res.x, rtol=0, atol=1e-9 * np.abs(res.x).max())


def test_stop_by_max_nfev_after_a_mispredicted_tick_leaves_a_consistent_gradient(tmp_path):
    """The first trial steps of this problem are rejected, i.e. mispredicted by the speculative reduction; stopping right
    there (max_nfev) must still return the gradient of the CURRENT point (LevenbergMarquardt.finalize rebuilds the system)."""
    import contextlib
    import io

    import torch.multiprocessing as mp

    os.environ["MCBA_TEST_MAX_NFEV"] = "3"
    try:
        mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), "device"), nprocs=2, join=True)
    finally:
        del os.environ["MCBA_TEST_MAX_NFEV"]
    r0 = np.load(tmp_path / "device0.npz")
    assert int(r0["status"]) == 0 and int(r0["nfev"]) == 3 and int(r0["rebuilds"]) >= 1
    import multicam_calibration_amd as m

    p = _problem(m)
    with contextlib.redirect_stdout(io.StringIO()):
        e, it, ps, use, res = m.bundle_adjust(p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=None,
                                              ftol=0.0, xtol=1e-12, gtol=1e-10, max_nfev=3, verbose=0, return_jac=False, reduced_solver="host", curvature=_curvature("device"))
    assert abs(res.cost - float(r0["cost"])) <= 1e-12 * res.cost
    np.testing.assert_allclose(r0["x"], res.x, rtol=0, atol=1e-9 * np.abs(res.x).max())
    np.testing.assert_allclose(r0["grad"], res.grad, rtol=0, atol=1e-9 * np.abs(res.grad).max())


# ------------------------------------------------------------------ BASELINE configs[3] / configs[4] at (a quarter of) full size, frame-sharded
def _full_worker(rank, world, port, out_dir, shape, n_frames):
    sys.path.insert(0, ROOT)
    import contextlib
    import faulthandler
    import io
    import time

    # a hung rank reports where and ends: the test fails instead of hanging -- BEFORE the collectives' own timeout (180 s) fires in a
    # rank that is only waiting for it, so that the rank that is stuck is the one whose stack is printed
    faulthandler.dump_traceback_later(int(os.environ.get("MCBA_TEST_WATCHDOG_S", "150")), exit=True)

    import torch.distributed as dist

    import multicam_calibration_amd as m

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))   # a lost rank fails the collectives instead of blocking them for 30 minutes
    C, F, rows, cols = shape
    p = m.synth.make_problem(C, F, rows=rows, cols=cols, seed=0)
    np.random.seed(7)   # (rank 0's global RNG draws the subsample; the others' state must not matter)
    buf = io.StringIO()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(buf):
        e, it, ps, use, res = m.bundle_adjust(p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=n_frames, device=0,
                                              ftol=1e-13, xtol=1e-12, gtol=1e-8, max_nfev=60, verbose=0, distributed=True, return_jac=False)
    dt = time.perf_counter() - t0
    hist = np.array([(h[0], h[1], h[2], h[5]) for h in res.lm["history"]])   # nfev, cost before, trial cost, damping: the decisions
    np.savez(os.path.join(out_dir, f"full{rank}.npz"), cam=res.x[: 12 * C], poses=ps, cost=res.cost, nfev=res.nfev, status=res.status, use=use, hist=hist,
             printed=np.array(buf.getvalue()), seconds=dt, n_local=len(res.lm["frame_positions"]), collectives=np.array(res.lm["collectives"]), fun_size=res.fun.size)
    dist.destroy_process_group()


@pytest.mark.parametrize("tag,shape,world,n_frames", [
    ("config4-6x100000x54", (6, 100000, 6, 9), 4, None),          # BASELINE configs[3]: every frame, four shards of 25 000
    ("config4-subsample", (6, 100000, 6, 9), 4, 40000),           # the reference's random subsample (n_frames < usable frames): ragged, scattered shards
    ("config5-24x12500x200", (24, 12500, 10, 20), 4, None),       # a quarter of BASELINE configs[4] (the frames two of its eight GPUs hold), 288 x 288 on-GPU solve
])
def test_full_size_frame_sharded_rehearsal(tmp_path, tag, shape, world, n_frames):
    """SURVEY 7-1's pin for the sharded configs, as far as a one-GPU box allows: the whole problem solved as `world` gloo ranks
    SHARING the GPU (each uploads and pre-filters only its slice of the frames; collectives host-staged) and as one process --
    identical decisions on every rank, the same selection, the cost to 1e-10 and the intrinsics to 1e-6.
    N > 1 RCCL ranks have still never executed (one GPU per box); this is the control flow and the arithmetic, not the fabric."""
    import contextlib
    import io

    import torch.multiprocessing as mp

    import multicam_calibration_amd as m

    mp.spawn(_full_worker, args=(world, _free_port(), str(tmp_path), shape, n_frames), nprocs=world, join=True)
    rs = [np.load(tmp_path / f"full{r}.npz") for r in range(world)]
    for r in rs[1:]:   # every rank: the same decisions (trial costs and dampings to the bit), the same cameras, the same assembled poses
        np.testing.assert_array_equal(r["hist"], rs[0]["hist"])
        np.testing.assert_array_equal(r["cam"], rs[0]["cam"])
        np.testing.assert_array_equal(r["poses"], rs[0]["poses"])
        np.testing.assert_array_equal(r["use"], rs[0]["use"])
        assert float(r["cost"]) == float(rs[0]["cost"]) and int(r["nfev"]) == int(rs[0]["nfev"]) and int(r["status"]) == int(rs[0]["status"])
        assert str(r["printed"]) == ""
    assert str(rs[0]["printed"]).startswith("Excluding ") and str(rs[0]["collectives"]) == "HostStagedGloo"
    assert sum(int(r["n_local"]) for r in rs) == rs[0]["use"].size and all(int(r["n_local"]) > 0 for r in rs)
    C, F, rows, cols = shape
    assert all(int(r["fun_size"]) == 2 * C * rows * cols * int(r["n_local"]) for r in rs)   # result.fun covers the rank's own frames

    p = m.synth.make_problem(C, F, rows=rows, cols=cols, seed=0)
    np.random.seed(7)
    with contextlib.redirect_stdout(io.StringIO()):
        e, it, ps, use, res = m.bundle_adjust(p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=n_frames,
                                              ftol=1e-13, xtol=1e-12, gtol=1e-8, max_nfev=60, verbose=0, return_jac=False)
    np.testing.assert_array_equal(use, rs[0]["use"])   # same frames in the same (selection) order, incl. the random subsample
    assert res.status > 0 and int(rs[0]["status"]) > 0
    assert abs(res.cost - float(rs[0]["cost"])) <= 1e-10 * res.cost
    cam_a, cam_b = rs[0]["cam"].reshape(C, 12), res.x[: 12 * C].reshape(C, 12)
    assert (np.abs(cam_a[:, :6] - cam_b[:, :6]) / np.abs(cam_b[:, :6])).max() < 1e-6   # intrinsics + distortion: gauge-free
    print(f"[{tag}] sharded {max(float(r['seconds']) for r in rs):.2f} s per rank, nfev {int(rs[0]['nfev'])}, cost {float(rs[0]['cost']):.9g}")


# ------------------------------------------------------------------ north_star's partition: EIGHT frame shards, full size (one process, one thread per shard)
@pytest.mark.timeout(900)
@pytest.mark.parametrize("tag,shape", [("config4-6x100000x54-8-shards", (6, 100000, 6, 9)),       # BASELINE configs[3]: 12 500 frames per shard -- the real shard
                                       ("config5-24x50000x200-8-shards", (24, 50000, 10, 20))])    # BASELINE configs[4], the FULL problem: 6 250 frames per shard, 288 x 288 on-GPU solve
def test_north_star_partition_eight_shards(tag, shape):
    """The 8-way frame partition north_star names, at full size: eight shards (ops.Problem + LevenbergMarquardt each, the
    frame-sharded tick with its one collective per iteration, speculative Schur reduction and all) in ONE process -- a GPU box
    admits at most six processes on its card, so eight gloo ranks cannot run there (the six-rank / four-rank process rehearsals
    are above) -- against ONE solve of the whole problem on the same GPU (3.84 GB of observations at configs[4]).  Every shard takes
    bit-identical decisions; cost to 1e-10, intrinsics to 1e-6 of the one-handle solve.  RCCL with N > 1 ranks remains unexecuted."""
    import threading
    import time

    import torch

    import multicam_calibration_amd as m

    C, F, rows, cols = shape
    world = 8
    p = m.synth.make_problem(C, F, rows=rows, cols=cols, seed=0)
    x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    kw = dict(ftol=1e-13, xtol=1e-12, gtol=1e-8, max_nfev=40)
    # ---- one handle, every frame
    prob = m.ops.Problem(p["uvs"], p["obj"])
    t0 = time.perf_counter()
    one = m.solver.lm_solve(prob, x0, **kw)
    t_one = time.perf_counter() - t0
    prob.close()
    # ---- eight shards
    bounds = np.linspace(0, F, world + 1).astype(int)
    assert set(np.diff(bounds)) == {F // world}
    shards = m.solver.InProcessShards(world)
    out, err = [None] * world, []

    def run(rank):
        try:
            lo, hi = bounds[rank], bounds[rank + 1]
            pr = m.ops.Problem(np.ascontiguousarray(p["uvs"][:, lo:hi]), p["obj"])
            comm = shards.comm(rank, pr, torch.device("cuda:0"))
            xs = np.concatenate([x0[: 12 * C], x0[12 * C:].reshape(F, 6)[lo:hi].ravel()])
            out[rank] = m.solver.lm_solve(pr, xs, comm=comm, **kw)
            pr.close()
        except BaseException as e:  # noqa: BLE001 -- a failed shard must not leave the others at the barrier
            err.append(e)
            shards.barrier.abort()

    t0 = time.perf_counter()
    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    t_eight = time.perf_counter() - t0
    assert not err, err
    h0 = np.array([(h[0], h[1], h[2], h[5]) for h in out[0].lm["history"]])
    for r in out[1:]:
        np.testing.assert_array_equal(np.array([(h[0], h[1], h[2], h[5]) for h in r.lm["history"]]), h0)   # trial costs and dampings to the bit
        np.testing.assert_array_equal(r.x[: 12 * C], out[0].x[: 12 * C])
        assert r.cost == out[0].cost and r.nfev == out[0].nfev and r.status == out[0].status
    assert one.status > 0 and out[0].status > 0
    assert abs(out[0].cost - one.cost) <= 1e-10 * one.cost
    ca, cb = out[0].x[: 12 * C].reshape(C, 12), one.x[: 12 * C].reshape(C, 12)
    assert (np.abs(ca[:, :6] - cb[:, :6]) / np.abs(cb[:, :6])).max() < 1e-6
    print(f"[{tag}] one handle: {t_one:.2f} s for nfev {one.nfev} ({1e3 * t_one / one.nfev:.2f} ms per evaluation); eight shards in one process: {t_eight:.2f} s, nfev {out[0].nfev}, cost {out[0].cost:.9g}")


def test_frame_shards_with_the_six_wide_camera_block():
    """BASELINE configs[1] frame-sharded: three shards in one process (solver.InProcessShards), every handle on the 6-wide camera block
    (intrinsics fixed): the collective carries the (6C)^2 + 3 * 6C + 16 + 8 doubles of that system.  Same decisions on every shard; the cost
    and the predictions of the one-handle solve; the intrinsics untouched."""
    import threading

    import torch

    import multicam_calibration_amd as m
    from oracle import ba_oracle as orc

    C, F, world = 6, 3 * 700, 3
    p = m.synth.make_problem(C, F, seed=12, missing=0.1)
    x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    kw = dict(ftol=1e-13, xtol=1e-12, gtol=1e-9, max_nfev=30)
    prob = m.ops.Problem(p["uvs"], p["obj"])
    assert prob.set_camera_block(6)
    one = m.solver.lm_solve(prob, x0, **kw)
    prob.close()
    bounds = np.linspace(0, F, world + 1).astype(int)
    shards = m.solver.InProcessShards(world)
    out, err = [None] * world, []

    def run(rank):
        try:
            lo, hi = bounds[rank], bounds[rank + 1]
            pr = m.ops.Problem(np.ascontiguousarray(p["uvs"][:, lo:hi]), p["obj"])
            assert pr.set_camera_block(6)            # before the collective's buffer is sized
            comm = shards.comm(rank, pr, torch.device("cuda:0"))
            assert pr.reduced_size() == (6 * C) ** 2 + 3 * 6 * C + 16 + 8 + 2 * 32
            xs = np.concatenate([x0[: 12 * C], x0[12 * C:].reshape(F, 6)[lo:hi].ravel()])
            out[rank] = m.solver.lm_solve(pr, xs, comm=comm, **kw)
            pr.close()
        except BaseException as e:  # noqa: BLE001
            err.append(e)
            shards.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not err, err
    h0 = np.array([(h[0], h[1], h[2], h[5]) for h in out[0].lm["history"]])
    for r in out[1:]:
        np.testing.assert_array_equal(np.array([(h[0], h[1], h[2], h[5]) for h in r.lm["history"]]), h0)
        np.testing.assert_array_equal(r.x[: 12 * C], out[0].x[: 12 * C])
    assert one.status > 0 and out[0].status > 0 and abs(out[0].cost - one.cost) <= 1e-10 * one.cost
    np.testing.assert_array_equal(out[0].x[: 12 * C].reshape(C, 12)[:, :6], x0[: 12 * C].reshape(C, 12)[:, :6])
    xa = np.concatenate([out[0].x[: 12 * C]] + [r.x[12 * C:] for r in out])
    assert np.abs(orc.predict_from_x(xa, C, p["obj"]) - orc.predict_from_x(one.x, C, p["obj"])).max() < 1e-5


def test_a_poll_timeout_on_one_shard_is_seen_by_all():
    """The fused back-substitution's bounded poll (mcba_backsub.h) running out on ONE shard only: its trial point is stale, so that
    tick must decide nothing -- on EVERY shard, or the shard that rebuilt alone takes other decisions from then on and, in a run with
    real collectives, ends with another number of them (a hang).  The flag travels in the all-reduced trial scalars (slot 5).
    Two shards in one process; shard 1's handle is created with MCBA_FUSE_MAX_POLLS=0 (its first fused poll gives up at once)."""
    import torch

    import multicam_calibration_amd as m
    from oracle import ba_oracle as orc

    C, F, world = 4, 2 * 320, 2
    p = m.synth.make_problem(C, F, seed=31, missing=0.1)
    x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
    kw = dict(ftol=1e-13, xtol=1e-12, gtol=1e-9, max_nfev=40)
    bounds = np.linspace(0, F, world + 1).astype(int)

    def solve(polls_of_shard_1):
        shards = m.solver.InProcessShards(world)
        probs, comms, xs = [], [], []
        for r in range(world):
            old = os.environ.get("MCBA_FUSE_MAX_POLLS")
            if r == 1 and polls_of_shard_1 is not None:
                os.environ["MCBA_FUSE_MAX_POLLS"] = polls_of_shard_1
            try:   # (the knob is read when the handle allocates its solver buffers: at the first linearisation)
                pr = m.ops.Problem(np.ascontiguousarray(p["uvs"][:, bounds[r]:bounds[r + 1]]), p["obj"])
                comms.append(shards.comm(r, pr, torch.device("cuda:0")))
                xs.append(np.concatenate([x0[: 12 * C], x0[12 * C:].reshape(F, 6)[bounds[r]:bounds[r + 1]].ravel()]))
                pr.set_params(0, xs[-1])
                pr.linearize(0)
                probs.append(pr)
            finally:
                if old is None:
                    os.environ.pop("MCBA_FUSE_MAX_POLLS", None)
                else:
                    os.environ["MCBA_FUSE_MAX_POLLS"] = old

        def run(rank):
            pr = probs[rank]
            try:
                res = m.solver.lm_solve(pr, xs[rank], comm=comms[rank], **kw)
                return res, pr.fuse_status()
            finally:
                pr.close()

        return shards.run(run)

    calm = solve(None)
    hit = solve("0")
    key = lambda r: np.array([(h[0], h[1], h[2], h[5]) for h in r.lm["history"]])
    for out in (calm, hit):
        np.testing.assert_array_equal(key(out[1][0]), key(out[0][0]))                       # both shards: the same decisions, to the bit
        np.testing.assert_array_equal(out[1][0].x[: 12 * C], out[0][0].x[: 12 * C])
        assert out[0][0].status > 0 and out[0][0].status == out[1][0].status and out[0][0].nfev == out[1][0].nfev
    assert calm[0][1][0] == 0 and calm[1][1][0] == 0 and calm[0][0].lm["rebuilds"] == calm[1][0].lm["rebuilds"]
    assert hit[1][1][0] > 0 and not hit[1][1][1]                                            # shard 1: the event is on record, its handle left the fused launch
    assert hit[0][1][0] == 0                                                                # shard 0 never timed out itself ...
    assert hit[0][0].lm["rebuilds"] == hit[1][0].lm["rebuilds"] > calm[0][0].lm["rebuilds"]  # ... and discarded the same tick
    assert abs(hit[0][0].cost - calm[0][0].cost) <= 1e-9 * calm[0][0].cost
    xa = lambda out: np.concatenate([out[0][0].x[: 12 * C]] + [o[0].x[12 * C:] for o in out])
    assert np.abs(orc.predict_from_x(xa(hit), C, p["obj"]) - orc.predict_from_x(xa(calm), C, p["obj"])).max() < 1e-5


# ------------------------------------------------------------------ bounds in a frame-sharded run (round 5)
def _bounds_worker(rank, world, port, out_dir, tag):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import contextlib
    import faulthandler
    import io

    faulthandler.dump_traceback_later(int(os.environ.get("MCBA_TEST_WATCHDOG_S", "150")), exit=True)
    import torch.distributed as dist

    import multicam_calibration_amd as m
    from conftest import GOLDEN, problem_from_npz

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    z = np.load(os.path.join(GOLDEN, f"tight_bounds_{tag}.npz"))
    uvs, ext, intr, obj, poses = problem_from_npz(z)
    with contextlib.redirect_stdout(io.StringIO()):
        e, it, ps, use, res = m.bundle_adjust(uvs, ext, intr, obj, poses, n_frames=None, device=0, bounds=(z["lo"], z["hi"]), ftol=1e-15, xtol=1e-15, gtol=1e-9, max_nfev=400,
                                              verbose=0, distributed=True, return_jac=False)
    np.savez(os.path.join(out_dir, f"b{rank}.npz"), x=res.x, cost=res.cost, active_mask=res.active_mask, use=use, status=res.status, nfev=res.nfev,
             collectives=np.array(res.lm["collectives"]), n_local=len(res.lm["frame_positions"]))
    dist.destroy_process_group()


@pytest.mark.parametrize("tag", ["config1", "missing3"])
def test_bounds_in_a_frame_sharded_run_two_ranks(tmp_path, tag):
    """bundle_adjust(distributed=True, bounds=...) as two gloo ranks sharing the GPU: the bounds are those of the whole parameter vector,
    each rank takes its cameras' and its own frames' part (k_clip, frozen frame coordinates in k_syrk / k_backsub per shard); the same
    decisions on both ranks, the reference's bounded optimum (tests/golden/tight_bounds_*.npz), the assembled active_mask."""
    import torch.multiprocessing as mp

    from conftest import GOLDEN

    mp.spawn(_bounds_worker, args=(2, _free_port(), str(tmp_path), tag), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "b0.npz"), np.load(tmp_path / "b1.npz")
    for k in ("x", "active_mask", "use"):
        np.testing.assert_array_equal(r0[k], r1[k])
    assert float(r0["cost"]) == float(r1["cost"]) and int(r0["nfev"]) == int(r1["nfev"]) and int(r0["status"]) in (1, 2, 3, 4)
    assert str(r0["collectives"]) == "HostStagedGloo" and int(r0["n_local"]) > 0 and int(r1["n_local"]) > 0
    z = np.load(os.path.join(GOLDEN, f"tight_bounds_{tag}.npz"))
    np.testing.assert_array_equal(r0["use"], z["use"])
    assert abs(float(r0["cost"]) - float(z["cost"])) <= 1e-9 * float(z["cost"])
    C = z["uvs"].shape[0] if "uvs" in z.files else int(r0["x"].size - 6 * r0["use"].size) // 12
    intr_idx = np.array([12 * c + k for c in range(C) for k in range(6)])
    np.testing.assert_array_equal(r0["active_mask"][intr_idx], z["active_mask"][intr_idx])
    if tag == "config1":
        np.testing.assert_array_equal(r0["active_mask"], z["active_mask"])
    am = r0["active_mask"]
    assert np.all(r0["x"] >= z["lo"]) and np.all(r0["x"] <= z["hi"])
    np.testing.assert_array_equal(r0["x"][am == 1], z["hi"][am == 1])
    np.testing.assert_array_equal(r0["x"][am == -1], z["lo"][am == -1])


# ------------------------------------------------------------------ a callable loss in a frame-sharded run (round 5)
def _callable_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import contextlib
    import faulthandler
    import io

    faulthandler.dump_traceback_later(int(os.environ.get("MCBA_TEST_WATCHDOG_S", "150")), exit=True)
    import torch.distributed as dist

    import multicam_calibration_amd as m
    from conftest import GOLDEN, problem_from_npz
    from losses import charbonnier_quarter

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    z = np.load(os.path.join(GOLDEN, "tight_config1_callable.npz"))
    uvs, ext, intr, obj, poses = problem_from_npz(z)
    with contextlib.redirect_stdout(io.StringIO()):
        e, it, ps, use, res = m.bundle_adjust(uvs, ext, intr, obj, poses, n_frames=None, device=0, loss=charbonnier_quarter, f_scale=0.7, ftol=1e-14, xtol=1e-14, gtol=1e-9, max_nfev=300,
                                              verbose=0, distributed=True, return_jac=False)
    hist = np.array([(h[0], h[1], h[2], h[5]) for h in res.lm["history"]])
    np.savez(os.path.join(out_dir, f"c{rank}.npz"), x=res.x, cost=res.cost, use=use, status=res.status, nfev=res.nfev, hist=hist, K=np.stack([k for k, _ in it]))
    dist.destroy_process_group()


def test_callable_loss_in_a_frame_sharded_run_two_ranks(tmp_path):
    """bundle_adjust(distributed=True, loss=<function>) as two gloo ranks sharing the GPU: every rank evaluates the function on the residuals of
    its own frames (table + trial cost), the costs meet in the all-reduce of the trial scalars: identical decisions, the golden optimum."""
    import torch.multiprocessing as mp

    from conftest import GOLDEN

    mp.spawn(_callable_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "c0.npz"), np.load(tmp_path / "c1.npz")
    for k in ("x", "use", "hist", "K"):
        np.testing.assert_array_equal(r0[k], r1[k])
    assert float(r0["cost"]) == float(r1["cost"]) and int(r0["status"]) in (1, 2, 3, 4)
    z = np.load(os.path.join(GOLDEN, "tight_config1_callable.npz"))
    np.testing.assert_array_equal(r0["use"], z["s0_use"])
    assert abs(float(r0["cost"]) - float(z["s0_cost"])) <= 1e-9 * float(z["s0_cost"])
    cam, cam_g = r0["x"][:24].reshape(2, 12), z["s0_x"][:24].reshape(2, 12)
    assert (np.abs(cam[:, :6] - cam_g[:, :6]) / np.abs(cam_g[:, :6])).max() < 1e-6
