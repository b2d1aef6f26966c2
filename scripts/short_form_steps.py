"""Per-step times of the driver's short bench form (--steps 20 --warmup 5): where do its 1-3 % against the default form go?"""
import sys, time
sys.path.insert(0, ".")
import torch
import multicam_calibration_amd as m
p = m.synth.make_problem(6, 10000, seed=0)
x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
prob = m.ops.Problem(p["uvs"], p["obj"])
for trial in range(3):
    lm = m.solver.LevenbergMarquardt(prob, ftol=0.0, xtol=0.0, gtol=0.0, reduced_solver="device", depth=2)
    lm.start(x0)
    for _ in range(300):
        lm.iterate(always_linearize=True)
    torch.cuda.synchronize()
    lm = m.solver.LevenbergMarquardt(prob, ftol=0.0, xtol=0.0, gtol=0.0, reduced_solver="device", depth=2)
    t_s = time.perf_counter()
    lm.start(x0)
    t_start = time.perf_counter() - t_s
    for _ in range(5):
        lm.iterate(always_linearize=True)
    torch.cuda.synchronize()
    ts = [time.perf_counter()]
    acc = []
    for _ in range(20):
        lm.iterate(always_linearize=True)
        ts.append(time.perf_counter())
        acc.append(int(lm.accepted))
    torch.cuda.synchronize()
    t_end = time.perf_counter()
    d = [(ts[i + 1] - ts[i]) * 1e6 for i in range(20)]
    print("trial %d: lm.start %.2f ms | total %.1f us/step | per-step host intervals (us): %s | accepted: %s | final sync %.1f us" % (
        trial, t_start * 1e3, (t_end - ts[0]) / 20 * 1e6, " ".join("%.0f" % v for v in d), "".join(str(a) for a in acc), (t_end - ts[-1]) * 1e6))
