import sys; sys.path.insert(0,'.')
import multicam_calibration_amd as m
p = m.synth.make_problem(6, 10000, seed=0)
x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
prob = m.ops.Problem(p["uvs"], p["obj"]); prob.set_params(0, x0)
prob.profile_enable(True)
for _ in range(3): prob.jacobian_eval(0, robust_scaled=True)
prob.profile_read()
for _ in range(20): prob.jacobian_eval(0, robust_scaled=True)
ms, n = prob.profile_read()["k_jacobian"]
print("k_jacobian %.1f us -> %.2f TB/s" % (1e3*ms/n, 1037.28e6/(ms/n*1e-3)/1e12))
