import contextlib, io, sys, cProfile, pstats
import numpy as np
sys.path.insert(0, ".")
import multicam_calibration_amd as m
p = m.synth.make_problem(6, 10000, seed=0)
def run():
    with contextlib.redirect_stdout(io.StringIO()):
        return m.bundle_adjust(p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"], n_frames=10000, verbose=0, return_jac=False)
run()
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
