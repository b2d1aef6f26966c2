"""cProfile of the user-level bundle_adjust() call (development aid: where the HOST time of a call goes).
    python3 scripts/e2e_cprofile.py [C F rows cols]"""
import contextlib
import cProfile
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import multicam_calibration_amd as m

C, F, rows, cols = (int(a) for a in (sys.argv[1:5] or (2, 50, 6, 9)))
p = m.synth.make_problem(C, F, rows=rows, cols=cols, seed=0, perturb_seed=1)
args = (p["uvs"], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"])


def call():
    with contextlib.redirect_stdout(io.StringIO()):
        return m.bundle_adjust(*args, n_frames=F, return_jac=False)


for _ in range(5):
    call()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    call()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue())
