"""A fixed launch sequence of calibrate()'s dense kernels for rocprofv3 (kernel stats and counter passes): at 6 x 10 000 x 54 (MCBA_SHAPE=C,F,rows,cols)
ten launches of k_pnp over every (camera, frame) -- cv2.solvePnP's job for 60 000 views each --, then the pose graph (k_pose_pairs + the radix select
of the medians, k_pose_consensus).  Prints the per-view evaluation statistics of the last launch."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, ".")
import multicam_calibration_amd as m
from multicam_calibration_amd import calibration as cal

C, F, rows, cols = (int(v) for v in os.environ.get("MCBA_SHAPE", "6,10000,6,9").split(","))
p = m.synth.make_problem(C, F, rows=rows, cols=cols, seed=0, missing=float(os.environ.get("MCBA_CAL_MISSING", "0.1")))
intr9 = np.c_[p["true_cam"][:, :6] * (1 + 1e-3), np.zeros((C, 3))]
prob = m.ops.Problem(p["uvs"], p["obj"], loss="linear")
for _ in range(10):
    ok, _, ev = prob.calib_poses(intr9, want_evals=True)
tree = cal._spanning_tree(ok, root=0)
for _ in range(5):
    tr, cnt = prob.calib_pairwise(tree)
    ext = cal._chain_extrinsics(C, tree, tr, 0)
    poses = prob.calib_consensus(ext)
N = rows * cols
print(json.dumps({"shape": [C, F, N], "views_with_a_pose": int(ok.sum()), "lm_evaluations_per_view": {"mean": float(ev[ok].mean()), "max": int(ev[ok].max()), "min": int(ev[ok].min())},
                  "algorithmic_bytes_per_k_pnp_launch": {"observations_read_once": 16 * C * F * N, "poses_out": 2 * 48 * C * F, "note": "the kernel re-reads the observations once per pass (3 start passes + one per LM evaluation): from L2 / MALL after the first"}}))
prob.close()
