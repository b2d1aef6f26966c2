"""mcba -- MI355X-native bundle adjustment behind the reference's `bundle_adjust()`.

Only what the hot path needs lives here:
  csrc/      HIP kernels for gfx950 + the C-ABI (`libmcba.so`, declared in include/mcba.h)
  build.py   hipcc driver that compiles csrc/ in-tree
  ops.py     ctypes binding of that ABI (fails loudly if the library is missing)
  solver.py  host-side Levenberg-Marquardt / Schur driver
  api.py     `bundle_adjust` with the reference's exact signature and return tuple
  diagnostics.py  reprojection_errors (numeric core of plot_residuals), undistort_points -- SURVEY.md section 8f-2
  io.py      save_calibration / load_calibration (json, jarvis; gimbal needs h5py) -- SURVEY.md section 8f-3
  synth.py   deterministic synthetic board detections for tests and bench
"""
from . import synth  # noqa: F401
from . import ops, solver  # noqa: F401
from .api import bundle_adjust, bundle_adjustment, serialize_params, deserialize_params  # noqa: F401
from . import calibration  # noqa: F401
from .triangulation import triangulate  # noqa: F401
from .io import save_calibration, load_calibration  # noqa: F401
from .diagnostics import reprojection_errors, undistort_points  # noqa: F401
from .calibration import calibrate, get_intrinsics, estimate_pose, estimate_all_extrinsics, consensus_calib_poses, get_camera_spanning_tree, estimate_pairwise_camera_transform  # noqa: F401

__all__ = ["bundle_adjust", "bundle_adjustment", "serialize_params", "deserialize_params", "ops", "solver", "synth", "calibration", "calibrate", "triangulate", "get_intrinsics",
           "save_calibration", "load_calibration", "reprojection_errors", "undistort_points", "estimate_pose", "estimate_all_extrinsics", "consensus_calib_poses", "get_camera_spanning_tree", "estimate_pairwise_camera_transform"]
