"""Golden file for the JSON calibration format WRITTEN BY THE REFERENCE ITSELF (multicam_calibration/io.py:54-66).

Run in the build container only:   python tests/golden/make_golden_io.py
Loads the reference's geometry.py and io.py unmodified, with empty stubs for cv2 and h5py (the JSON branch of
`save_calibration` touches neither), calls `save_calibration(..., save_format="json")` on a seeded calibration and stores
    calibration_ref.json   the bytes the reference wrote
    calibration_io.npz     the inputs (extrinsics, camera matrices, distortion coefficients, camera names)
so that tests/test_io_cpu.py can demand byte equality from multicam-calibration_amd/io.py and read the file back.
The fixtures are data; nothing of the reference's source is stored."""
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


def load_reference_io():
    for stub in ("cv2", "h5py"):
        sys.modules.setdefault(stub, types.ModuleType(stub))
    pkg = types.ModuleType("multicam_calibration")
    pkg.__path__ = ["/root/reference/multicam_calibration"]
    sys.modules["multicam_calibration"] = pkg
    mods = {}
    for name in ("geometry", "io"):
        spec = importlib.util.spec_from_file_location(f"multicam_calibration.{name}", f"/root/reference/multicam_calibration/{name}.py")
        m = importlib.util.module_from_spec(spec)
        sys.modules[spec.name] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods["io"]


def calibration():
    """Four cameras incl. the zero-vector root camera (theta = 0 branch of rodrigues), non-zero p1 p2 k3, awkward doubles."""
    from multicam_calibration_amd import synth

    p = synth.make_problem(4, 3, seed=17)
    ext = np.array(p["extrinsics"], dtype=np.float64)
    ext[0] = 0.0
    Ks = np.stack([K for K, _ in p["intrinsics"]])
    dist = np.stack([np.array([d[0], d[1], 1e-4 * (c + 1), -2e-4, 3e-3 / 7.0]) for c, (_, d) in enumerate(p["intrinsics"])])
    names = ["top", "side_b", "side_a", "bottom"]
    return ext, Ks, dist, names


def main():
    ref_io = load_reference_io()
    ext, Ks, dist, names = calibration()
    intr = [(Ks[c], dist[c]) for c in range(len(names))]
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "calib")                         # the reference appends ".json" (io.py:62-63)
        ref_io.save_calibration(ext, intr, names, path, save_format="json")
        raw = open(path + ".json", "rb").read()
    with open(os.path.join(HERE, "calibration_ref.json"), "wb") as f:
        f.write(raw)
    np.savez(os.path.join(HERE, "calibration_io.npz"), extrinsics=ext, camera_matrices=Ks, dist_coefs=dist, camera_names=np.array(names))
    print("calibration_ref.json: %d bytes written by the reference's save_calibration" % len(raw))


if __name__ == "__main__":
    main()
