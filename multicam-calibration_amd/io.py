"""`save_calibration` / `load_calibration` -- the reference's calibration file formats (multicam_calibration/io.py:8-245,
SURVEY.md section 8f-3), same signatures, so results of the GPU solver plug into the same downstream tools.

Formats (io.py:12-26): the extrinsics are stored as 3x3 rotation matrix + 3x1 translation (world -> camera), the intrinsics
as 3x3 camera matrix + distortion coefficients (k1, k2, p1, p2, k3).
  json    one file, camera name -> {"R", "T", "camera_matrix", "distortion_coefs"}  (io.py:54-66): written byte-compatibly.
          UPSTREAM MISMATCH, fixed on load: the reference's reader looks up "rotation" / "translation" (io.py:161-164) while
          its writer stores "R" / "T" (io.py:59-60), so upstream cannot read its own JSON files (KeyError).  This reader
          accepts both spellings; the writer keeps upstream's "R" / "T" so files stay identical to what upstream writes.
  jarvis  directory, one OpenCV-FileStorage YAML file per camera, rotation and camera matrix transposed (io.py:68-80, 185-214).
          The reference writes / reads them with cv2.FileStorage; OpenCV is absent here, so the small subset of the format
          that FileStorage emits for dense double matrices (`!!opencv-matrix`, rows / cols / dt / data) is written and
          parsed directly.  Parity with cv2's byte-level output is UNPINNED (no cv2 to produce a file); the layout follows
          OpenCV's documented YAML persistence format and the round trip is tested.
  gimbal  one HDF5 file, group "camera_parameters" with datasets camera_names / dist_coefs / intrinsic / rotation / translation
          (format description: io.py:22-25).  Needs h5py, like the reference -- an ImportError says so when it is missing.  h5py
          is not in this image: the array logic (_gimbal_pack / _gimbal_unpack) is tested directly and the two h5py loops against
          a dict-backed stand-in (tests/test_io_cpu.py); a real HDF5 file has not been written here.
"""
import json
import os
import re

import numpy as np

from .calibration import get_transformation_matrix, rodrigues_inv


def _cv_yaml_matrix(name, M):
    M = np.atleast_2d(np.asarray(M, dtype=np.float64))
    vals = []
    for v in M.ravel():
        s = repr(float(v))
        if s.endswith(".0"):
            s = s[:-1]          # FileStorage writes integral doubles as "1."
        vals.append(s.replace("inf", ".Inf").replace("nan", ".Nan"))
    lines, cur = [], "   data: [ "
    for i, s in enumerate(vals):
        piece = s + (", " if i + 1 < len(vals) else " ]")
        if len(cur) + len(piece) > 76 and cur.strip():
            lines.append(cur.rstrip())
            cur = "       "
        cur += piece
    lines.append(cur)
    return f"{name}: !!opencv-matrix\n   rows: {M.shape[0]}\n   cols: {M.shape[1]}\n   dt: d\n" + "\n".join(lines) + "\n"


def _cv_yaml_read(path):
    text = open(path).read()
    out = {}
    for m in re.finditer(r"^(\w+):\s*!!opencv-matrix\s*\n\s*rows:\s*(\d+)\s*\n\s*cols:\s*(\d+)\s*\n\s*dt:\s*(\w+)\s*\n\s*data:\s*\[(.*?)\]", text, re.S | re.M):
        name, rows, cols, dt, data = m.groups()
        vals = [float(t.replace(".Inf", "inf").replace(".Nan", "nan").replace(".NaN", "nan")) for t in re.split(r"[,\s]+", data.strip()) if t]
        out[name] = np.array(vals, dtype=np.float64).reshape(int(rows), int(cols))
    return out


# ---- gimbal: one HDF5 file, group "camera_parameters" with five datasets (the format description in the reference's
# docstring, io.py:22-25): camera_names (n strings), dist_coefs (n, 5), intrinsic (n, 3, 3), rotation (n, 3, 3), translation (n, 3).
# The array logic lives in two pure functions that are tested without h5py; the h5py calls themselves are the two loops in
# save_calibration / load_calibration (executed in the tests against a dict-backed stand-in when h5py is not installed).
_GIMBAL_GROUP = "camera_parameters"
_GIMBAL_KEYS = ("camera_names", "dist_coefs", "intrinsic", "rotation", "translation")


def _require_h5py():
    try:
        import h5py
    except ImportError as e:  # same dependency as the reference (io.py:3)
        raise ImportError("the gimbal calibration format is an HDF5 file: it needs h5py, as the reference does") from e
    return h5py


def _gimbal_pack(transforms, all_intrinsics, camera_names):
    """4x4 world -> camera transforms (n, 4, 4) + intrinsics -> the five datasets, camera axis first."""
    n = len(camera_names)
    out = {
        "camera_names": list(camera_names),   # h5py stores a list of str as variable-length UTF-8 strings
        "dist_coefs": np.empty((n, 5)),
        "intrinsic": np.empty((n, 3, 3)),
        "rotation": np.ascontiguousarray(transforms[:, :3, :3]),
        "translation": np.ascontiguousarray(transforms[:, :3, 3]),
    }
    for c, (K, dist) in enumerate(all_intrinsics):
        out["intrinsic"][c] = K
        out["dist_coefs"][c] = np.ravel(dist)
    return out


def _gimbal_unpack(stored, camera_names=None):
    """The five datasets as read back ([()] of each) -> (all_extrinsics, all_intrinsics, camera_names), optionally a subset /
    re-ordering by name.  String datasets come back from h5py as bytes objects."""
    names = [n.decode("utf-8") if isinstance(n, bytes) else str(n) for n in np.asarray(stored["camera_names"]).tolist()]
    pick = range(len(names))
    if camera_names is not None:
        assert set(camera_names) <= set(names), "Camera names must be a subset of names in calibration file"
        pick = [names.index(n) for n in camera_names]
    else:
        camera_names = names
    rot, tra = np.asarray(stored["rotation"]), np.asarray(stored["translation"])
    K, dist = np.asarray(stored["intrinsic"]), np.asarray(stored["dist_coefs"])
    all_extrinsics = [np.concatenate([rodrigues_inv(rot[c]), tra[c]]) for c in pick]
    all_intrinsics = [(K[c], dist[c]) for c in pick]
    return all_extrinsics, all_intrinsics, list(camera_names)


def save_calibration(all_extrinsics, all_intrinsics, camera_names, save_path, save_format="json"):
    """Save calibration results (io.py:8-99; parameters as there)."""
    assert len(all_extrinsics) == len(all_intrinsics) == len(camera_names), "Number of camera names must match number of extrinsics and intrinsics"
    transforms = get_transformation_matrix(np.array(all_extrinsics))
    if save_format == "json":
        data = {}
        for i, name in enumerate(camera_names):
            data[name] = {
                "R": transforms[i, :3, :3].tolist(),
                "T": transforms[i, :3, 3:].tolist(),
                "camera_matrix": np.asarray(all_intrinsics[i][0]).tolist(),
                "distortion_coefs": np.asarray(all_intrinsics[i][1]).tolist(),
            }
        if not save_path.endswith(".json"):
            save_path += ".json"
        with open(save_path, "w") as f:
            json.dump(data, f, indent=4)
    elif save_format == "jarvis":
        os.makedirs(save_path, exist_ok=True)
        for i, name in enumerate(camera_names):
            with open(os.path.join(save_path, f"{name}.yaml"), "w") as f:
                f.write("%YAML:1.0\n---\n")
                f.write(_cv_yaml_matrix("intrinsicMatrix", np.asarray(all_intrinsics[i][0]).T))
                f.write(_cv_yaml_matrix("distortionCoefficients", np.asarray(all_intrinsics[i][1]).reshape(1, -1)))
                f.write(_cv_yaml_matrix("R", transforms[i, :3, :3].T))
                f.write(_cv_yaml_matrix("T", transforms[i, :3, 3:]))
    elif save_format == "gimbal":
        h5py = _require_h5py()
        if not save_path.endswith(".h5"):
            save_path += ".h5"
        with h5py.File(save_path, "w") as h5:
            grp = h5.create_group(_GIMBAL_GROUP)
            for key, value in _gimbal_pack(transforms, all_intrinsics, camera_names).items():
                grp.create_dataset(key, data=value)
    else:
        raise ValueError(f"Unknown format {save_format}")


def load_calibration(load_path, load_format="json", camera_names=None):
    """Load calibration results (io.py:102-245): returns (all_extrinsics list of 6-vectors, all_intrinsics list of
    (camera_matrix, dist_coefs), camera_names)."""
    if load_format == "json":
        with open(load_path, "r") as f:
            data = json.load(f)
        if camera_names is None:
            camera_names = sorted(data.keys())
        else:
            assert set(camera_names) == set(data.keys()), "Camera names must match keys in calibration file"
        all_extrinsics, all_intrinsics = [], []
        for name in camera_names:
            d = data[name]
            R = d["rotation"] if "rotation" in d else d["R"]          # upstream reader's key | upstream writer's key
            T = d["translation"] if "translation" in d else d["T"]
            all_extrinsics.append(np.concatenate([rodrigues_inv(np.array(R)), np.array(T, dtype=np.float64).reshape(-1)]))
            all_intrinsics.append((np.array(d["camera_matrix"]), np.array(d["distortion_coefs"])))
        return all_extrinsics, all_intrinsics, camera_names
    elif load_format == "jarvis":
        files = [f for f in sorted(os.listdir(load_path)) if os.path.splitext(f)[1] in [".yaml", ".YAML"]]
        names_to_files = {os.path.splitext(f)[0]: f for f in files}
        if camera_names is None:
            camera_names = sorted(names_to_files.keys())
        else:
            assert set(camera_names) <= set(names_to_files.keys()), "Camera names must be a subset of yaml files in calibration directory"
        all_extrinsics, all_intrinsics = [], []
        for name in camera_names:
            fs = _cv_yaml_read(os.path.join(load_path, names_to_files[name]))
            all_extrinsics.append(np.concatenate([rodrigues_inv(fs["R"].T), fs["T"].reshape(-1)]))
            all_intrinsics.append((fs["intrinsicMatrix"].T, fs["distortionCoefficients"].reshape(-1)))
        return all_extrinsics, all_intrinsics, camera_names
    elif load_format == "gimbal":
        h5py = _require_h5py()
        if not load_path.endswith(".h5"):
            load_path += ".h5"
        with h5py.File(load_path, "r") as h5:
            grp = h5[_GIMBAL_GROUP]
            stored = {key: grp[key][()] for key in _GIMBAL_KEYS}
        return _gimbal_unpack(stored, camera_names)
    else:
        raise ValueError(f"Unknown format {load_format}")
