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
  gimbal  one HDF5 file (io.py:82-95, 216-242): needs h5py, exactly like the reference -- an ImportError says so when it
          is missing (it is not in this image, so that branch is untested here).
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
        import h5py  # same dependency as the reference (io.py:3); not part of this image

        if not save_path.endswith(".h5"):
            save_path += ".h5"
        with h5py.File(save_path, "w") as h5:
            grp = h5.create_group("camera_parameters")
            grp.create_dataset("dist_coefs", data=np.stack([np.asarray(k[1]) for k in all_intrinsics]))
            grp.create_dataset("intrinsic", data=np.stack([np.asarray(k[0]) for k in all_intrinsics]))
            grp.create_dataset("rotation", data=transforms[:, :3, :3])
            grp.create_dataset("translation", data=transforms[:, :3, 3])
            grp.create_dataset("camera_names", data=camera_names)
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
        import h5py  # see save_calibration

        if not load_path.endswith(".h5"):
            load_path += ".h5"
        with h5py.File(load_path, "r") as h5:
            grp = h5["camera_parameters"]
            h5_names = [n.decode("utf-8") for n in grp["camera_names"][()].tolist()]
            all_intrinsics = list(zip(grp["intrinsic"][()], grp["dist_coefs"][()]))
            all_extrinsics = np.concatenate([rodrigues_inv(grp["rotation"][()]), grp["translation"][()]], axis=1)
            if camera_names is None:
                camera_names = h5_names
            else:
                assert set(camera_names) <= set(h5_names), "Camera names must be a subset of names in calibration file"
                ix = np.array([h5_names.index(n) for n in camera_names])
                all_extrinsics = all_extrinsics[ix]
                all_intrinsics = [all_intrinsics[i] for i in ix]
            return list(all_extrinsics), all_intrinsics, camera_names
    else:
        raise ValueError(f"Unknown format {load_format}")
