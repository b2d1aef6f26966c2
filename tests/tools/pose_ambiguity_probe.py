import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, contextlib, io
import multicam_calibration_amd as mc
from oracle import calibration_oracle as co
from test_gpu_calibration import _draw_rig
for it in (464, 1021, 1204):
    mk, root, ns = _draw_rig(it)
    p = mc.synth.make_problem(**mk)
    C, F, N = p["uvs"].shape[:3]
    np.random.seed(it)
    ext, intr, poses, tree = mc.calibrate(p["uvs"], [(1280, 1024)] * C, p["obj"], root=root, verbose=False, n_samples_for_intrinsics=ns)
    intr9 = np.array([np.r_[K[0, 0], K[1, 1], K[0, 2], K[1, 2], d] for K, d in intr])
    prob = mc.ops.Problem(p["uvs"], p["obj"], loss="linear")
    ok, per_cam, evals = prob.calib_poses(intr9, want_poses=True, want_evals=True)
    prob.close()
    # truth in camera coordinates
    Tt = co.get_transformation_matrix(p["true_cam"][:, None, 6:]) @ co.get_transformation_matrix(p["true_poses"][None])
    truth = co.get_transformation_vector(Tt)
    Ra, Rb = co.rodrigues_batch(per_cam[..., :3]), co.rodrigues_batch(truth[..., :3])
    ang = np.arccos(np.clip((np.einsum("cfij,cfij->cf", Ra, Rb) - 1) / 2, -1, 1))
    ang[~ok] = 0
    flipped = np.argwhere(ang > 0.15)
    print(it, "views", int(ok.sum()), "mirrored", len(flipped), "max angle", ang.max())
    for c, f in flipped[:4]:
        uv = p["uvs"][c, f]
        K = np.array([[intr9[c, 0], 0, intr9[c, 2]], [0, intr9[c, 1], intr9[c, 3]], [0, 0, 1.0]])
        start = co.poses_from_homographies(co.homographies(p["obj"][:, :2], co.undistort_normalized(uv[None], K, intr9[c, 4:])), np.eye(3))[0]
        x_ref, c_ref = co.solve_pnp(uv, p["obj"], intr9[c], start)
        x_tr, c_tr = co.solve_pnp(uv, p["obj"], intr9[c], truth[c, f])
        mine = 0.5 * np.sum((uv - co.project5(p["obj"], per_cam[c, f], intr9[c])) ** 2)
        print("   view", c, f, "angle %.3f" % ang[c, f], "kernel cost %.6f" % mine, "numpy/scipy from the homography start %.6f" % c_ref, "from the truth %.6f" % c_tr, "pose diff kernel-vs-ref %.2e" % np.abs(per_cam[c, f] - x_ref).max())
    # where do the mirrored frames of the bundle-adjusted solution come from?  camera-frame board rotations of calibrate()'s outputs (the start), of
    # bundle_adjust from them (a) and of bundle_adjust from a perturbed truth (b)
    seen = ~np.isnan(poses).any(1)
    with contextlib.redirect_stdout(io.StringIO()):
        a = mc.bundle_adjust(p["uvs"][:, seen], ext, intr, p["obj"], poses[seen], n_frames=None, outlier_threshold=1e30, ftol=1e-13, xtol=1e-13, gtol=1e-11, max_nfev=300, verbose=0, return_jac=False)
        b = mc.bundle_adjust(p["uvs"][:, seen], p["extrinsics"], p["intrinsics"], p["obj"], p["poses"][seen], n_frames=None, outlier_threshold=1e30, ftol=1e-13, xtol=1e-13, gtol=1e-11, max_nfev=300, verbose=0, return_jac=False)

    def cam_board(e, q):
        return co.rodrigues_batch(np.asarray(e)[:, None, :3]) @ co.rodrigues_batch(np.asarray(q)[None, :, :3])

    def angle(X, Y):
        return np.arccos(np.clip((np.einsum("cfij,cfij->cf", X, Y) - 1) / 2, -1, 1))
    R0, Ra, Rb = cam_board(ext, poses[seen][a[3]]), cam_board(a[0], a[2]), cam_board(b[0], b[2])
    print("   costs a %.4f b %.4f; frames with a camera-frame rotation > 0.15 rad away from b's: in the start %d, after bundle_adjust from it %d (max %.3f / %.3f)"
          % (a[4].cost, b[4].cost, int((angle(R0, Rb).max(0) > 0.15).sum()), int((angle(Ra, Rb).max(0) > 0.15).sum()), angle(R0, Rb).max(), angle(Ra, Rb).max()))
    k = np.array([np.r_[K[0, 0], K[1, 1], K[0, 2], K[1, 2], d[:2]] for K, d in a[1]]); kb = np.array([np.r_[K[0, 0], K[1, 1], K[0, 2], K[1, 2], d[:2]] for K, d in b[1]])
    print("   intrinsics a", k.round(3).tolist(), "\n   intrinsics b", kb.round(3).tolist(), "\n   start", intr9[:, :6].round(3).tolist())
