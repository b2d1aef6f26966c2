"""k_triangulate throughput (development aid): C cameras x P points, kernel time by HIP events."""
import sys
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import multicam_calibration_amd as m
from test_triangulate_cpu import scene

C = int(sys.argv[1]) if len(sys.argv) > 1 else 6
P = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
uvs, ext, intr, X = scene(C=C, P=P, seed=1, noise=0.2, p_unseen=0.1)
for rep in range(3):
    out, ms = m.triangulate(uvs, ext, intr, return_kernel_ms=True)
pairs = C * (C - 1) // 2
print("C=%d P=%d: kernel %.3f ms -> %.1f M points/s, %.1f M pair-triangulations/s; HBM %.1f GB/s (16 B in per view + 24 B out per point)" % (
    C, P, ms, P / ms / 1e3, P * pairs / ms / 1e3, (16 * C + 24) * P / ms / 1e6))
print("median |error| vs truth (0.2 px noise): %.3g mm" % np.nanmedian(np.abs(out - X)))
