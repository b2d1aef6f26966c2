"""Per-wavefront stamps of k_gram (development aid; needs a build with MCBA_HIPCC_FLAGS=-DMCBA_GRAM_TIMING).
usage: python scripts/gram_stamps.py [lib.so]"""
import ctypes
import sys

import numpy as np

sys.path.insert(0, ".")
import multicam_calibration_amd as m

if len(sys.argv) > 1:
    m.ops.LIB_PATH = sys.argv[1]
C, F = 6, 10000
p = m.synth.make_problem(C, F, seed=0)
x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
prob = m.ops.Problem(p["uvs"], p["obj"])
prob.set_params(0, x0)
for _ in range(200):
    prob.linearize(0)
prob.synchronize()
nfb = (F + 63) // 64
out = np.zeros((C, nfb, 8))
f = prob.lib.mcba_debug_gram_stamps
f.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
f.restype = ctypes.c_int
assert f(prob.handle, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))) == 0
pro, loop, epi, w0, w1, hwid, xcc = (out[..., i].ravel() for i in range(7))
t0 = w0.min()
print("waves %d | shader cycles: prologue %.0f..%.0f  loop min %.0f med %.0f max %.0f  epilogue med %.0f max %.0f" % (len(loop), pro.min(), pro.max(), loop.min(), np.median(loop), loop.max(), np.median(epi), epi.max()))
print("wall clock (100 MHz, 10 ns ticks): body start spread %.2f us; wave body duration min %.2f med %.2f max %.2f us; first start -> last end %.2f us" % (
    (w0.max() - t0) / 100, (w1 - w0).min() / 100, np.median(w1 - w0) / 100, (w1 - w0).max() / 100, (w1.max() - t0) / 100))
cyc = pro + loop + epi
print("implied shader clock: %.3f GHz (median over waves of cycles / wall time)" % np.median(cyc / ((w1 - w0) * 10.0)))
hw = hwid.astype(np.int64)
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
simd = (hw >> 4) & 0x3
key = xcc.astype(np.int64) * 1000 + se * 100 + sh * 20 + cu
print("distinct (xcc, se, sh, cu): %d; waves per CU max %d; per XCC: %s" % (len(set(key)), np.bincount(np.unique(key, return_inverse=True)[1]).max(), np.bincount(xcc.astype(np.int64)).tolist()))
per_simd = {}
for k, s_ in zip(key, simd):
    per_simd[(k, s_)] = per_simd.get((k, s_), 0) + 1
print("waves per SIMD histogram:", np.bincount(list(per_simd.values())).tolist())
order = np.argsort(loop)
print("slowest waves: loop cycles", loop[order[-5:]].astype(int).tolist(), "dur us", ((w1 - w0)[order[-5:]] / 100).round(2).tolist())
prob.close()
