"""Per-wavefront stamps of k_gram (development aid; needs a build with MCBA_HIPCC_FLAGS=-DMCBA_GRAM_TIMING).
usage: python scripts/gram_stamps.py [lib.so] [C,F,rows,cols]      (MCBA_GRAM_SPLIT / MCBA_GRAM_NPW select the variant)
Round 6: the stamps (and the mcba_debug_* export this script reads) live in profiles/round6/patches/experiments_and_stamps.patch, not in the product sources:
`cd multicam-calibration_amd && git apply -p0 ../profiles/round6/patches/experiments_and_stamps.patch` first."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, ".")
import multicam_calibration_amd as m

args = [a for a in sys.argv[1:]]
if args and args[0].endswith(".so"):
    m.ops.LIB_PATH = args.pop(0)
C, F, rows, cols = (int(v) for v in args[0].split(",")) if args else (6, 10000, 6, 9)
p = m.synth.make_problem(C, F, rows=rows, cols=cols, seed=0)
x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
prob = m.ops.Problem(p["uvs"], p["obj"])
prob.set_params(0, x0)
for _ in range(200):
    prob.linearize(0)
prob.synchronize()
nfb = (F + 63) // 64
out = np.zeros((C, nfb, 32))
f = prob.lib.mcba_debug_gram_stamps
f.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
f.restype = ctypes.c_int
assert f(prob.handle, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))) == 0
if out[..., 8:].any():  # point split inside the workgroup: 8 stamps per part
    npw = 4 if out[..., 24:].any() else 2
    st = out[..., : 8 * npw].reshape(C, nfb, npw, 8)
    t0 = st[..., 4].min()
    for part in range(npw):
        s = st[:, :, part]
        print("part %d | cycles: set-up med %.0f  loop med %.0f max %.0f  exchange+barrier med %.0f max %.0f  finish med %.0f max %.0f | wall us: start %.2f..%.2f end med %.2f max %.2f" % (
            part, np.median(s[..., 0]), np.median(s[..., 1]), s[..., 1].max(), np.median(s[..., 2]), s[..., 2].max(), np.median(s[..., 3]), s[..., 3].max(),
            (s[..., 4].min() - t0) / 100, (s[..., 4].max() - t0) / 100, (np.median(s[..., 5]) - t0) / 100, (s[..., 5].max() - t0) / 100))
    print("kernel entry -> body start: med %.2f us max %.2f us; first entry -> last end %.2f us" % (np.median(st[..., 4] - st[..., 6]) / 100, (st[..., 4] - st[..., 6]).max() / 100, (st[..., 5].max() - st[..., 6].min()) / 100))
    cyc = st[..., :4].sum(-1)
    print("implied shader clock: %.3f GHz" % np.median(cyc / ((st[..., 5] - st[..., 4]) * 10.0)))
    print("workgroups %d; XCC histogram %s" % (C * nfb, np.bincount(st[:, :, 0, 7].astype(np.int64).ravel()).tolist()))
    prob.close()
    sys.exit(0)
out = out[..., :8]
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
