"""Compile libmcba.so (HIP, gfx950) in-tree with hipcc.  `python -m multicam_calibration_amd.build`."""
import hashlib
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libmcba.so")
SOURCES = ["mcba_kernels.hip", "mcba_solve.hip", "mcba_triangulate.hip", "mcba_diag.hip", "mcba_calib.hip", "mcba_pnp.hip", "mcba_api.hip"]
DEPS = SOURCES + ["mcba_gram_finish.inc", "mcba_math.h", "mcba_pnp_math.h", "mcba_device.h", "mcba_backsub.h", "mcba_kernels.h", "mcba_lm.h", "mcba_lm_state.h", os.path.join("..", "..", "include", "mcba.h")]


# Register-pressure-bound kernels (k_gram keeps 87 FP64 accumulators per lane): LLVM's "unclustered high register pressure"
# rescheduling stage and its default local assignment order cost k_gram 24 VALU instructions per point-observation in moves
# between the two register files (1529 -> 1427 instructions per 4-point loop body; measured 54.9 -> 52.4 us on MI355X; the
# other kernels are unaffected).
SCHED_FLAGS = ["-mllvm", "-amdgpu-disable-unclustered-high-rp-reschedule=1", "-mllvm", "-greedy-reverse-local-assignment=1"]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libmcba.so cannot be built (ROCm toolchain required)")


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=False, out=None):
    """hipcc --offload-arch=gfx950 -> multicam-calibration_amd/libmcba.so (cross-compiles without a GPU).  One object per source,
    compiled in parallel and only when stale (an object depends on its source and on every header), then linked.
    out: another output path (development: A/B builds for scripts/gram_ab.py, selected at run time with MCBA_LIB)."""
    if out is None and not force and not is_stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor

    extra = os.environ.get("MCBA_HIPCC_FLAGS", "").split()  # development only (e.g. -DMCBA_SOLVE_TIMING, -save-temps)
    common = [hipcc_path(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC"] + SCHED_FLAGS + extra
    # one object directory per (output, flags): named by a digest that is the same in every process (Python's hash() of a string is not),
    # so that an A/B build finds its objects again and a change of flags never re-uses objects compiled with other ones
    tag = hashlib.sha1(repr((out, common[1:])).encode()).hexdigest()[:12]
    objdir = os.path.join(CSRC, ".obj" if out is None and not extra else ".obj_" + tag)
    os.makedirs(objdir, exist_ok=True)
    flags_file = os.path.join(objdir, "flags.txt")
    flags_now = " ".join(common[1:])
    flags_same = os.path.exists(flags_file) and open(flags_file).read() == flags_now
    headers = [os.path.join(CSRC, d) for d in DEPS if d not in SOURCES]
    hdr_time = max(os.path.getmtime(hh) for hh in headers)

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        srcp = os.path.join(CSRC, src)
        if not force and flags_same and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(srcp), hdr_time):
            return obj
        tmp = obj + ".%d.tmp" % os.getpid()   # (several ranks may build at once: nobody ever sees a half-written object)
        cmd = common + ["-c", "-o", tmp, srcp]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)
        os.replace(tmp, obj)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 2)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    with open(flags_file, "w") as fh:
        fh.write(flags_now)
    target = out or LIB
    tmp_lib = target + ".%d.tmp" % os.getpid()
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", tmp_lib, "-ldl"] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(tmp_lib, target)
    return target


if __name__ == "__main__":
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
    print(build(force="--force" in sys.argv, verbose=True, out=out))
