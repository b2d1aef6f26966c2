"""FP64 instruction mix of k_gram's point loop, counted in the ISA of the built kernel -> profiles/gram_flops.json, the constants
bench.py prices the kernel with (real flops = 2 FMA + MUL + ADD + transcendental seeds; issue slots = every FP64 instruction).
usage: python scripts/gram_isa_mix.py [pmc_summary.json]     (run in the build container; needs hipcc, no GPU)
The optional PMC summary (scripts/pmc_summary.py output holding SQ_INSTS_VALU_{FMA,MUL,ADD}_F64 of k_gram) is the cross-check:
the per-(camera, frame) work outside the loop is branchy (both sides of every branch are in the listing), so its DYNAMIC count
is taken from the counters: outside = (counter total - loop share) / wavefronts."""
import json
import os
import re
import subprocess
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multicam_calibration_amd import build  # noqa: E402

KERNEL = "k_gramILi1ELb1E"   # soft_l1, planar board + unit f_scale: the instance the bench workload runs
LISTING = "/tmp/mcba_kernels.s"
C, F, N = 6, 10000, 54
WAVES = C * ((F + 63) // 64)


def listing():
    cmd = [build.hipcc_path(), "-O3", "--offload-arch=gfx950", "-std=c++17"] + build.SCHED_FLAGS + ["-S", "--cuda-device-only", "-o", LISTING, os.path.join(build.CSRC, "mcba_kernels.hip")]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    lines = open(LISTING).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^[A-Za-z_0-9]*%s[A-Za-z_0-9]*:" % re.escape(KERNEL), l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start:end]


def classify(op):
    if op.startswith("v_fma_f64") or op.startswith("v_fmac_f64"):
        return "fma"
    if op.startswith("v_mul_f64"):
        return "mul"
    if op.startswith("v_add_f64"):
        return "add"
    if op.startswith("v_") and "_f64" in op:
        return "other_f64"   # v_rcp_f64, v_rsq_f64, compares, min / max, conversions
    if op.startswith("v_accvgpr"):
        return "accvgpr_mov"
    if op.startswith("v_"):
        return "valu_other"
    return "non_valu"


def count(lines):
    c = Counter()
    for l in lines:
        t = l.strip().split()
        if not t or t[0].startswith((".", ";")) or t[0].endswith(":"):
            continue
        c[classify(t[0])] += 1
    return c


def main():
    body = listing()
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB[0-9_]+):", l))}
    best = None
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB[0-9_]+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i and "Loop Header" in body[labels[m.group(1)]]:   # (other backward branches are exits)
            if best is None or i - labels[m.group(1)] > best[1] - best[0]:
                best = (labels[m.group(1)], i)
    loop = count(body[best[0]:best[1] + 1])
    pts_per_iter = 4   # the software-pipelined loop body covers four points (RD = 4)
    per_point = {k: loop[k] / pts_per_iter for k in ("fma", "mul", "add", "other_f64", "accvgpr_mov", "valu_other")}
    per_point["fp64"] = per_point["fma"] + per_point["mul"] + per_point["add"] + per_point["other_f64"]
    per_point["valu"] = per_point["fp64"] + per_point["accvgpr_mov"] + per_point["valu_other"]
    per_point["flop"] = 2 * per_point["fma"] + per_point["mul"] + per_point["add"] + per_point["other_f64"]
    out = {"kernel": "k_gram<soft_l1, planar + unit f_scale>", "source": "scripts/gram_isa_mix.py on hipcc -S of csrc/mcba_kernels.hip (loop of %d lines = %d points)" % (best[1] - best[0], pts_per_iter),
           "per_point_observation": per_point,
           "per_pair_outside_loop": {"fp64": 1900.0, "flop": 1900.0 * per_point["flop"] / per_point["fp64"], "source": "static estimate (round 2: ~1900 FP64 instructions per (camera, frame) outside the loop), priced with the loop's flop / instruction ratio"}}
    if len(sys.argv) > 1:
        pmc = json.load(open(sys.argv[1]))
        g = pmc.get("k_gram", pmc)
        fma, mul, add = g["SQ_INSTS_VALU_FMA_F64"], g["SQ_INSTS_VALU_MUL_F64"], g["SQ_INSTS_VALU_ADD_F64"]
        pts = C * (-(-F // 64) * 64) * N   # every lane of every wavefront runs the loop, padding frames included
        out["pmc_cross_check"] = {
            "source": sys.argv[1], "wave_instructions_per_launch": {"fma": fma, "mul": mul, "add": add},
            "flop_per_launch": 64 * (2 * fma + mul + add),
            "loop_share_from_isa": {k: per_point[k] * pts / 64 for k in ("fma", "mul", "add")},
        }
        outside = {k: (v - per_point[k] * pts / 64) / WAVES for k, v in (("fma", fma), ("mul", mul), ("add", add))}
        fp64 = sum(outside.values())
        out["per_pair_outside_loop"] = {"fp64": fp64, "flop": 2 * outside["fma"] + outside["mul"] + outside["add"], "fma": outside["fma"], "mul": outside["mul"], "add": outside["add"],
                                        "source": "dynamic: (PMC FMA / MUL / ADD wave-instructions per launch - the loop's share) / %d wavefronts" % WAVES}
    path = os.path.join(ROOT, "profiles", "gram_flops.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
