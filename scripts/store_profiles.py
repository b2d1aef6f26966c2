"""Copy one profile set of scripts/gpu_profile.sh from gpurun_out/ into profiles/roundN/ and derive profiles/pmc_traffic.json.

    python3 scripts/store_profiles.py TAG ROUND_DIR      e.g.  v14 round2
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag, rnd):
    src = os.path.join(ROOT, "gpurun_out")
    dst = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, f"{tag}_bench.json"), os.path.join(dst, f"bench_{tag}.json"))
    shutil.copy(os.path.join(src, f"{tag}_kernel_stats.csv"), os.path.join(dst, f"bench_default_kernel_stats_{tag}.csv"))
    shutil.copy(os.path.join(src, f"{tag}_pmc_summary.json"), os.path.join(dst, f"pmc_{tag}_summary.json"))
    summary = json.load(open(os.path.join(src, f"{tag}_pmc_summary.json")))
    keys = ("dispatches", "hbm_bytes_per_launch", "hbm_read_bytes_per_launch", "hbm_write_bytes_per_launch")
    traffic = {k: {q: v[q] for q in keys if q in v} for k, v in summary.items()}
    traffic["_source"] = (f"profiles/{rnd}/pmc_{tag}_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over "
                          "scripts/profile_kernels.py; reads = 2 x FETCH_SIZE x 1024 per the gfx950 rule)")
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
