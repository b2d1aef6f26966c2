"""Copy one profile set of scripts/gpu_profile.sh from gpurun_out/ into profiles/roundN/ and derive profiles/pmc_traffic.json
and profiles/gram_flops.json (the two files bench.py reads).

    python3 scripts/store_profiles.py TAG ROUND_DIR      e.g.  v3 round3
"""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag, rnd):
    src = os.path.join(ROOT, "gpurun_out")
    dst = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    for a, b in ((f"{tag}_bench.json", f"bench_{tag}.json"), (f"{tag}_kernel_stats.csv", f"bench_default_kernel_stats_{tag}.csv"), (f"{tag}_pmc_summary.json", f"pmc_{tag}_summary.json"),
                 (f"{tag}_dist_kernel_stats.csv", f"forced_dist_kernel_stats_{tag}.csv"), (f"{tag}_dist_bench.json", f"forced_dist_bench_{tag}.json"),
                 (f"{tag}_e2e_breakdown.json", f"e2e_breakdown_{tag}.json"), (f"{tag}_other_shapes.json", f"other_shapes_{tag}.json"),
                 (f"{tag}_shard5_kernel_stats.csv", f"shard5_kernel_stats_{tag}.csv"),
                 (f"{tag}_config1_kernel_stats.csv", f"config1_kernel_stats_{tag}.csv"), (f"{tag}_c1free_kernel_stats.csv", f"shape_6x1000_free_kernel_stats_{tag}.csv"),
                 (f"{tag}_config0_kernel_stats.csv", f"config0_kernel_stats_{tag}.csv"), (f"{tag}_shard4_kernel_stats.csv", f"shard4_kernel_stats_{tag}.csv"),
                 (f"{tag}_pmc_config1_summary.json", f"pmc_config1_{tag}_summary.json"), (f"{tag}_config5_full.json", f"config5_full_one_gpu_{tag}.json"),
                 (f"{tag}_e2e_tutorial.json", f"e2e_breakdown_{tag}_tutorial.json"), (f"{tag}_e2e_config0.json", f"e2e_breakdown_{tag}_config0.json"),
                 (f"{tag}_e2e_missing.json", f"e2e_missing_{tag}.json"), (f"{tag}_e2e_kernel_stats.csv", f"e2e_kernel_stats_{tag}.csv"),
                 (f"{tag}_tick_timeline_config0.txt", f"tick_timeline_config0_{tag}.txt"), (f"{tag}_tick_timeline_config1.txt", f"tick_timeline_config1_{tag}.txt"),
                 (f"{tag}_tick_timeline_tutorial.txt", f"tick_timeline_tutorial_{tag}.txt"),
                 (f"{tag}_calibrate.json", f"calibrate_{tag}.json"), (f"{tag}_calibrate_kernel_stats.csv", f"calibrate_kernel_stats_{tag}.csv"),
                 (f"{tag}_calibrate_kernels.json", f"calibrate_dense_kernels_{tag}.json"), (f"{tag}_calibrate_dense_kernel_stats.csv", f"calibrate_dense_kernel_stats_{tag}.csv"),
                 (f"{tag}_pmc_calibrate_summary.json", f"pmc_calibrate_{tag}_summary.json")):
        if os.path.exists(os.path.join(src, a)):
            shutil.copy(os.path.join(src, a), os.path.join(dst, b))
        else:
            print("missing:", a)
    summary = json.load(open(os.path.join(src, f"{tag}_pmc_summary.json")))
    keys = ("dispatches", "hbm_bytes_per_launch", "hbm_read_bytes_per_launch", "hbm_write_bytes_per_launch")
    traffic = {k: {q: v[q] for q in keys if q in v} for k, v in summary.items()}
    traffic["_source"] = (f"profiles/{rnd}/pmc_{tag}_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over "
                          "scripts/profile_kernels.py; reads = 2 x FETCH_SIZE x 1024 per the gfx950 rule)")
    # rocprofv3's own average durations of the bench command's kernels in this set (bench.py prints them next to what it measures itself)
    stats = os.path.join(dst, f"bench_default_kernel_stats_{tag}.csv")
    if os.path.exists(stats):
        import csv
        import re

        avg = {}
        for row in csv.DictReader(open(stats)):
            mm = re.search(r"mcba::(k_[a-z0-9_]+)", row["Name"])
            if mm and mm.group(1) not in avg:   # (rows are sorted by total time: the first instance of a kernel is the loop's)
                avg[mm.group(1)] = {"avg_us": float(row["AverageNs"]) / 1e3, "calls": int(row["Calls"])}
        traffic["_rocprofv3_kernel_stats"] = {"kernels": avg, "source": f"profiles/{rnd}/bench_default_kernel_stats_{tag}.csv (rocprofv3 --kernel-trace --stats of `bench.py --no-cpu-baseline --no-end-to-end --no-other-configs`)"}
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1, sort_keys=True)
        f.write("\n")
    # FP64 instruction mix of k_gram: ISA of the CURRENT build + the FP64 counters of this profile set
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "gram_isa_mix.py"), os.path.join("profiles", rnd, f"pmc_{tag}_summary.json")], cwd=ROOT, stdout=subprocess.DEVNULL)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
