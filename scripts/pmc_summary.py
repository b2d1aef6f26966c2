"""Summarise rocprofv3 --pmc CSV output(s) per kernel.

usage: python scripts/pmc_summary.py OUT.json DIR [DIR ...]
Every *counter_collection.csv below the given directories is read; per (kernel, counter) the mean over
dispatches is reported.  HBM traffic per launch follows /opt/skills/guides/MI355X_MICROARCH.md section HBM:
  FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide
  (16 B/lane) coalesced streaming read, so reads = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact for 16 B/lane
  streaming stores, so writes = WRITE_SIZE * 1024.  The two counters are collected in separate passes."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(k_[a-z_]+)", name)
    return m.group(1) if m else name


def main(out, dirs):
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    res = {}
    for k, cs in sorted(acc.items()):
        e = {c: sum(v) / len(v) for c, v in cs.items()}
        e["dispatches"] = max(len(v) for v in cs.values())
        if "FETCH_SIZE" in e or "WRITE_SIZE" in e:
            rd = 2.0 * e.get("FETCH_SIZE", 0.0) * 1024.0
            wr = e.get("WRITE_SIZE", 0.0) * 1024.0
            e["hbm_read_bytes_per_launch"] = rd
            e["hbm_write_bytes_per_launch"] = wr
            e["hbm_bytes_per_launch"] = rd + wr
        res[k] = e
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    for k, e in res.items():
        print(k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in e.items()})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
