"""Timeline of the LM tick's kernels from a rocprofv3 --kernel-trace csv: per kernel of a steady-state tick its duration and the gap
to the previous kernel's end (= what a kernel boundary really costs in the stream), averaged over the ticks found.
usage: python scripts/tick_timeline.py <kernel_trace.csv> [first-kernel-substring, default k_gram]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "k_gram"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(r["Kernel_Name"].split("(")[0].replace("void mcba::", "").replace("mcba::", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# ticks = runs that start with `first`
starts = [i for i, e in enumerate(ev) if first in e[0]]
ticks = [ev[a:b] for a, b in zip(starts[:-1], starts[1:])]
ticks = ticks[len(ticks) // 2:]            # steady state: the second half of the run
from collections import Counter
shape = Counter(tuple(e[0] for e in t) for t in ticks).most_common(1)[0][0]
sel = [t for t in ticks if tuple(e[0] for e in t) == shape]
print(f"{len(sel)} ticks of shape {len(shape)} kernels")
n = len(sel)
tot = 0.0
for k, name in enumerate(shape):
    dur = sum(t[k][2] - t[k][1] for t in sel) / n / 1e3
    gap = sum((t[k][1] - (t[k - 1][2] if k else 0)) for t in sel) / n / 1e3 if k else float("nan")
    print(f"  {name[:60]:60s} duration {dur:7.2f} us   gap before it {gap:6.2f} us")
    tot += dur + (gap if k else 0)
period = sum(b[0][1] - a[0][1] for a, b in zip(sel[:-1], sel[1:]) if b[0][1] - a[0][1] < 1e6) / max(1, len(sel) - 1) / 1e3
print(f"  tick period (start to start of consecutive selected ticks, incl. the gap to the next tick): {period:.2f} us")
