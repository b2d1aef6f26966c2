"""Instruction mix of the main (hottest) loop of a kernel in a hipcc -S listing.
usage: python scripts/isa_count.py listing.s mangled_substring
The hottest loop = the longest backward-branch body (label .. s_cbranch to that label)."""
import re
import sys
from collections import Counter

lst, key = sys.argv[1], sys.argv[2]
lines = open(lst).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^[A-Za-z_0-9]*%s[A-Za-z_0-9]*:" % re.escape(key), l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB[0-9_]+):", l))}
best = None
for i, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB[0-9_]+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        span = (labels[m.group(1)], i)
        if best is None or span[1] - span[0] > best[1] - best[0]:
            best = span
print("kernel lines %d, hottest loop lines %d..%d" % (len(body), best[0], best[1]))
ops = Counter()
for l in body[best[0]:best[1] + 1]:
    t = l.strip().split()
    if not t or t[0].startswith((".", ";")) or t[0].endswith(":"):
        continue
    ops[t[0]] += 1
cat = Counter()
for op, n in ops.items():
    if op.startswith("v_") and "_f64" in op:
        cat["valu_f64"] += n
    elif op.startswith("v_accvgpr"):
        cat["accvgpr_mov"] += n
    elif op.startswith("v_mfma"):
        cat["mfma"] += n
    elif op.startswith("v_"):
        cat["valu_other"] += n
    elif op.startswith("s_"):
        cat["salu/ctl"] += n
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        cat["vmem"] += n
    elif op.startswith("ds_"):
        cat["lds"] += n
    else:
        cat["other"] += n
print(dict(cat), "total", sum(cat.values()))
whole = Counter()
for l in body:
    t = l.strip().split()
    if t and not t[0].startswith((".", ";")) and not t[0].endswith(":"):
        whole["valu_f64" if (t[0].startswith("v_") and "_f64" in t[0]) else "valu_other" if t[0].startswith("v_") else "other"] += 1
print("whole kernel:", dict(whole), "| outside the hottest loop:", {k: whole[k] - (cat.get(k, 0) if k != "valu_other" else cat.get("valu_other", 0) + cat.get("accvgpr_mov", 0)) for k in ("valu_f64", "valu_other")})
print(ops.most_common(40))
