"""Register / LDS / scratch usage of every kernel in a .hip source (hipcc -Rpass-analysis=kernel-resource-usage), one line each.
usage: python scripts/kernel_resources.py [source.hip ...]   (default: the three kernel files of csrc/)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "multicam-calibration_amd", "csrc")
srcs = sys.argv[1:] or [os.path.join(CSRC, f) for f in ("mcba_kernels.hip", "mcba_solve.hip", "mcba_triangulate.hip")]
for src in srcs:
    out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage", "-mllvm", "-amdgpu-disable-unclustered-high-rp-reschedule=1", "-mllvm", "-greedy-reverse-local-assignment=1"]
                         + os.environ.get("MCBA_HIPCC_FLAGS", "").split(), capture_output=True, text=True).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r"remark: [^ ]* +(Function Name|[A-Za-z ]+): *(.*?) *\[-Rpass", line)
        if not m:
            m = re.search(r": +([A-Za-z ]+\[?[A-Za-z/]*\]?): *(.*?) *\[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k in ("Function Name", "Name"):
            if cur:
                print(cur)
            name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
            cur = {"kernel": re.sub(r"\(.*", "", name).replace("void mcba::", "")}
        else:
            cur[k] = v
    if cur:
        print(cur)
