#!/bin/bash
# Same box, alternating: the headline tick with the library of an older commit (a git worktree built next to this tree) against the current one.
# usage: bash scripts/lib_ab.sh <worktree-dir>
OLD=$1
for i in 1 2 3; do
  (cd $OLD && python bench.py --no-cpu-baseline --no-end-to-end --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old', round(d['value']), d['roofline']['avg_launch_us'], d['roofline']['live_issue_ceiling']['tflops'])")
  python bench.py --no-cpu-baseline --no-end-to-end --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new', round(d['value']), d['roofline']['avg_launch_us'], d['roofline']['live_issue_ceiling']['tflops'])"
done
