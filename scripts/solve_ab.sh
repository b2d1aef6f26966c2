# A/B of reduced-solve builds at the configs[4] shard (24 x 6 250 x 200): per build the LM tick and k_solve_cam by HIP events, with the stager /
# helper workgroups and without (MCBA_SOLVE_STAGERS=0).  usage: bash scripts/solve_ab.sh lib1.so lib2.so ...   (run through gpurun; one box)
for rep in 1 2; do
for lib in "$@"; do for st in default 0; do
  if [ "$st" = "default" ]; then unset MCBA_SOLVE_STAGERS; else export MCBA_SOLVE_STAGERS=$st; fi
  R=$(MCBA_LIB=$lib MCBA_SHAPES="24,6250,10,20" timeout -k 10 120 python scripts/other_shapes.py 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for k,x in d.items():
    if isinstance(x,dict): print(x['us_per_iteration'], x['kernels_us_by_hip_events']['k_solve_cam'])
")
  echo "$(basename $lib) stagers=$st tick,solve: $R"
done; done; done
