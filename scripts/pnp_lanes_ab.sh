#!/bin/bash
# A/B on one box: k_pnp with one lane per view against four lanes per view (MCBA_PNP_LANES), calibrate() wall time and its stages.
# usage (on the GPU box): bash scripts/pnp_lanes_ab.sh <tag>
set -e
TAG=${1:-x}
for lanes in 1 4 1 4 0; do
  MCBA_PNP_LANES=$lanes MCBA_CAL_REPS=7 python scripts/calibrate_time.py 6,2130,5,7 6,10000,6,9 2,50,6,9 4,5000,6,9 > gpurun_out/pnp_lanes_${TAG}_${lanes}_$RANDOM.json
done
python - <<'PY'
import glob, json, os
tag = os.environ.get("TAG", "")
for f in sorted(glob.glob("gpurun_out/pnp_lanes_*json")):
    d = json.load(open(f))
    print(os.path.basename(f), {k: (round(v["calibrate_ms"], 3), round(v["stages_ms"].get("ops.calib_poses", 0), 3), round(v["stages_ms"].get("ops.calib_view_poses", 0), 3), round(v["stages_ms"].get("ops.calib_homographies", 0), 3)) if isinstance(v, dict) and "calibrate_ms" in v else None for k, v in d.items()})
PY
