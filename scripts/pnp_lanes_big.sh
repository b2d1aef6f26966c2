#!/bin/bash
# k_pnp, one lane against four lanes per view at shapes beyond one wavefront per SIMD (dense launch = ops.calib_poses)
set -e
for lanes in 1 4 1 4; do
  MCBA_PNP_LANES=$lanes MCBA_CAL_REPS=5 python scripts/calibrate_time.py 6,20000,6,9 12,20000,6,9 24,12500,10,20 > gpurun_out/pnp_big_${lanes}_$RANDOM.json
done
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob("gpurun_out/pnp_big_*json")):
    d = json.load(open(f))
    print(os.path.basename(f), {k: (round(v["calibrate_ms"], 3), round(v["stages_ms"].get("ops.calib_poses", 0), 3)) for k, v in d.items()})
PY
