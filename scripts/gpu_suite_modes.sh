#!/bin/bash
# The GPU test tier in its five modes (default; relaxed reader of the release word; per-tick Python loop; unfilled buffers pre-set to NaN;
# pre-set to finite garbage), one pytest process each, one after the other.  usage (on the GPU box): bash scripts/gpu_suite_modes.sh TAG
TAG=${1:-fin}
run() {  # name, then env assignments
  local name=$1; shift
  env "$@" timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/${TAG}_${name}.txt 2>&1
  local rc=$?
  echo "$name rc=$rc $(tail -n 1 gpurun_out/${TAG}_${name}.txt)"
  return $rc
}
run default MCBA_NOOP=1 && run relaxed MCBA_STRICT_SYNC=0 && run hostloop MCBA_HOST_LOOP=1 && run poison1 MCBA_POISON=1 && run poison2 MCBA_POISON=2
