#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
for env in "X=1" "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1" "HSA_ENABLE_INTERRUPT=0" "GPU_MAX_HW_QUEUES=1" "DEBUG_CLR_LIMIT_BLIT_WG=1"; do
  echo "== $env"
  env $env timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-end-to-end --no-hbm-regime --streams 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step']*1e3,1), d['host_us_per_step'])"
done
