#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
for env in "FLIMO_DBG=0" "FLIMO_DBG=1" "FLIMO_DBG=2" "FLIMO_DBG=4" "FLIMO_DBG=7"; do
  echo "== $env"
  env $env timeout 300 python tests/dev/gpu_pass_times.py 2>&1 | grep -E "level 1"
done
