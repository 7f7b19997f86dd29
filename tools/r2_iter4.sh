#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/iter_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/iter_tests.log | tail -3
grep -E "^(FAILED|ERROR)" gpurun_out/iter_tests.log | head -20
for combo in "FLIMO_X=0" "FLIMO_FUSE=0"; do
  echo "== pass times with $combo"; env $combo timeout 300 python tests/dev/gpu_pass_times.py 2>&1 | grep "^level" | grep -v "level 2"
done
X0=tstar timeout 200 python tools/gpu_trace.py 2>&1 | grep -v "^    " | grep -v xcd | tail -20
