#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
for combo in "FLIMO_TAIL=1 FLIMO_FIT2=0" "FLIMO_TAIL=0 FLIMO_FIT2=1" "FLIMO_TAIL=1 FLIMO_FIT2=1"; do
  echo "== corridor test with $combo"; env $combo timeout 300 python -m pytest tests/test_gpu_sequence.py -m gpu -x -q -k corridor 2>&1 | grep -E "passed|failed|AssertionError" | head -3
done
for combo in "FLIMO_X=0" "FLIMO_TAIL=0" "FLIMO_TAIL=0 FLIMO_WIDEN_TIGHT=1" "FLIMO_TAIL=0 FLIMO_WIDEN_R3=1" "FLIMO_TAIL=0 FLIMO_WIDEN_TIGHT=1 FLIMO_WIDEN_R3=1" "FLIMO_TAIL_PASS1=1" "FLIMO_FIT_PPW=64" "FLIMO_FIT2=0"; do
  echo "== pass times with $combo"; env $combo timeout 300 python tests/dev/gpu_pass_times.py 2>&1 | grep "^level"
done
for combo in "FLIMO_FIT_PPW=64" "FLIMO_FIT_PPW=32" "FLIMO_FIT2=0"; do
  echo "== trace $combo X0=tstar"; env $combo X0=tstar timeout 300 python tools/gpu_trace.py 2>&1 | grep -A12 "kernel fit" | head -14
done
echo "== trace knn X0=tstar"; X0=tstar timeout 300 python tools/gpu_trace.py 2>&1 | grep -B0 -A8 "kernel knn5" | head -10
