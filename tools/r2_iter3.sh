#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/iter_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/iter_tests.log | tail -3
grep -E "^(FAILED|ERROR)" gpurun_out/iter_tests.log | head -20
for combo in "FLIMO_X=0" "FLIMO_FUSE=0" "FLIMO_FUSE=0 FLIMO_FIT_PPW=64" "FLIMO_TAIL=0 FLIMO_FIT2=0"; do
  echo "== pass times with $combo"; env $combo timeout 300 python tests/dev/gpu_pass_times.py 2>&1 | grep "^level\|^fused" | grep -v "level 2"
done
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps ${STEPS:-50} --warmup 5 --no-cpu-baseline > gpurun_out/iter_$tag.json 2> gpurun_out/iter_$tag.err
  python3 - gpurun_out/iter_$tag.json $tag <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]; s = r.get("stage_us_per_pass", {})
    print(f"{sys.argv[2]:28s} value {d['value']:8.1f} scans/s  ms/step {d['ms_per_step']:.4f}  knn {s.get('knn')}  widen {s.get('widen')}  fit {s.get('fit_reduce')}  frac {r.get('frac')}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run default FLIMO_X=0
run nofuse FLIMO_FUSE=0
run nofuse_ppw64 FLIMO_FUSE=0 FLIMO_FIT_PPW=64
run old FLIMO_TAIL=0 FLIMO_FIT2=0
run default_notiming FLIMO_BENCH_TIMING=0
run old_notiming FLIMO_TAIL=0 FLIMO_FIT2=0 FLIMO_BENCH_TIMING=0
echo "== trace knn X0=tstar (fused)"; X0=tstar timeout 300 python tools/gpu_trace.py 2>&1 | grep -A9 "kernel knn5\|kernel fit" | head -24
