#!/bin/bash
# developer loop on the GPU box: GPU tests, bench variants (env switches), rocprofv3 kernel stats of the default build
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
mkdir -p gpurun_out
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/iter_tests.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/iter_tests.log | tail -3
  grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/iter_tests.log | head -20
fi
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
for v in "${VARIANTS[@]:-default}"; do :; done
run default FLIMO_X=0
run tail0_fit2_0 FLIMO_TAIL=0 FLIMO_FIT2=0
run tail1_fit2_0 FLIMO_TAIL=1 FLIMO_FIT2=0
run tail0_fit2_1 FLIMO_TAIL=0 FLIMO_FIT2=1
run ppw64 FLIMO_FIT_PPW=64
run ppw16 FLIMO_FIT_PPW=16
run default_again FLIMO_X=0
tools/prof_bench.sh r2iter --steps 50 --warmup 5 --no-cpu-baseline
