#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
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
run default_notiming FLIMO_BENCH_TIMING=0
run nofuse_notiming FLIMO_FUSE=0 FLIMO_BENCH_TIMING=0
run old_notiming FLIMO_TAIL=0 FLIMO_FIT2=0 FLIMO_BENCH_TIMING=0
FLIMO_PROF_PASS=1 FLIMO_BENCH_TIMING=0 timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>&1 >/dev/null | grep "flimo pass" | tail -2
