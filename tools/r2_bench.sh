#!/bin/bash
# developer loop (GPU box): bench variants by env switches; prints one compact line each
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps ${STEPS:-50} --warmup 5 ${BENCH_ARGS:---no-cpu-baseline} > gpurun_out/iter_$tag.json 2> gpurun_out/iter_$tag.err
  python3 - gpurun_out/iter_$tag.json $tag <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]; s = r.get("stage", {}); e = d.get("end_to_end") or {}
    print(f"{sys.argv[2]:24s} value {d['value']:8.1f}  ms/step {d['ms_per_step']:.4f}  one-launch {r.get('mean_launch_us')}  separate {s.get('separate_dispatch_pass_us')}  frac {r.get('frac')}  e2e tied {e.get('tied_stamps',{}).get('ms_per_sweep')} unique {e.get('unique_stamps',{}).get('ms_per_sweep')}")
except Exception as ex:
    print(sys.argv[2], "FAILED", ex); print(open(sys.argv[1].replace('.json', '.err')).read()[-1500:])
PY
}
if [ $# -eq 0 ]; then run default FLIMO_X=0; else for v in "$@"; do run "$(echo $v | tr ' =' '__')" $v; done; fi
