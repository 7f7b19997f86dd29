#!/bin/bash
# developer: same-box A/B of two builds of libflimo_hip.so (same C ABI): the short bench with each, interleaved.
# usage (inside gpurun): bash tools/ab_lib.sh gpurun_ab/libflimo_hip_base.so [rounds] [extra bench args]
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
base=$1; rounds=${2:-2}; shift; shift
cp fast_limo_amd/libflimo_hip.so gpurun_out/ab_new.so
for r in $(seq 1 $rounds); do
  for which in base new; do
    if [ $which = base ]; then cp $base fast_limo_amd/libflimo_hip.so; else cp gpurun_out/ab_new.so fast_limo_amd/libflimo_hip.so; fi
    timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-end-to-end --no-crowded --streams 0 "$@" 2> gpurun_out/ab_$which.err > gpurun_out/ab_$which.json
    python - $which <<'PY'
import json, sys
d = json.load(open('gpurun_out/ab_%s.json' % sys.argv[1])); r = d['roofline']; h = r.get('hbm_regime') or {}
st = r['stage']['dense_after_timed_region']
print(sys.argv[1], 'value %.0f' % d['value'], 'pass_us %.2f' % r['mean_launch_us'], 'first', {k: round(v, 1) for k, v in st['separate_dispatch_pass_us'].items()},
      'hbm ms %.3f pass %.1f' % (h.get('ms_per_step', 0), h.get('one_launch_pass_us', 0)), 'insert', (h.get('map_insert_ms') or {}).get('repeat'), 'idx', (h.get('index_bytes') or {}).get('over_map_bytes'))
PY
  done
done
cp gpurun_out/ab_new.so fast_limo_amd/libflimo_hip.so
