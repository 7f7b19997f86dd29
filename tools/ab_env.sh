#!/bin/bash
# developer A/B of an environment switch on the short bench (interleaved repeats): tools/ab_env.sh VAR "v1 v2 ..." [reps]
var=$1; vals=$2; reps=${3:-3}
for rep in $(seq $reps); do for v in $vals; do env $var=$v python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-end-to-end --no-hbm-regime --no-crowded --streams 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$var=$v', 'value', round(d['value']), 'us/step', round(d['ms_per_step']*1e3,1), 'pass_us', round(r['mean_launch_us'],2))"; done; done
