#!/bin/bash
# developer A/B of an environment switch on the bench (interleaved repeats): tools/ab_env.sh VAR "v1 v2 ..." [reps]
var=$1; vals=$2; reps=${3:-3}
for rep in $(seq $reps); do for v in $vals; do env $var=$v python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['stage_us_per_pass']; print('$var=$v', round(d['value']), round(d['ms_per_step']*1e3,1), 'knn', round(r['mean_launch_us'],2))"; done; done
