#!/bin/bash
# scratch probe: host share of a step (update - in_match_reduce), three runs
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
for i in 1 2 3; do
  timeout 600 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-end-to-end --no-hbm-regime --no-crowded --streams 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d['host_us_per_step']
print('value',round(d['value']),'step_us',round(d['ms_per_step']*1e3,2),'host outside match_reduce',round(h['update']-h['in_match_reduce'],2),'pose',d.get('pose_err_vs_cpu'))"
done
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "ieskf or golden or degenerate" 2>&1 | grep -E "passed|failed"
