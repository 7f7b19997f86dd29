#!/bin/bash
# round 3, first GPU check: suite + default bench + hbm-regime leg with the old / new straggler bound
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3_pytest.log
timeout 600 python bench.py --steps 50 --warmup 5 2> gpurun_out/r3_bench.err > gpurun_out/r3_bench.json; echo "bench rc=$?"; head -c 700 gpurun_out/r3_bench.json; echo
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_bench.json'))
r=d['roofline']; print('frac',r['frac'],'us',r['mean_launch_us']); print('stage',r.get('knn_stage_separate_dispatches'))
print('hbm', json.dumps(r.get('hbm_regime'))[:3000])
print('e2e', d['end_to_end']['ms'], 'cpu', d['cpu_baseline']['value'], 'host_us', d['host_us_per_step'])
PY
FLIMO_TAIL_MAX=1024 timeout 600 python bench.py --hbm-regime-only --no-cpu-baseline 2>/dev/null > gpurun_out/r3_hbm_tail1024.json; head -c 1500 gpurun_out/r3_hbm_tail1024.json; echo
