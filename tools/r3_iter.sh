#!/bin/bash
# round 3 iteration: GPU suite (stop at first failure) + default bench summary
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q $PYTEST_ARGS > gpurun_out/r3_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|error|assert" gpurun_out/r3_pytest.log | tail -15
timeout 600 python bench.py --steps 50 --warmup 5 $BENCH_ARGS 2> gpurun_out/r3_bench.err > gpurun_out/r3_bench.json; echo "bench rc=$?"; tail -3 gpurun_out/r3_bench.err
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/r3_bench.json'))
except Exception as e:
    print("no bench json", e); raise SystemExit
r=d['roofline']; print('value',d['value'],'ms',d['ms_per_step'],'frac',r['frac'],'us',r['mean_launch_us'], 'bitrepro', d['config']['steps_bit_reproducible'])
print('stage',r.get('stage'))
print('knn_stage',{k:v for k,v in (r.get('knn_stage_separate_dispatches') or {}).items() if k!='note'})
h=r.get('hbm_regime') or {}
print('hbm', {k:h.get(k) for k in ('ms_per_step','passes_per_step','passes_in_one_launch','passes_total','stragglers_last_pass','one_launch_pass_us','separate_dispatch_pass_us','E_evals_per_query','frac','knn_stage_separate_dispatches','pose_err_vs_cpu')})
print('e2e', (d.get('end_to_end') or {}).get('ms'), 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'host_us', d['host_us_per_step'], 'pose', d.get('pose_err_vs_cpu'))
print('streams', (d.get('concurrent_streams') or {}).get('scans_per_s_aggregate'), 'insert', d.get('with_map_insert'))
cr=r.get('crowded') or {}; print('crowded', {k:cr.get(k) for k in ('clean_map_us','crowded_map_us','first_pass_ratio','later_passes_ratio','insert_ms_per_sweep')})
PY
