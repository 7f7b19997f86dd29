#!/bin/bash
# scratch probe: the 256k x 20M leg with the straggler threshold forced open / shut
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
for tm in 0 3000 8000 1000000; do
  echo "== FLIMO_TAIL_MAX=$tm"
  FLIMO_TAIL_MAX=$tm timeout 600 python bench.py --hbm-regime-only --no-cpu-baseline --hbm-steps 10 2>/dev/null | python -c "
import json,sys
h=json.loads(sys.stdin.read())['roofline']['hbm_regime']
print({k:h.get(k) for k in ('ms_per_step','passes_in_one_launch','passes_total','stragglers_last_pass','one_launch_pass_us','separate_dispatch_pass_us')})"
done
