#!/bin/bash
# scratch probe: crowded-map first pass after the split walk; differential fuzz (probe forced in half of the trials)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
FINES=1 timeout 600 python tests/dev/gpu_crowded_bench.py 2>&1 | grep "kernels"
TRIALS=120 SEED=77 timeout 900 python tests/dev/gpu_fuzz.py 2>&1 | tail -2
