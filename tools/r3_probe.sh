#!/bin/bash
# scratch probe (developer): edit freely
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
echo "=== clean map, first pass"; X0=tstar FIRSTPASS=1 timeout 300 python tools/gpu_trace.py 2>&1 | sed -n 1,40p
echo "=== crowded map, first pass"; X0=tstar FIRSTPASS=1 NINS=50 timeout 300 python tools/gpu_trace.py 2>&1 | sed -n 1,40p
