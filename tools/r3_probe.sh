#!/bin/bash
# scratch probe: host-side stage times of the map insert during the bench's end-to-end leg
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
FLIMO_PROF_INSERT=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hbm-regime --streams 0 2> gpurun_out/probe_insert.err > gpurun_out/probe_insert.json
grep "flimo insert" gpurun_out/probe_insert.err | sed -n 2,12p
