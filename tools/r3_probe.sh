#!/bin/bash
# scratch probe: host-side stage times of cloud materialisation during the bench's end-to-end leg
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
FLIMO_PROF_CLOUDS=1 FLIMO_PROF_DESKEW=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-hbm-regime --streams 0 2> gpurun_out/probe_clouds.err > gpurun_out/probe_clouds.json
grep "flimo clouds" gpurun_out/probe_clouds.err | sed -n 3,8p
grep "flimo clouds" gpurun_out/probe_clouds.err | tail -4
grep -i "deskew\]" gpurun_out/probe_clouds.err | sed -n 3,6p
grep -i "deskew\]" gpurun_out/probe_clouds.err | tail -3
