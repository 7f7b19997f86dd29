#!/bin/bash
# builds tools/ieskf_bench (with phase stamps) here; run it on the GPU box: gpurun -- tools/ieskf_bench 200
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DIESKF_STAMPS -Ifast_limo_amd/csrc/hip -Ifast_limo_amd/csrc/host -Iinclude \
  tools/ieskf_bench.hip fast_limo_amd/csrc/host/flimo_ikfom.cpp -o tools/ieskf_bench "$@" 2>&1 | grep -E "error|remark" | head -20
