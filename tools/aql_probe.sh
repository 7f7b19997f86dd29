#!/bin/bash
# builds tools/aql_probe (+ its code object) here; run on the GPU box: gpurun -- 'tools/aql_probe tools/aql_probe.hsaco'
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 --cuda-device-only --no-gpu-bundle-output -O3 tools/aql_probe_kernel.hip -o tools/aql_probe.hsaco 2>&1 | grep -E "error" | head
/opt/rocm/bin/hipcc -O2 -std=c++17 tools/aql_probe.cpp -o tools/aql_probe -L/opt/rocm/lib -lhsa-runtime64 2>&1 | grep -E "error|undefined" | head -20
ls -la tools/aql_probe tools/aql_probe.hsaco
