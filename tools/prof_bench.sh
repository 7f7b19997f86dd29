#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_bench.sh <tag> [bench args...]
# rocprofv3 kernel-trace summary of the bench command; prints the top kernels (name truncated).
tag=${1:-run}; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/bench.py "$@" > $out.json 2>/dev/null
cd $root
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    name = r["Name"].replace("void ", "").replace("flimo::", "")[:48]
    print(f'{name:48s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us  min {float(r["MinNs"])/1e3:7.2f}  max {float(r["MaxNs"])/1e3:7.2f}  {r["Percentage"]:>6s}%')
PY
