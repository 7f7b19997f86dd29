#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_cmd.sh <tag> <python script> [args...]   (environment passes through)
# rocprofv3 kernel-trace summary of any python developer script; prints the top kernels.
tag=${1:-run}; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
script=$root/$1; shift
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $script "$@" > $out.log 2>&1
cd $root
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:18]:
    name = r["Name"].replace("void ", "").replace("flimo::", "")[:60]
    print(f'{name:60s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.2f} us  max {float(r["MaxNs"])/1e3:9.2f}  {r["Percentage"]:>6s}%')
PY
