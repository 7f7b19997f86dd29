#!/bin/bash
# usage (GPU box): tools/prof_full_path.sh <tag>   -- rocprofv3 kernel summary of tests/dev/gpu_full_path.py (GPU only)
tag=${1:-fp}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
ORACLE=0 NSCANS=${NSCANS:-8} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/tests/dev/gpu_full_path.py > $out.log 2>&1
cd $root
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    name = r["Name"].replace("void ", "").replace("flimo::", "")
    if "rocprim" in name:
        i = name.find("detail::", name.find("trampoline")) 
        name = "rocprim:" + name[name.find("wrapped_") if "wrapped_" in name else 0:][:40]
    print(f'{name[:52]:52s} calls {r["Calls"]:>5s} total {float(r["TotalDurationNs"])/1e3:9.1f} us avg {float(r["AverageNs"])/1e3:8.2f} us')
PY
grep "^scan" $out.log | tail -3
