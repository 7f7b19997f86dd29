#!/bin/bash
# usage (GPU box): tools/pmc_collect.sh <tag> "<COUNTER1 COUNTER2 ...>"   -- one rocprofv3 --pmc pass of a short bench run;
# prints the per-launch mean of every counter for the three per-pass kernels (developer tool).
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_$tag
rm -rf $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $out -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $root
python3 - "$out" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(set)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "knn5" if "knn5_kernel" in k else "widen" if "widen_kernel" in k else "fit" if "fit_kernel" in k else None
        if not name: continue
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[name].add(r["Dispatch_Id"])
for name in ("knn5", "widen", "fit"):
    n = max(len(cnt[name]), 1)
    print(name, "launches", n, " ".join(f"{c}={v / n:.4g}" for c, v in sorted(acc[name].items())))
PY
