#!/bin/bash
# round-2 evidence (GPU box): default bench line, rocprofv3 kernel stats of the same command, PMC traffic passes
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-final}
cd $root; mkdir -p gpurun_out
# --streams 0: without the informational concurrent-streams leg, whose overlapped launches would enter the per-kernel averages
args="--steps 50 --warmup 5 --no-cpu-baseline --no-end-to-end --streams 0"
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/prof_$tag $root/gpurun_out/pmc_fetch_$tag $root/gpurun_out/pmc_write_$tag
# PMC passes first (counters in their own runs), so that the bench line below carries `traffic` for these very sources
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $root/gpurun_out/pmc_fetch_$tag -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --streams 0 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $root/gpurun_out/pmc_write_$tag -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --streams 0 > /dev/null 2>&1
cd $root
python3 tools/pmc_summary.py gpurun_out/pmc_fetch_$tag gpurun_out/pmc_write_$tag gpurun_out/pmc_fetch_write_per_kernel_$tag.json | head -5
cp gpurun_out/pmc_fetch_write_per_kernel_$tag.json profiles/r02/pmc_fetch_write_per_kernel.json      # (this box's copy; the caller commits the one merged back)
python bench.py 2>gpurun_out/bench_$tag.err > gpurun_out/bench_$tag.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -- python3 $root/bench.py $args > $root/gpurun_out/prof_$tag.json 2>/dev/null
cd $root
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/bench_${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:7]:
    name = r["Name"].replace("void ", "").replace("flimo::", "")[:40]
    print(f'{name:40s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us  min {float(r["MinNs"])/1e3:7.2f}  max {float(r["MaxNs"])/1e3:7.2f}  {r["Percentage"]:>6s}%')
PY
head -c 1500 gpurun_out/bench_$tag.json; echo
# k-NN stage on its own: the same command with the pass split into dispatches (A/B switch)
cd /tmp
rm -rf $root/gpurun_out/prof_${tag}_nofuse
FLIMO_FUSE=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_${tag}_nofuse -- python3 $root/bench.py $args > /dev/null 2>&1
cd $root
f=$(find gpurun_out/prof_${tag}_nofuse -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/bench_${tag}_nofuse_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:6]:
    name = r["Name"].replace("void ", "").replace("flimo::", "")[:40]
    print(f'FUSE=0  {name:40s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us  min {float(r["MinNs"])/1e3:7.2f}  max {float(r["MaxNs"])/1e3:7.2f}')
PY
