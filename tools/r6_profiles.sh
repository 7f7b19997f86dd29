#!/bin/bash
# round-6 evidence (GPU box): default bench line, rocprofv3 kernel stats of the same command, PMC traffic passes for the headline
# workload and for the HBM regime (256k x 20M).  Results are copied into profiles/r06/ (the caller commits the copy merged back
# through gpurun_out/profiles_r06/).
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out/profiles_r06 profiles/r06
out=$root/gpurun_out/profiles_r06
# --streams 0: without the informational concurrent-streams leg, whose overlapped launches would enter the per-kernel averages
args="--steps 100 --warmup 5 --no-cpu-baseline --no-end-to-end --no-hbm-regime --no-crowded --streams 0"
cd /tmp && export TMPDIR=/tmp
rm -rf $out/pmc_fetch $out/pmc_write $out/prof $out/prof_nopipeline $out/prof_nofuse $out/pmc_fetch_hbm $out/pmc_write_hbm $out/prof_hbm
# ---- PMC passes first (counters in their own runs), so that the bench line below carries `traffic` for these very sources ----
# (FLIMO_PIPELINE=0 for the counter passes and for one of the kernel summaries: a pass queued ahead of the host's algebra is the same
#  kernel under its chained name, its duration includes the wait for its pose and one per scan leaves without working -- per-kernel
#  averages of traffic and time are taken from launches that start with their pose)
export FLIMO_PIPELINE=0
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --no-hbm-regime --no-crowded --streams 0 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --no-hbm-regime --no-crowded --streams 0 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch_hbm -- python3 $root/bench.py --hbm-regime-only --no-cpu-baseline --hbm-steps 6 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write_hbm -- python3 $root/bench.py --hbm-regime-only --no-cpu-baseline --hbm-steps 6 > /dev/null 2>&1
unset FLIMO_PIPELINE
cd $root
python3 tools/pmc_summary.py $out/pmc_fetch $out/pmc_write $out/pmc_fetch_write_per_kernel.json | head -6
python3 tools/pmc_summary.py $out/pmc_fetch_hbm $out/pmc_write_hbm $out/pmc_hbm_regime.json | head -6
cp $out/pmc_fetch_write_per_kernel.json $out/pmc_hbm_regime.json profiles/r06/      # this box's copy: the bench below reads it
# ---- the default bench line (what the driver runs) with more steps, and the same command under rocprofv3 --stats ----
python bench.py --steps 100 --warmup 5 2>$out/bench_r06.err > $out/bench_r06.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 $root/bench.py $args > $out/bench_r06_under_rocprof.json 2>/dev/null
FLIMO_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_nopipeline -- python3 $root/bench.py $args > /dev/null 2>&1
FLIMO_FUSE=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_nofuse -- python3 $root/bench.py $args > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_hbm -- python3 $root/bench.py --hbm-regime-only --no-cpu-baseline > $out/bench_r06_hbm_under_rocprof.json 2>/dev/null
cd $root
for t in prof prof_nopipeline prof_nofuse prof_hbm; do
  f=$(find $out/$t -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $out/bench_r06_${t}_kernel_stats.csv
done
python3 - "$out" <<'PY'
import csv, sys, glob, os
for t in ("prof", "prof_nopipeline", "prof_nofuse", "prof_hbm"):
    f = os.path.join(sys.argv[1], "bench_r06_%s_kernel_stats.csv" % t)
    if not os.path.exists(f): continue
    print("==", t)
    for r in list(csv.DictReader(open(f)))[:7]:
        name = r["Name"].replace("void ", "").replace("flimo::", "")[:44]
        print(f'{name:44s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us  min {float(r["MinNs"])/1e3:7.2f}  max {float(r["MaxNs"])/1e3:7.2f}  {r["Percentage"]:>6s}%')
PY
head -c 600 $out/bench_r06.json; echo
rm -rf $out/pmc_fetch $out/pmc_write $out/prof $out/prof_nopipeline $out/prof_nofuse $out/pmc_fetch_hbm $out/pmc_write_hbm $out/prof_hbm
ls -la $out
