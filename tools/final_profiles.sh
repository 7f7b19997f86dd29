root=${GRAFT_REPO_ROOT:-$(pwd)}
python bench.py 2>/dev/null > gpurun_out/bench_final.json
tools/prof_bench.sh final
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/pmc_fetch $root/gpurun_out/pmc_write
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $root/gpurun_out/pmc_fetch -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $root/gpurun_out/pmc_write -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $root
python3 tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_fetch_write_per_kernel.json
python3 -c "
import json; d=json.load(open('gpurun_out/pmc_fetch_write_per_kernel.json')); print(d.get('knn5_kernel_traffic_bytes_per_launch'))"
cp $(find gpurun_out/prof_final -name '*kernel_stats.csv' | head -1) gpurun_out/bench_final_kernel_stats.csv 2>/dev/null
head -c 600 gpurun_out/bench_final.json
