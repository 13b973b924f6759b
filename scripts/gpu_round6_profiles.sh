#!/bin/bash
# Round 6: the numbers the docs and the bench line quote — the GPU test tier, the default bench line, rocprofv3
# kernel statistics of the bench command, HBM traffic of the grad / GAE kernels (separate --pmc passes), SQ
# counters of the grad kernels, kernel statistics + cache counters of the SAC update, the cfg-5 and cfg-4 lines.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r06_gpu_tests.log
tail -4 gpurun_out/r06_gpu_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench.err
tail -c 400 gpurun_out/r06_bench_line.json; echo; tail -2 gpurun_out/r06_bench.err
bash scripts/gpu_profile.sh 2>&1 | tail -6
cp gpurun_out/prof_kernel_stats.csv gpurun_out/r06_bench_kernel_stats.csv
tail -1 gpurun_out/bench_prof.log > gpurun_out/r06_bench_line_profiled.json
bash scripts/gpu_pmc_traffic.sh 2>&1 | tail -12
for c in FETCH_SIZE WRITE_SIZE; do cp gpurun_out/pmc_traffic_$c.csv gpurun_out/r06_pmc_traffic_$c.csv; done
bash scripts/gpu_pmc.sh 4 2>&1 | tail -45 > gpurun_out/r06_pmc_grad.txt; tail -30 gpurun_out/r06_pmc_grad.txt
for p in a b c; do cp gpurun_out/pmc_${p}_w4.csv gpurun_out/r06_pmc_${p}_grad16.csv 2>/dev/null; done
bash scripts/gpu_profile_sac.sh graph 2>&1 | tail -24 > gpurun_out/r06_sac_profile.txt; tail -24 gpurun_out/r06_sac_profile.txt
cp gpurun_out/sac_kernel_stats.csv gpurun_out/r06_sac_kernel_stats.csv
bash scripts/gpu_pmc_sac.sh 2>&1 | tail -60 > gpurun_out/r06_pmc_sac.txt
for p in a b c; do cp gpurun_out/pmc_sac_$p.csv gpurun_out/r06_pmc_sac_$p.csv 2>/dev/null; done
timeout 300 python bench.py --workload cfg5 --steps 3 --warmup 1 > gpurun_out/r06_bench_line_cfg5.json 2>> gpurun_out/r06_bench.err
tail -c 300 gpurun_out/r06_bench_line_cfg5.json; echo
timeout 300 python bench.py --workload cfg4 --steps 20 --warmup 3 > gpurun_out/r06_bench_line_cfg4.json 2>> gpurun_out/r06_bench.err
tail -c 300 gpurun_out/r06_bench_line_cfg4.json; echo
timeout 400 python bench.py --gpus 2 --workload cfg4 --steps 10 --warmup 2 > gpurun_out/r06_bench_gpus2_cfg4_shared_device.json 2>> gpurun_out/r06_bench.err
tail -c 600 gpurun_out/r06_bench_gpus2_cfg4_shared_device.json; echo
