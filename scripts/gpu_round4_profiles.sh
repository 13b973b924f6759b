#!/bin/bash
# Round 4: every profile the docs quote, from ONE build: bench kernel statistics, HBM traffic of the
# grad / GAE kernels, SQ counters of the grad kernels, SAC kernel statistics + counters.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
bash scripts/gpu_profile.sh 2>&1 | tail -4
cp gpurun_out/bench_prof.log gpurun_out/r04_bench_prof.log
bash scripts/gpu_pmc_traffic.sh 2>&1 | tail -12
bash scripts/gpu_pmc.sh 2>&1 | tail -45 > gpurun_out/r04_pmc_grad.txt; tail -45 gpurun_out/r04_pmc_grad.txt
bash scripts/gpu_profile_sac.sh graph 2>&1 | tail -24 > gpurun_out/r04_sac_profile.txt; tail -8 gpurun_out/r04_sac_profile.txt
bash scripts/gpu_pmc_sac.sh 2>&1 | tail -60 > gpurun_out/r04_pmc_sac.txt; tail -30 gpurun_out/r04_pmc_sac.txt
