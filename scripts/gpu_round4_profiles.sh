#!/bin/bash
# Round 4, final build: the profiles the docs quote — bench kernel statistics + trace, HBM traffic of the
# grad / GAE kernels, SQ counters and phase stamps of the grad kernels.  (The off-policy kernels did not
# change after profiles/r04_sac_*: scripts/gpu_profile_sac.sh, gpu_pmc_sac.sh.)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
bash scripts/gpu_profile.sh 2>&1 | tail -4
cp gpurun_out/bench_prof.log gpurun_out/r04_bench_prof.log
bash scripts/gpu_pmc_traffic.sh 2>&1 | tail -12
bash scripts/gpu_pmc.sh 4 2>&1 | tail -45 > gpurun_out/r04_pmc_grad.txt; tail -45 gpurun_out/r04_pmc_grad.txt
timeout 300 python scripts/grad_phases.py 2>&1 | tail -14 | tee gpurun_out/r04_grad_phases.txt
