cd ${GRAFT_REPO_ROOT:-/root/repo}
for t in 2 3; do
  echo "== transport $t"
  TONIC_AMD_COLLECTOR_STAMPS=1 TONIC_AMD_COLLECTOR_TRANSPORT=$t timeout 300 python scripts/host_loop_probe.py 3000 2>&1 | grep -v amdgpu.ids | tail -8
done > gpurun_out/r05_push_stamps.txt 2>&1
