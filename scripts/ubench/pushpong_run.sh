cd scripts/ubench
for rep in 1 2; do
for push in 0 1 2 3; do
 for cfg in "18 7168 2048 1" "18 9216 2048 1"; do
  timeout 60 ./pingpong.bin $cfg 0 $push 2>&1 | grep "median\|stale\|LargeBar"
 done
done
done > /root/repo/gpurun_out/r05_pushpong3.txt 2>&1
