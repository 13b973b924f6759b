cd scripts/ubench
for rep in 1 2; do
for push in 0 2; do
  timeout 60 ./pingpong.bin 18 4352 1536 1 0 $push 2>&1 | grep "median\|tail\|stale"
done
done > /root/repo/gpurun_out/r05_pushpong_tail.txt 2>&1
