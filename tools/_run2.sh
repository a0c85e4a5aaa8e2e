mkdir -p gpurun_out/r03b
for crit in 0 0.3; do
 MFSGD_PLAN_CRIT=$crit timeout -k 10 200 python tools/part_profile.py --world 8 > gpurun_out/r03b/pp8_crit$crit.log 2>&1
done
for W in 2 4; do for B in 64 96 128; do
 timeout -k 10 200 python tools/part_profile.py --world 8 --blocks $B --waves $W > gpurun_out/r03b/pp8_B${B}_W$W.log 2>&1
done; done
timeout -k 10 200 python tools/part_profile.py --world 8 --waves 4 > gpurun_out/r03b/pp8_W4.log 2>&1
timeout -k 10 200 python tools/part_profile.py --world 8 --round-launch > gpurun_out/r03b/pp8_rl.log 2>&1
grep -h "sum\|plan" gpurun_out/r03b/*.log
