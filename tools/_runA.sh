OUT=gpurun_out/final
mkdir -p $OUT
timeout -k 10 300 python3 tools/part_profile.py --world 8 > $OUT/part_profile_emu8_lpt.log 2>&1
timeout -k 10 300 python3 tools/part_profile.py --world 8 --plan-crit 0.3 > $OUT/part_profile_emu8_chain_aware.log 2>&1
timeout -k 10 400 python3 tools/part_profile.py --world 8 --workload cfg3_netflix --scaling strong > $OUT/part_profile_netflix_strong8_lpt.log 2>&1
timeout -k 10 400 python3 tools/part_profile.py --world 8 --workload cfg3_netflix --scaling strong --plan-crit 0.3 > $OUT/part_profile_netflix_strong8_chain_aware.log 2>&1
timeout -k 10 300 python3 tools/sched_trace.py cfg3_netflix 1.0 device > $OUT/sched_trace_netflix.log 2>&1
timeout -k 10 300 python3 tools/phase_profile.py cfg2_uniform 1.0 > $OUT/phase_profile_uniform.log 2>&1
timeout -k 10 400 python3 bench.py --no-cpu-baseline --workload cfg3_netflix --emulate-world 8 > $OUT/bench_netflix_strong_emu8.json 2>> $OUT/bench.err
grep -h "sum\|plan" $OUT/part_profile_*.log | cut -c1-200
