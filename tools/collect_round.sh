#!/bin/bash
# Collects one round's evidence on the GPU box (run via gpurun, in two halves so that each stays inside gpurun's limit):
#   tools/collect_round.sh prof    -> rocprofv3 kernel stats + PMC traffic for k = 64 / 128 / 256 (tools/prof.sh)
#   tools/collect_round.sh bench   -> the bench lines DESIGN.md section 5 quotes, phase profiles, the loop microbenchmark
#   tools/collect_round.sh sched   -> set_ratings traces of the three large shapes
# Everything lands in gpurun_out/final/; tools/collect_round.sh copy TAG (run locally) files it under profiles/TAG_*.
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/final
mkdir -p $OUT
case "$1" in
prof)
    timeout -k 10 300 tools/prof.sh k64 > $OUT/prof_k64.log 2>&1
    timeout -k 10 300 tools/prof.sh k128 --workload cfg3_netflix --scale 0.2 > $OUT/prof_k128.log 2>&1
    timeout -k 10 400 tools/prof.sh k256 --workload cfg4_powerlaw --scale 0.02 > $OUT/prof_k256.log 2>&1
    ;;
bench)
    timeout -k 10 200 python3 bench.py > $OUT/bench_default_with_cpu_baseline.json 2> $OUT/bench_default.err
    for w in cfg2_uniform cfg2_zipf cfg2_r1; do
        timeout -k 10 200 python3 bench.py --no-cpu-baseline --workload $w > $OUT/bench_$w.json 2>> $OUT/bench.err
    done
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --workload cfg3_netflix --scale 0.2 > $OUT/bench_cfg3_netflix_0.2.json 2>> $OUT/bench.err
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --workload cfg4_powerlaw --scale 0.02 > $OUT/bench_cfg4_powerlaw_0.02.json 2>> $OUT/bench.err
    for n in 2 4 8; do
        timeout -k 10 200 python3 bench.py --no-cpu-baseline --emulate-world $n > $OUT/bench_emu$n.json 2>> $OUT/bench.err
    done
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --workload cfg3_netflix > $OUT/bench_cfg3_netflix_1.0.json 2>> $OUT/bench.err
    timeout -k 10 120 python3 tools/phase_profile.py cfg2_ml20m 1.0 > $OUT/phase_profile_solo.log 2>&1
    timeout -k 10 120 python3 tools/phase_profile.py cfg2_ml20m 1.0 0 0 16 > $OUT/phase_profile_nosolo.log 2>&1
    MFSGD_EMU=8 timeout -k 10 120 python3 tools/phase_profile.py cfg2_ml20m 1.0 > $OUT/phase_profile_emu8_partition.log 2>&1
    for L in 16 32 64; do timeout -k 5 60 tools/bin/ub3_$L; done > $OUT/ubench3.log 2>&1
    ;;
sched)
    # set_ratings phase by phase (MFSGD_SCHED_TRACE), second call of a process: the numbers of DESIGN.md section 8
    timeout -k 10 200 python3 tools/sched_trace.py cfg2_ml20m 1.0 host > $OUT/sched_trace_cfg2_ml20m.log 2>&1
    timeout -k 10 200 python3 tools/sched_trace.py cfg3_netflix 1.0 device > $OUT/sched_trace_cfg3_netflix_1.0.log 2>&1
    timeout -k 10 200 python3 tools/sched_trace.py cfg4_powerlaw 0.125 device > $OUT/sched_trace_cfg4_x0.125.log 2>&1
    ;;
copy)
    TAG=$2
    for k in k64 k128 k256; do
        d=gpurun_out/prof_$k
        cp $d/summary.json profiles/${TAG}_traffic_$k.json
        cp $d/bench_stats.json profiles/${TAG}_${k}_bench_under_rocprof.json
        cp "$(ls -t $d/stats/*/*kernel_stats.csv | head -1)" profiles/${TAG}_${k}_kernel_stats.csv
        cp "$(ls -t $d/fetch/*/*counter_collection.csv | head -1)" profiles/${TAG}_${k}_pmc_FETCH_SIZE.csv
        cp "$(ls -t $d/write/*/*counter_collection.csv | head -1)" profiles/${TAG}_${k}_pmc_WRITE_SIZE.csv
    done
    for f in $OUT/bench_*.json $OUT/phase_profile_*.log $OUT/ubench3.log; do cp $f profiles/${TAG}_$(basename $f); done
    for f in $OUT/sched_trace_*.log; do if [ -f $f ]; then grep -v amdgpu.ids $f > profiles/${TAG}_$(basename $f); fi; done
    ;;
*) echo "usage: $0 prof|bench|sched|copy TAG"; exit 2 ;;
esac
