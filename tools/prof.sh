#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run via gpurun), for ONE workload:
#   1. --kernel-trace --stats           -> gpurun_out/prof_<tag>/stats
#   2. --kernel-trace --pmc FETCH_SIZE  -> gpurun_out/prof_<tag>/fetch   (separate pass, MI355X guide: TCC has 4 slots)
#   3. --kernel-trace --pmc WRITE_SIZE  -> gpurun_out/prof_<tag>/write
# then tools/traffic.py turns the three into gpurun_out/prof_<tag>/summary.json.
#   tools/prof.sh TAG [bench.py arguments ...]
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="bench.py --steps 5 --warmup 1 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/bench_stats.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/bench_write.json 2> $OUT/write.err
python3 tools/traffic.py $OUT > $OUT/summary.json
cat $OUT/summary.json
