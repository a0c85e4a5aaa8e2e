#!/bin/bash
# Collects the rocprofv3 evidence for bench.py (run on the GPU box via gpurun):
#   1. --kernel-trace --stats           -> gpurun_out/prof_stats
#   2. --pmc FETCH_SIZE                 -> gpurun_out/prof_fetch   (separate pass, MI355X guide)
#   3. --pmc WRITE_SIZE                 -> gpurun_out/prof_write
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="bench.py --steps 5 --warmup 1 --no-cpu-baseline $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 $ARGS > gpurun_out/prof_stats.json 2> gpurun_out/prof_stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -- python3 $ARGS > gpurun_out/prof_fetch.json 2> gpurun_out/prof_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -- python3 $ARGS > gpurun_out/prof_write.json 2> gpurun_out/prof_write.err
find gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write -name '*.csv' | head -20
