#!/bin/bash
# A/B of two builds of the library on the bench workloads:  tools/ab_lib.sh OUTDIR TAG [LIBRARY]
# (LIBRARY: a libmfsgd.so built with another -D switch; default: the in-tree build)
OUT=$1; TAG=$2; LIB=$3
mkdir -p $OUT
[ -n "$LIB" ] && export MFSGD_LIBRARY=$LIB
run() { name=$1; shift; timeout -k 10 300 python3 bench.py --no-cpu-baseline "$@" > $OUT/${TAG}_$name.json 2> $OUT/${TAG}_$name.err || exit 1; python3 - $OUT/${TAG}_$name.json $TAG $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], sys.argv[3], "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"])
PY
}
run cfg2_ml20m --steps 20 --warmup 3 &&
run cfg2_uniform --workload cfg2_uniform --steps 20 --warmup 3 &&
run cfg3_netflix_0.2 --workload cfg3_netflix --scale 0.2 --steps 10 --warmup 2 &&
run cfg4_powerlaw_0.02 --workload cfg4_powerlaw --scale 0.02 --steps 10 --warmup 2 &&
run emu8 --emulate-world 8 --steps 10 --warmup 2
