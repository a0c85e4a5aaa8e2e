set -o pipefail
mkdir -p gpurun_out/r03a
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "recovers or cpp_host or planned or dsgd or bench" > gpurun_out/r03a/tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03a/tests.log; tail -3 gpurun_out/r03a/tests.log
for n in 2 4 8; do
  for crit in 0 0.3; do
    MFSGD_PLAN_CRIT=$crit timeout -k 10 300 python bench.py --no-cpu-baseline --emulate-world $n > gpurun_out/r03a/emu${n}_crit${crit}.json 2>> gpurun_out/r03a/bench.err || echo "emu $n $crit failed"
  done
done
MFSGD_PLAN_CRIT=0.15 timeout -k 10 300 python bench.py --no-cpu-baseline --emulate-world 8 > gpurun_out/r03a/emu8_crit0.15.json 2>> gpurun_out/r03a/bench.err
timeout -k 10 300 python bench.py --no-cpu-baseline --emulate-world 8 --parts-per-rank 2 > gpurun_out/r03a/emu8_ppr2.json 2>> gpurun_out/r03a/bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03a/*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], '%.3e'%d['value'], '%.3f ms'%d['ms_per_step'], 'frac %.3f'%d['roofline']['frac'], d['config']['plan'], d['config']['blocks'], d['config']['waves'])
    except Exception as e: print(f, 'ERR', e)
PY
