mkdir -p gpurun_out/r03d
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_k or blocks_and_waves or fuzz or cfg or chunked or hot_item or run_loops or solo or lone_tile or edge or device_packer" > gpurun_out/r03d/tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03d/tests.log; tail -4 gpurun_out/r03d/tests.log
for wl in cfg2_ml20m cfg2_uniform; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --workload $wl > gpurun_out/r03d/bench_$wl.json 2>> gpurun_out/r03d/bench.err
done
timeout -k 10 300 python bench.py --no-cpu-baseline --workload cfg3_netflix --scale 0.2 > gpurun_out/r03d/bench_netflix02.json 2>> gpurun_out/r03d/bench.err
timeout -k 10 300 python bench.py --no-cpu-baseline --workload cfg4_powerlaw --scale 0.02 > gpurun_out/r03d/bench_powerlaw002.json 2>> gpurun_out/r03d/bench.err
timeout -k 10 300 python bench.py --no-cpu-baseline --emulate-world 8 > gpurun_out/r03d/emu8.json 2>> gpurun_out/r03d/bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03d/*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], '%.3e'%d['value'], '%.3f ms'%d['ms_per_step'], 'frac %.3f'%d['roofline']['frac'], d.get('emulation',{}).get('sum_ms'), d['rmse_after'])
    except Exception as e: print(f, 'ERR', e)
PY
