"""Where mfsgd_set_ratings spends its time (MFSGD_SCHED_TRACE=1 prints the scheduler's phases).

    MFSGD_SCHED_TRACE=1 python tools/ingest_trace.py [WORKLOAD] [SCALE]
"""
import sys
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import mfsgd_amd as mf  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_ml20m"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
w = mf.synth.workload(name, scale)
for rep in range(2):
    with mf.MatrixFactorizationSGD(w["U"], w["I"], w["k"], 0.01, 0.05, 3, host_threads=16) as m:
        t0 = time.perf_counter()
        m.set_ratings(w["u"], w["i"], w["r"])
        t1 = time.perf_counter()
        m.init_factors()
        t2 = time.perf_counter()
        m.fit(1, rmse=False)
        t3 = time.perf_counter()
        print(f"rep {rep}: set_ratings {1e3 * (t1 - t0):.0f} ms, init_factors {1e3 * (t2 - t1):.0f} ms, "
              f"first epoch (uploads + probe + graph) {1e3 * (t3 - t2):.0f} ms", flush=True)
