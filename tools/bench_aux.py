"""Timings of the paths around training: predict (batched), recommend (top-N), RMSE pass,
set_ratings (schedule build).  Wall-clock, through the C-ABI (host arrays in and out, so PCIe
copies are included); run it under `rocprofv3 --kernel-trace --stats` for the kernel times.

    python tools/bench_aux.py [WORKLOAD] [SCALE]
"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import mfsgd_amd as mf  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_ml20m"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
w = mf.synth.workload(name, scale)
k = w["k"]
with mf.MatrixFactorizationSGD(w["U"], w["I"], k, 0.01, 0.05, 3, host_threads=16) as m:
    t0 = time.perf_counter()
    m.set_ratings(w["u"], w["i"], w["r"])
    t_set = time.perf_counter() - t0
    m.init_factors()
    m.fit(1, rmse=False)
    m.rmse()
    t0 = time.perf_counter()
    for _ in range(5):
        m.rmse()
    t_rmse = (time.perf_counter() - t0) / 5
    n = w["nnz"]
    m.predict(w["u"][:1000], w["i"][:1000])
    t0 = time.perf_counter()
    out = m.predict(w["u"], w["i"])
    t_pred = time.perf_counter() - t0
    users = np.arange(0, w["U"], max(1, w["U"] // 4096), dtype=np.int32)[:4096]
    m.recommend(users[:16], 10)
    t0 = time.perf_counter()
    items, scores = m.recommend(users, 10)
    t_rec = time.perf_counter() - t0
print(f"{name} x{scale}: {n} ratings, {w['U']} x {w['I']}, k = {k}")
print(f"  set_ratings (schedule build + ingest)  {t_set * 1e3:9.1f} ms")
print(f"  rmse pass                              {t_rmse * 1e3:9.3f} ms  = {n / t_rmse / 1e9:.2f} G ratings/s")
print(f"  predict, {n} pairs (host in/out)   {t_pred * 1e3:9.1f} ms  = {n / t_pred / 1e9:.3f} G pairs/s")
print(f"  recommend top-10, {users.size} users x {w['I']} items {t_rec * 1e3:9.1f} ms  = "
      f"{users.size * w['I'] / t_rec / 1e9:.2f} G scores/s")
