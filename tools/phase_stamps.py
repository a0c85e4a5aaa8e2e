#!/usr/bin/env python3
"""Diagnostic: per-phase shader-clock shares of the sgd cell kernel (gather /
rating steps / scatter) on the bench workload.  Not part of the product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mfsgd_amd
from mfsgd_amd import synth

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_ml20m"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
w = synth.workload(name, scale)
m = mfsgd_amd.MatrixFactorizationSGD(w["U"], w["I"], w["k"], 0.01, 0.05, 3, host_threads=16)
m.set_ratings(w["u"], w["i"], w["r"]); m.init_factors()
m.fit(1, rmse=False)
info = m.schedule_info()
print({k: info[k] for k in ("blocks", "waves", "slots", "lds_bytes", "total_steps", "sum_round_steps", "max_cell_steps")})
acc = []
for rd in range(0, info["blocks"], max(1, info["blocks"] // 16)):
    s = m.debug_round_stamps(rd).astype(np.int64)
    t0 = s[:, 0].min()
    g, st, sc = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2]
    tot = s[:, 3].max() - t0
    slow = np.argmax(s[:, 3])
    acc.append((tot, g.mean(), st.mean(), sc.mean(), g[slow], st[slow], sc[slow], (s[:, 0] - t0).max()))
a = np.array(acc, float)
print("clock ticks (100 MHz s_memtime? see guide: shader clock) per round, mean over sampled rounds:")
print("  round span %.0f | mean WG: gather %.0f steps %.0f scatter %.0f | slowest WG: gather %.0f steps %.0f scatter %.0f | last WG start +%.0f"
      % tuple(a.mean(axis=0)))
