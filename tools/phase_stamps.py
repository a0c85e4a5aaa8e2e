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
from mfsgd_amd import _lib
m = mfsgd_amd.MatrixFactorizationSGD(w["U"], w["I"], w["k"], 0.01, 0.05, 3, host_threads=16, flags=_lib.FLAG_ROUND_LAUNCH)
m.set_ratings(w["u"], w["i"], w["r"]); m.init_factors()
m.fit(1, rmse=False)
info = m.schedule_info()
print({k: info[k] for k in ("blocks", "waves", "slots", "lds_bytes", "total_steps", "sum_round_steps", "max_cell_steps")})
acc = []
import time
for _ in range(3): m.fit(1, rmse=False)
for rd in range(0, info["blocks"], max(1, info["blocks"] // 16)):
    s = m.debug_round_stamps(rd).astype(np.int64)
    g, st, sc = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2]
    dur_cyc = s[:, 3] - s[:, 0]
    dur_real = (s[:, 5] - s[:, 4]) * 10.0  # ns
    clk = dur_cyc / np.maximum(dur_real, 1)  # GHz
    span_ns = (s[:, 5].max() - s[:, 4].min()) * 10.0
    start_spread_ns = (s[:, 4].max() - s[:, 4].min()) * 10.0
    slow = np.argmax(s[:, 5])
    acc.append((g.mean(), st.mean(), sc.mean(), g[slow], st[slow], sc[slow], np.median(clk), span_ns, start_spread_ns, dur_real.mean(), dur_real.max()))
    lt = m.last_loop_timers.astype(np.float64)
    if rd == 0:
        print("slowest WG", slow, "loop timers [wave, sub-round, (gen cyc, run cyc, gen steps, run steps)]:")
        print(lt[slow].astype(np.int64))
    gs, rs = lt[..., 2].sum(), lt[..., 3].sum()
    tot_g = (tot_g[0] + lt[..., 0].sum(), tot_g[1] + gs) if "tot_g" in dir() else (lt[..., 0].sum(), gs)
    tot_r = (tot_r[0] + lt[..., 1].sum(), tot_r[1] + rs) if "tot_r" in dir() else (lt[..., 1].sum(), rs)
a = np.array(acc, float).mean(axis=0)
print("mean WG cycles: gather %.0f steps %.0f scatter %.0f | last-finishing WG: gather %.0f steps %.0f scatter %.0f" % tuple(a[:6]))
print("cycles per general step %.1f (n=%d) | per run step %.1f (n=%d)" % (tot_g[0] / max(tot_g[1], 1), tot_g[1], tot_r[0] / max(tot_r[1], 1), tot_r[1]))
print("shader clock %.2f GHz | first WG start -> last WG end %.2f us | WG start spread %.2f us | WG lifetime mean %.2f us max %.2f us" % (a[6], a[7]/1e3, a[8]/1e3, a[9]/1e3, a[10]/1e3))
