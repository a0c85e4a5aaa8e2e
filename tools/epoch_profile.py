#!/usr/bin/env python3
"""Diagnostic: where a persistent epoch spends its cycles, per phase (mean over workgroups,
and the workgroup with the largest total)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mfsgd_amd
from mfsgd_amd import synth

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_ml20m"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
w = synth.workload(name, scale)
m = mfsgd_amd.MatrixFactorizationSGD(w["U"], w["I"], w["k"], 0.01, 0.05, 3, host_threads=16)
m.set_ratings(w["u"], w["i"], w["r"]); m.init_factors(); m.fit(2, rmse=False)
info = m.schedule_info()
p = m.debug_epoch_profile().astype(np.float64)
names = ["drain+issue", "wait tile", "barrier", "gather tile", "ratings", "publish tile", "store own"]
tot = p.sum(axis=1)
print(name, scale, "B", info["blocks"], "rounds", info["rounds"], "workgroups", p.shape[0])
print("cycles per round, mean over workgroups: " + "  ".join(f"{n} {v:.0f}" for n, v in zip(names, p.mean(axis=0) / info["rounds"])))
print("  total per round %.0f cycles = %.2f us at 2.29 GHz" % (tot.mean() / info["rounds"], tot.mean() / info["rounds"] / 2290))
