"""Diagnostic: two independent handles on the same ratings must give bit-identical factors.
    python tools/determinism.py [SCALE] [FLAGS ...]          (workload: MFSGD_WORKLOAD, default cfg2_ml20m)"""
import os
import sys
import numpy as np
sys.path.insert(0, '.')
import mfsgd_amd as mf
w = mf.synth.workload(os.environ.get("MFSGD_WORKLOAD", "cfg2_ml20m"), float(sys.argv[1]) if len(sys.argv) > 1 else 1.0)
for flags in [int(x) for x in sys.argv[2:]] or [0]:
    facs = []
    for rep in range(3):
        with mf.MatrixFactorizationSGD(w["U"], w["I"], w["k"], 0.01, 0.05, 3, host_threads=16, flags=flags) as m:
            rm = m.train(w["u"], w["i"], w["r"], 3)
            facs.append(m.get_factors() + (rm,))
            print("   rmse", rm.tolist(), m.debug_counters(), m.schedule_info()["device_ingest"], flush=True)
    for rep in (1, 2):
        dp = (facs[rep][0] != facs[0][0]).any(axis=1).sum()
        dq = (facs[rep][1] != facs[0][1]).any(axis=1).sum()
        print(f"flags {flags}: run {rep} vs 0: P rows differing {dp}, Q rows differing {dq}, rmse equal {np.array_equal(facs[rep][2], facs[0][2])}", flush=True)
