#!/usr/bin/env python3
"""Builder-run extra (not collected by pytest): the bench workload at full size for MANY epochs -- every epoch a
replayed launch of the persistent kernel (start barrier, flag hand-off, mailbox hand-off of the giants' tiles with
the launch generation in their tags) -- against the multithreaded oracle on the same schedule, factors compared
bit for bit every ten epochs.  What the suite checks for three epochs, for forty.

    python tests/gpu_soak_extra.py [EPOCHS [WORKLOAD [SCALE]]]      (log committed under profiles/)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import mfsgd_amd as mf  # noqa: E402
from tests.oracle_bind import Oracle  # noqa: E402

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
name = sys.argv[2] if len(sys.argv) > 2 else "cfg2_ml20m"
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
LR, LAM, SEED = 0.01, 0.05, 3
orc = Oracle()
w = mf.synth.workload(name, scale)
t0 = time.time()
bad = 0
with mf.MatrixFactorizationSGD(w["U"], w["I"], w["k"], LR, LAM, SEED, host_threads=16) as m:
    m.set_ratings(w["u"], w["i"], w["r"])
    m.init_factors()
    order, cell_ptr = m.order()
    info = m.schedule_info()
    lone = int((m.debug_schedule()[0][:, 5] & 1).sum())
    print(f"{name} x{scale}: {info['nnz']} ratings, k={w['k']}, B={info['blocks']} W={info['waves']}, cells marked for the mailbox: {lone}", flush=True)
    P, Q = orc.init_factors(w["U"], w["I"], w["k"], SEED)
    done = 0
    while done < epochs:
        n = min(10, epochs - done)
        m.fit(n, rmse=False)
        for _ in range(n):
            orc.sgd_epoch_mt(P, Q, w["u"], w["i"], w["r"], order, cell_ptr, info["rounds"], info["blocks"], LR, LAM, 16)
        done += n
        Pg, Qg = m.get_factors()
        same = np.array_equal(Pg, P) and np.array_equal(Qg, Q)
        bad += not same
        print(f"  after {done} epochs: factors {'bit-exact' if same else 'DIFFER'}; rmse {m.rmse():.9f} (oracle {orc.rmse(P, Q, w['u'], w['i'], w['r']):.9f}); {time.time() - t0:.0f} s", flush=True)
    print("counters:", m.debug_counters())
print("done, mismatching checkpoints:", bad)
sys.exit(1 if bad else 0)
