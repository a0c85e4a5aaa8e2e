#!/usr/bin/env python3
"""Builder-run extra (not collected by pytest: too long for the suite): BASELINE configs[4]'s shape
-- power-law popularity, k = 256, DSGD x 8 -- at ONE EIGHTH of its full size, i.e. what the 8-GPU job
gives ONE GPU (125 M ratings, 1.25 M users) but with all eight devices virtual on this one GPU:
the global set cut by mfsgd_dsgd_plan, 8 handles, real kernels, blocks rotated by pointer, one epoch,
bit for bit against the sequential DSGD definition run by the oracle (multithreaded inside each
(device, partition) block).  The full 1 B-rating set does not fit a single host's test budget
(its generation alone takes half an hour).

    python tests/gpu_large_extra.py [SCALE]      (log committed under profiles/)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import mfsgd_amd as mf  # noqa: E402
from tests.oracle_bind import Oracle  # noqa: E402
from tests.test_gpu_parity import _virtual_dsgd  # noqa: E402

import threading  # noqa: E402


def _heartbeat():  # the GPU pool takes seven silent minutes for a hang
    t = time.time()
    while True:
        time.sleep(60)
        print(f"  ... {time.time() - t:.0f} s", flush=True)


threading.Thread(target=_heartbeat, daemon=True).start()
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.125
t0 = time.time()
w = mf.synth.workload("cfg4_powerlaw", scale)
print(f"cfg4_powerlaw x {scale}: {w['U']} x {w['I']}, {w['nnz']} ratings, k = {w['k']} (generated in {time.time() - t0:.0f} s)", flush=True)
t0 = time.time()
rm, ub, ip, infos = _virtual_dsgd(mf, Oracle(), w, 8, 1, mt_threads=16, host_threads=16)
du = np.bincount(w["u"], minlength=w["U"])
shard = [int(du[ub[g]:ub[g + 1]].sum()) for g in range(8)]
print(f"DSGD x 8 (virtual devices), one epoch: factors and SSE bit-exact against the sequential definition; rmse {rm.tolist()}")
print(f"  ratings per device {shard}; partitions per device "
      f"{[[infos[g][p]['nnz'] for p in range(8)] for g in range(2)]} ...; device-packed schedules: "
      f"{sum(x['device_ingest'] == 2 for row in infos for x in row)} of 64; chunked cells "
      f"{sum(x['split_cells'] for row in infos for x in row)}; {time.time() - t0:.0f} s")
