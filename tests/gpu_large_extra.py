#!/usr/bin/env python3
"""Builder-run extra (not collected by pytest: too long for the suite): BASELINE configs[4]'s shape
-- power-law popularity, k = 256, DSGD x 8 -- at ONE EIGHTH of its full size, i.e. what the 8-GPU job
gives ONE GPU (125 M ratings, 1.25 M users) but with all eight devices virtual on this one GPU:
the global set cut by mfsgd_dsgd_plan, 8 handles, real kernels, blocks rotated by pointer, one epoch,
bit for bit against the sequential DSGD definition run by the oracle (multithreaded inside each
(device, partition) block).  The full 1 B-rating set does not fit a single host's test budget
(its generation alone takes half an hour).

    python tests/gpu_large_extra.py [SCALE]      (log committed under profiles/)
    python tests/gpu_large_extra.py full         BASELINE configs[4] at its FULL size on ONE device: 10 M x 1 M,
        1 B ratings, k = 256 (generated on the GPU: synth.make_ratings_device), through size-independent properties --
        the exported order is a conflict-free permutation (oracle's checker); the first rounds of the epoch, run one
        launch per round, are bit-exact against the oracle replaying the same ratings; two independent runs of two
        full epochs (persistent kernel, graph replay) give identical factors and a falling RMSE; predictions of a
        sample are the oracle's bits -- and the epoch is timed.
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


def full_size_single_device():
    LR, LAM, SEED = 0.01, 0.05, 3
    orc = Oracle()
    t0 = time.time()
    w = mf.synth.workload("cfg4_powerlaw", 1.0, generator="device", log=lambda s: print(s, flush=True))
    U, I, k, u, i, r = w["U"], w["I"], w["k"], w["u"], w["i"], w["r"]
    print(f"cfg4_powerlaw x 1.0 (device generator): {U} x {I}, {w['nnz']} ratings, k = {k}; generated in {time.time() - t0:.0f} s; "
          f"heaviest item {int(np.bincount(i, minlength=I).max())}, heaviest user {int(np.bincount(u, minlength=U).max())}", flush=True)
    with mf.MatrixFactorizationSGD(U, I, k, LR, LAM, SEED, host_threads=16) as m:
        t0 = time.time()
        m.set_ratings(u, i, r)
        info = m.schedule_info()
        print(f"set_ratings {time.time() - t0:.1f} s: B={info['blocks']} W={info['waves']} lds={info['lds_bytes']} steps={info['total_steps']} "
              f"rows/epoch={info['total_rows']} chunks={info['chunks']} split_cells={info['split_cells']} device_ingest={info['device_ingest']}", flush=True)
        t0 = time.time()
        order, cell_ptr = m.order()
        rc = orc.check_block_schedule(u, i, U, I, order, cell_ptr, info["rounds"], info["blocks"])
        print(f"exported order: conflict-free permutation check = {rc} ({time.time() - t0:.0f} s)", flush=True)
        assert rc == 0
        # the first rounds, one launch per round, against the oracle
        t0 = time.time()
        m.init_factors(SEED)
        n_rounds = 2
        for rd in range(n_rounds):
            m.debug_round_stamps(rd)
        P, Q = m.get_factors()
        Po, Qo = orc.init_factors(U, I, k, SEED)
        hi = int(cell_ptr[n_rounds * info["blocks"]])
        orc.sgd_pass_ordered(Po, Qo, u, i, r, order[:hi], LR, LAM)
        assert np.array_equal(P, Po) and np.array_equal(Q, Qo), "first rounds differ from the oracle"
        print(f"rounds 0..{n_rounds - 1} ({hi} ratings, one launch per round): factors bit-exact against the oracle ({time.time() - t0:.0f} s)", flush=True)
        del Po, Qo
        # two independent runs of two full epochs
        t0 = time.time()
        m.init_factors(SEED)
        rm = m.fit(2)
        P1, Q1 = m.get_factors()
        m.init_factors(SEED)
        m.fit(2, rmse=False)
        P2, Q2 = m.get_factors()
        assert np.array_equal(P1, P2) and np.array_equal(Q1, Q2), "two runs of the same epochs differ"
        assert rm[1] < rm[0]
        print(f"two runs of two epochs: identical factors; rmse {rm.tolist()} ({time.time() - t0:.0f} s)", flush=True)
        del P2, Q2
        n_s = 2_000_000
        np.testing.assert_array_equal(m.predict(u[:n_s], i[:n_s]), orc.predict(P1, Q1, u[:n_s], i[:n_s]))
        print(f"predict: {n_s} pairs, the oracle's bits", flush=True)
        ms, launches = m.train_timed(3)
        c = m.debug_counters()
        print(f"timed: {ms / 3:.2f} ms per epoch = {w['nnz'] / (ms / 3) / 1e6:.3f} G updates/s, algorithmic roofline fraction "
              f"{w['nnz'] * (16 * k + 12) / (ms / 3 * 1e-3) / 8e12:.3f}; launches {launches}; counters {c}", flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "full":
    full_size_single_device()
    sys.exit(0)
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
