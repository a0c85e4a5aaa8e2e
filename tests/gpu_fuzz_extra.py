"""One-off stress beyond the test suite: 400 small random problems, 80 chunk-heavy ones and 80 with dominant items, two
epochs each on the GPU, factors compared bit for bit with the oracle replaying the exported order --
once with the schedule built on the host, once with degrees, bucket order and step packing on the
device (round 2: pack.hip, mixed mode for the chunk-heavy ones), whose canonical order must also be
the host's.

    python tests/gpu_fuzz_extra.py         # prints "done, mismatches: 0"

Uses the oracle, so it lives under tests/ (not collected by pytest; run it by hand on a GPU box)."""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mfsgd_amd as mf
from tests.oracle_bind import Oracle
from mfsgd_amd import _lib
from tests.dsgd_common import fuzz_cases, fuzz_chunked_cases
orc = Oracle()
bad = 0
n_dev = 0
def check(c, tag, n):
    global bad, n_dev
    u, i, r = c["u"], c["i"], c["r"]
    try:
        with mf.MatrixFactorizationSGD(c["U"], c["I"], c["k"], c["lr"], c["lam"], 5, blocks=c["blocks"], waves=c["waves"],
                                       flags=_lib.FLAG_HOST_INGEST) as m:
            m.set_ratings(u, i, r)
            host_order = m.order()[0]
    except mf.MfsgdError as e:
        assert e.code == -7, e  # an explicit B too small for the LDS image: a legal refusal
        return
    with mf.MatrixFactorizationSGD(c["U"], c["I"], c["k"], c["lr"], c["lam"], 5, blocks=c["blocks"], waves=c["waves"],
                                   flags=_lib.FLAG_DEVICE_INGEST) as m:
        m.train(u, i, r, 2, rmse=False)
        P, Q = m.get_factors()
        order, _ = m.order()
        n_dev += m.schedule_info()["device_ingest"] == 2
    if not np.array_equal(order, host_order):
        bad += 1
        print("ORDER MISMATCH", tag, n, c["U"], c["I"], c["k"], len(u), c["blocks"], c["waves"], flush=True)
    Po, Qo = orc.init_factors(c["U"], c["I"], c["k"], 5)
    for _ in range(2):
        orc.sgd_pass_ordered(Po, Qo, u, i, r, order, c["lr"], c["lam"])
    if not (np.array_equal(P, Po) and np.array_equal(Q, Qo)):
        bad += 1
        print("MISMATCH", tag, n, c["U"], c["I"], c["k"], len(u), c["blocks"], c["waves"], flush=True)
for n, c in enumerate(fuzz_cases(400, seed=31337)):
    check(c, "small", n)
for n, c in enumerate(fuzz_chunked_cases(80, seed=777, max_ratings=20000)):
    check(c, "chunked", n)
# third family (round 2): a few dominant items, so that tiles of one item form and travel through the mailbox
# (kernels.hip run_ring) -- three epochs, i.e. three launches: the tags carry the launch generation
def lone_cases(count, seed):
    rng = np.random.default_rng(seed)
    for _ in range(count):
        k = int(rng.choice([64, 64, 100, 128, 256]))
        B = int(rng.integers(2, 33))
        U = int(rng.integers(40, 120)) * B
        I = int(rng.integers(30, 120))
        hot = rng.choice(I, size=int(rng.integers(1, 4)), replace=False)
        u, i = [], []
        for h in hot:
            sel = np.flatnonzero(rng.random(U) < rng.uniform(0.5, 1.0))
            u += sel.tolist()
            i += [int(h)] * sel.size
        m = int(rng.integers(2, 6)) * U
        u += rng.integers(0, U, m).tolist()
        i += rng.integers(0, I, m).tolist()
        key = rng.permutation(np.unique(np.array(u, np.int64) * I + np.array(i, np.int64)))
        yield dict(U=U, I=I, k=k, blocks=B, waves=int(rng.choice([1, 2, 2, 4])), lr=0.01, lam=0.05,
                   u=(key // I).astype(np.int32), i=(key % I).astype(np.int32),
                   r=(rng.random(key.size) * 4 + 1).astype(np.float32))
n_lone = 0
for n, c in enumerate(lone_cases(80, seed=4242)):
    try:
        with mf.MatrixFactorizationSGD(c["U"], c["I"], c["k"], c["lr"], c["lam"], 5, blocks=c["blocks"], waves=c["waves"]) as m:
            m.train(c["u"], c["i"], c["r"], 3, rmse=False)
            P, Q = m.get_factors()
            order, _ = m.order()
            n_lone += int((m.debug_schedule()[0][:, 5] & 1).any())
    except mf.MfsgdError as e:
        assert e.code == -7, e
        continue
    Po, Qo = orc.init_factors(c["U"], c["I"], c["k"], 5)
    for _ in range(3):
        orc.sgd_pass_ordered(Po, Qo, c["u"], c["i"], c["r"], order, c["lr"], c["lam"])
    if not (np.array_equal(P, Po) and np.array_equal(Q, Qo)):
        bad += 1
        print("MISMATCH lone", n, c["U"], c["I"], c["k"], len(c["u"]), c["blocks"], c["waves"], flush=True)
print("schedules with mailbox tiles:", n_lone, "of 80")
print("done, mismatches:", bad, "| schedules the device packed (pure or mixed):", n_dev)
