"""The CPU oracle against the only external pins that exist (java.util.Random as
specified by the JDK; the hand-computed KAT of SURVEY.md 8c) and against its own
committed golden vectors.  PARITY UNPINNED: the reference holds no code, fixture
or vector (/root/reference/README.md:1-2)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def test_jrandom_known_answers(oracle):
    # java.util.Random: new Random(42).nextInt() and new Random(0).nextInt()
    assert oracle.jrandom_ints(42, 1)[0] == -1170105035
    assert oracle.jrandom_ints(0, 1)[0] == -1155484576
    # second draws follow from the recurrence seed = seed*0x5DEECE66D + 0xB mod 2^48
    def ref(seed, n):
        s = (seed ^ 0x5DEECE66D) & ((1 << 48) - 1)
        out = []
        for _ in range(n):
            s = (s * 0x5DEECE66D + 0xB) & ((1 << 48) - 1)
            v = s >> 16
            out.append(v - (1 << 32) if v >= (1 << 31) else v)
        return out
    for seed in (0, 1, 42, -7, 2**40 + 12345):
        assert oracle.jrandom_ints(seed, 16) == ref(seed, 16)


def test_jrandom_float_double(oracle):
    fl = oracle.jrandom_floats(42, 4)
    assert all(0.0 <= x < 1.0 for x in fl)
    s = (42 ^ 0x5DEECE66D) & ((1 << 48) - 1)
    s = (s * 0x5DEECE66D + 0xB) & ((1 << 48) - 1)
    assert fl[0] == np.float32((s >> 24) / float(1 << 24))
    d = oracle.jrandom_doubles(42, 2)
    assert all(0.0 <= x < 1.0 for x in d)
    # new Random(42).nextDouble() as printed by the JDK: 0.7275636800328681
    assert d[0] == 0.7275636800328681


def test_hand_kat_2x2(oracle):
    # e = 0.5 for both ratings, lr = 0.1, lambda = 0
    P = np.array([[0.5], [0.25]], np.float32)
    Q = np.array([[1.0], [2.0]], np.float32)
    oracle.sgd_pass(P, Q, [0, 1], [0, 1], [1.0, 1.0], 0.1, 0.0)
    np.testing.assert_array_equal(P.ravel(), np.array([0.55, 0.35], np.float32))
    np.testing.assert_array_equal(Q.ravel(), np.array([1.025, 2.0125], np.float32))


def test_update_formula_op_by_op(oracle):
    # the canonical arithmetic written out in numpy float32, k = 4 (one chunk)
    rng = np.random.default_rng(5)
    for _ in range(50):
        p = rng.random(4, dtype=np.float32)
        q = rng.random(4, dtype=np.float32)
        r, lr, lam = np.float32(3.25), np.float32(0.01), np.float32(0.05)
        f = np.float32
        t0 = f(p[0] * q[0]); t1 = f(p[1] * q[1])
        t0 = f(np.float64(p[2]) * np.float64(q[2]) + np.float64(t0))  # fma: one rounding
        t1 = f(np.float64(p[3]) * np.float64(q[3]) + np.float64(t1))
        dot = f(t0 + t1)
        assert oracle.dot(p, q) == dot
        e = f(r - dot); c = f(f(1.0) - f(lr * lam))
        s = f(-np.float64(lr) * np.float64(dot) + np.float64(f(lr * r)))  # fma(-lr, dot, lr*r)
        pe = np.array([f(np.float64(s) * np.float64(q[j]) + np.float64(f(c * p[j]))) for j in range(4)], f)
        qe = np.array([f(np.float64(s) * np.float64(p[j]) + np.float64(f(c * q[j]))) for j in range(4)], f)
        p2, q2 = p.copy(), q.copy()
        err = oracle.sgd_update(p2, q2, float(r), float(lr), float(lam))
        assert err == e
        np.testing.assert_array_equal(p2, pe)
        np.testing.assert_array_equal(q2, qe)


def test_dot_close_to_float64(oracle):
    rng = np.random.default_rng(1)
    for k in (1, 3, 4, 7, 8, 31, 32, 64, 100, 128, 256):
        p = rng.standard_normal(k).astype(np.float32)
        q = rng.standard_normal(k).astype(np.float32)
        ref = float(np.dot(p.astype(np.float64), q.astype(np.float64)))
        bound = 2e-6 * float(np.abs(p.astype(np.float64) * q.astype(np.float64)).sum()) + 1e-30
        assert abs(oracle.dot(p, q) - ref) <= bound


def test_init_factors_layout(oracle):
    P, Q = oracle.init_factors(5, 3, 4, 42)
    fl = np.array(oracle.jrandom_floats(42, 32), np.float32) * np.float32(1.0 / np.sqrt(4.0))
    np.testing.assert_array_equal(P.ravel(), fl[:20])
    np.testing.assert_array_equal(Q.ravel(), fl[20:32])


def test_multithreaded_equals_sequential(oracle, mf):
    w = mf.synth.workload("cfg1_ml100k", scale=0.3)
    with mf.MatrixFactorizationSGD(w["U"], w["I"], w["k"], 0.01, 0.05, 1) as m:
        m.set_ratings(w["u"], w["i"], w["r"])
        info = m.schedule_info()
        order, cell_ptr = m.order()
    P1, Q1 = oracle.init_factors(w["U"], w["I"], w["k"], 9)
    P2, Q2 = P1.copy(), Q1.copy()
    for _ in range(2):
        oracle.sgd_pass_ordered(P1, Q1, w["u"], w["i"], w["r"], order, 0.01, 0.05)
        oracle.sgd_epoch_mt(P2, Q2, w["u"], w["i"], w["r"], order, cell_ptr, info["rounds"], info["blocks"], 0.01, 0.05, 4)
    np.testing.assert_array_equal(P1, P2)
    np.testing.assert_array_equal(Q1, Q2)


def test_checker_rejects_bad_schedules(oracle):
    u = np.array([0, 0, 1, 1], np.int32)
    i = np.array([0, 1, 0, 1], np.int32)
    order = np.arange(4, dtype=np.int64)
    # one round, two cells {0,1} {2,3}: both touch item 0 -> conflict
    assert oracle.check_block_schedule(u, i, 2, 2, order, np.array([0, 2, 4], np.int64), 1, 2) == 2
    # two rounds of two cells, diagonal: ok
    order = np.array([0, 3, 1, 2], np.int64)
    assert oracle.check_block_schedule(u, i, 2, 2, order, np.array([0, 1, 2, 3, 4], np.int64), 2, 2) == 0
    # not a permutation
    order = np.array([0, 0, 1, 2], np.int64)
    assert oracle.check_block_schedule(u, i, 2, 2, order, np.array([0, 1, 2, 3, 4], np.int64), 2, 2) == 1


def test_golden_vectors(oracle, mf):
    """Self-generated fixtures (tests/golden/make_golden.py): RMSE trajectory and factor
    checksums of the oracle in natural rating order.  They pin the oracle against
    accidental change; they are NOT reference outputs (none exist)."""
    with open(os.path.join(HERE, "golden", "oracle_golden.json")) as f:
        gold = json.load(f)
    from tests.golden.make_golden import run_case

    for case in gold["cases"]:
        got = run_case(oracle, mf, case["workload"], case["scale"], case["seed"], case["epochs"], case["lr"], case["lambda"])
        assert got["rmse"] == case["rmse"], case["workload"]
        assert got["p_sha256"] == case["p_sha256"] and got["q_sha256"] == case["q_sha256"], case["workload"]


def _textbook_double(U, I, k, u, i, r, order, seed, epochs, lr, lam, oracle):
    """Plain textbook SGD in float64, the loop a Java reference would contain:
    e = r - p.q;  p += lr*(e*q - lam*p);  q += lr*(e*p_old - lam*q), same initial factors, same order."""
    P32, Q32 = oracle.init_factors(U, I, k, seed)
    P, Q = P32.astype(np.float64), Q32.astype(np.float64)
    out = []
    for _ in range(epochs):
        for j in order:
            p, q = P[u[j]].copy(), Q[i[j]].copy()  # old values on the right-hand sides
            e = r[j] - p.dot(q)
            P[u[j]] = p + lr * (e * q - lam * p)
            Q[i[j]] = q + lr * (e * p - lam * q)
        err = r - np.einsum("nk,nk->n", P[u], Q[i])
        out.append(float(np.sqrt(np.mean(err * err))))
    return np.array(out)


def test_contract_tracks_textbook_double_sgd_within_1e5(oracle, mf):
    """BASELINE.json's tolerance is an RMSE trajectory within 1e-5 of the reference CPU path.
    No such path exists to compare with, so the nearest stand-in is checked: a float64
    textbook loop over the same order and seeds.  (The GPU equals the oracle bit for bit.)"""
    # k = 8, 32, then the headline shape k = 64 and the DSGD configs' k = 128 and k = 256 (scaled so that the
    # pure-Python float64 loop finishes in seconds)
    for name, scale, epochs in (("cfg0_dense100x80", 1.0, 5), ("cfg1_ml100k", 0.3, 3), ("cfg2_ml20m", 0.002, 3),
                                ("cfg3_netflix", 0.002, 3), ("cfg4_powerlaw", 0.0001, 3)):
        w = mf.synth.workload(name, scale)
        with mf.MatrixFactorizationSGD(w["U"], w["I"], w["k"], 0.01, 0.05, 4) as m:
            m.set_ratings(w["u"], w["i"], w["r"])
            order, _ = m.order()
        P, Q = oracle.init_factors(w["U"], w["I"], w["k"], 4)
        got = []
        for _ in range(epochs):
            oracle.sgd_pass_ordered(P, Q, w["u"], w["i"], w["r"], order, 0.01, 0.05)
            got.append(oracle.rmse(P, Q, w["u"], w["i"], w["r"]))
        ref = _textbook_double(w["U"], w["I"], w["k"], w["u"], w["i"], w["r"].astype(np.float64), order, 4, epochs,
                               0.01, 0.05, oracle)
        assert np.abs(np.array(got) - ref).max() < 1e-5, (name, got, ref.tolist())


def test_order_matters_more_than_arithmetic(oracle, mf):
    """What "identical seeds" buys a CPU path that does NOT follow mfsgd_get_order: SGD is order
    dependent, and the RMSE trajectory of the same arithmetic in natural input order differs from
    the canonical (scheduled) order by 1e-4 .. 1e-2 -- orders of magnitude above BASELINE.json's
    1e-5, which is therefore only attainable (and attained: the test above, ~1e-7) by a path that
    visits the ratings in the exported order.  Measured at full cfg1 (k = 32): 6.3e-4; cfg2 x 0.02
    (k = 64): 1.4e-3; cfg3 x 0.004 (k = 128): 2.8e-3; cfg4 x 0.0005 (k = 256): 2.9e-3 (10 epochs;
    DESIGN.md section 3)."""
    for name, scale in (("cfg1_ml100k", 0.5), ("cfg2_ml20m", 0.005)):
        w = mf.synth.workload(name, scale)
        with mf.MatrixFactorizationSGD(w["U"], w["I"], w["k"], 0.01, 0.05, 4) as m:
            m.set_ratings(w["u"], w["i"], w["r"])
            order, _ = m.order()
        P1, Q1 = oracle.init_factors(w["U"], w["I"], w["k"], 4)
        P2, Q2 = P1.copy(), Q1.copy()
        gap = 0.0
        for _ in range(5):
            oracle.sgd_pass(P1, Q1, w["u"], w["i"], w["r"], 0.01, 0.05)
            oracle.sgd_pass_ordered(P2, Q2, w["u"], w["i"], w["r"], order, 0.01, 0.05)
            gap = max(gap, abs(oracle.rmse(P1, Q1, w["u"], w["i"], w["r"]) - oracle.rmse(P2, Q2, w["u"], w["i"], w["r"])))
        assert 1e-5 < gap < 2e-2, (name, gap)


def test_textbook_fp32_loop_tracks_the_contract_within_1e5(oracle, mf):
    """The second CPU baseline bench.py reports (oracle/mfsgd_oracle.c mfo_textbook_*: a plain left-to-right
    fp32 loop, p += lr*(e*q - lambda*p), no tree, no regrouping) on the same order and seeds: its RMSE
    trajectory stays within BASELINE.json's 1e-5 of the contract's (the gap is rounding only), its
    multithreaded block-schedule form equals its sequential form bit for bit, and it is NOT the contract
    (the factors differ in the last bits)."""
    for name, scale, epochs in (("cfg1_ml100k", 1.0, 5), ("cfg2_ml20m", 0.01, 5), ("cfg3_netflix", 0.004, 3)):
        w = mf.synth.workload(name, scale)
        with mf.MatrixFactorizationSGD(w["U"], w["I"], w["k"], 0.01, 0.05, 4) as m:
            m.set_ratings(w["u"], w["i"], w["r"])
            order, cell_ptr = m.order()
            info = m.schedule_info()
        P, Q = oracle.init_factors(w["U"], w["I"], w["k"], 4)
        Pt, Qt = P.copy(), Q.copy()
        Pm, Qm = P.copy(), Q.copy()
        gap = 0.0
        for _ in range(epochs):
            oracle.sgd_pass_ordered(P, Q, w["u"], w["i"], w["r"], order, 0.01, 0.05)
            oracle.textbook_pass_ordered(Pt, Qt, w["u"], w["i"], w["r"], order, 0.01, 0.05)
            oracle.textbook_epoch_mt(Pm, Qm, w["u"], w["i"], w["r"], order, cell_ptr, info["rounds"], info["blocks"], 0.01, 0.05, 4)
            gap = max(gap, abs(oracle.rmse(P, Q, w["u"], w["i"], w["r"]) - oracle.rmse(Pt, Qt, w["u"], w["i"], w["r"])))
        assert gap < 1e-5, (name, gap)
        np.testing.assert_array_equal(Pt, Pm)
        np.testing.assert_array_equal(Qt, Qm)
        assert not np.array_equal(P, Pt)
