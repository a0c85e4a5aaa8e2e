"""The host scheduler, without a GPU: every schedule is a conflict-free
permutation (oracle checker), and replaying the device arrays with the kernel's
exact LDS access order (tests/kernel_emulator.py) reproduces the sequential
oracle bit for bit -- i.e. the hazard rules the pipelined kernel relies on hold."""
import numpy as np
import pytest

from tests.kernel_emulator import replay_epoch

LR, LAM = 0.01, 0.05


def _check(mf, oracle, U, I, k, u, i, r, replay=True, **kw):
    u = np.asarray(u, np.int32)
    i = np.asarray(i, np.int32)
    r = np.asarray(r, np.float32)
    with mf.MatrixFactorizationSGD(U, I, k, LR, LAM, 3, **kw) as m:
        m.set_ratings(u, i, r)
        info = m.schedule_info()
        order, cell_ptr = m.order()
        sched = m.debug_schedule()
    assert info["nnz"] == u.size
    assert oracle.check_block_schedule(u, i, U, I, order, cell_ptr, info["rounds"], info["blocks"]) == 0
    # tiles marked for the mailbox hand-off (CellDesc.rsv[0] bit 0): ONE item, the same in every cell of the tile, every
    # cell a single chunk with work (a regression: "one item row per cell" alone let two items share such a tile)
    B = info["blocks"]
    if B >= 1 and sched[0].shape[0] >= B * B and not info["swapped"]:
        lone = (sched[0][: B * B, 5] & 1).reshape(B, B)  # [block, tile]
        tile_item = {}
        for c in range(B * B):  # canonical order: rounds, then blocks; block b holds tile (b + round) % B
            rnd, b = divmod(c, B)
            t = (b + rnd) % B
            if lone[b, t]:
                assert lone[:, t].all() and sched[0][b * B + t, 4] == 0, (b, t)
                items = np.unique(i[order[cell_ptr[c]:cell_ptr[c + 1]]])
                assert items.size == 1 and tile_item.setdefault(t, items[0]) == items[0], (b, t, items)
    if replay and u.size:
        P, Q = oracle.init_factors(U, I, k, 3)
        Pe, Qe = P.copy(), Q.copy()
        for _ in range(2):
            oracle.sgd_pass_ordered(P, Q, u, i, r, order, LR, LAM)
            if info["swapped"]:  # users sit on the kernel's q side: hand the replay (Q, P)
                replay_epoch(oracle, Qe, Pe, k, LR, LAM, sched, info["blocks"], info["waves"], info["slots"], info["group_lanes"])
            else:
                replay_epoch(oracle, Pe, Qe, k, LR, LAM, sched, info["blocks"], info["waves"], info["slots"], info["group_lanes"])
        np.testing.assert_array_equal(Pe, P)
        np.testing.assert_array_equal(Qe, Q)
    return info


def test_cfg0_dense(mf, oracle):
    w = mf.synth.workload("cfg0_dense100x80")
    _check(mf, oracle, w["U"], w["I"], w["k"], w["u"], w["i"], w["r"])


def test_cfg1_scaled(mf, oracle):
    w = mf.synth.workload("cfg1_ml100k", scale=0.2)
    _check(mf, oracle, w["U"], w["I"], w["k"], w["u"], w["i"], w["r"])


def test_cfg2_scaled_skewed(mf, oracle):
    w = mf.synth.workload("cfg2_ml20m", scale=0.002)
    info = _check(mf, oracle, w["U"], w["I"], w["k"], w["u"], w["i"], w["r"])
    assert info["slots"] == 4 and info["group_lanes"] == 16


@pytest.mark.parametrize("k", [1, 3, 4, 5, 8, 17, 32, 64, 100, 128, 200, 256])
def test_all_geometries(mf, oracle, k):
    rng = np.random.default_rng(k)
    U, I, n = 60, 45, 900
    key = rng.choice(U * I, n, replace=False)
    info = _check(mf, oracle, U, I, k, key // I, key % I, rng.random(n) * 4 + 1)
    L = info["group_lanes"]
    assert L * info["slots"] == 64 and info["kp"] == 4 * L and 4 * L >= k


@pytest.mark.parametrize("B,W", [(1, 1), (2, 1), (3, 2), (4, 4), (5, 8)])
def test_explicit_blocks_and_waves(mf, oracle, B, W):
    rng = np.random.default_rng(B * 10 + W)
    U, I, n = 80, 70, 1500
    key = rng.choice(U * I, n, replace=False)
    info = _check(mf, oracle, U, I, 16, key // I, key % I, rng.random(n), blocks=B, waves=W)
    assert info["blocks"] == B and info["waves"] == W


def test_edge_cases(mf, oracle):
    # empty
    _check(mf, oracle, 5, 5, 8, [], [], [])
    # a single rating
    _check(mf, oracle, 5, 5, 8, [2], [3], [4.0])
    # one user rating everything / one item rated by everyone (pure chains)
    _check(mf, oracle, 1, 40, 8, [0] * 40, list(range(40)), np.arange(40) * 0.1)
    _check(mf, oracle, 40, 1, 8, list(range(40)), [0] * 40, np.arange(40) * 0.1)
    # duplicate (u, i) pairs: legal input, must be applied one after the other
    _check(mf, oracle, 3, 3, 4, [0, 0, 0, 1, 1, 2, 0], [1, 1, 1, 2, 2, 0, 1], [1, 2, 3, 4, 5, 6, 7])
    # users / items that never occur, ragged degrees
    _check(mf, oracle, 50, 50, 12, [49, 49, 49, 0, 7], [0, 49, 25, 0, 7], [1, 2, 3, 4, 5])


def test_hot_item_runs(mf, oracle):
    # one item rated by 300 users + background: exercises run mode (resident q rows)
    rng = np.random.default_rng(9)
    U, I = 300, 50
    u = list(range(U)) + list(rng.integers(0, U, 600))
    i = [7] * U + list(rng.integers(0, I, 600))
    key = np.unique(np.array(u) * I + np.array(i))
    info = _check(mf, oracle, U, I, 64, key // I, key % I, rng.random(key.size), blocks=2, waves=2)
    assert info["total_steps"] > 0


def test_hot_user_swaps_roles(mf, oracle):
    # one user rating 300 items: the chain is on the user side, so the roles are exchanged
    rng = np.random.default_rng(10)
    U, I = 50, 300
    u = [7] * I + list(rng.integers(0, U, 600))
    i = list(range(I)) + list(rng.integers(0, I, 600))
    key = np.unique(np.array(u) * I + np.array(i))
    info = _check(mf, oracle, U, I, 64, key // I, key % I, rng.random(key.size), blocks=2, waves=2)
    assert info["swapped"] == 1


def test_schedule_is_deterministic(mf):
    w = mf.synth.workload("cfg1_ml100k", scale=0.3)
    orders = []
    for threads in (1, 4):
        with mf.MatrixFactorizationSGD(w["U"], w["I"], w["k"], LR, LAM, 3, host_threads=threads) as m:
            m.set_ratings(w["u"], w["i"], w["r"])
            orders.append(m.order()[0])
    np.testing.assert_array_equal(orders[0], orders[1])


def test_lpt_bucket_ring_and_heap_give_the_same_assignment(mf, monkeypatch):
    """[r3] lpt_assign walks the bins through a ring of buckets indexed by load (large inputs) or a binary heap (small
    ones, MFSGD_LPT_HEAP=1): the same greedy, so the same schedule, word for word -- skewed counts, many equal counts,
    an item with a tile of its own, auto and explicit geometry."""
    rng = np.random.default_rng(11)
    U, I = 30000, 9000
    n = 400000
    u = rng.integers(0, U, n)
    i = np.minimum((rng.pareto(1.1, n) * 40).astype(np.int64), I - 1)  # heavy head, long tail of equal small counts
    key = np.unique(u.astype(np.int64) * I + i)
    u, i = (key // I).astype(np.int32), (key % I).astype(np.int32)
    r = (rng.random(key.size) * 4 + 1).astype(np.float32)
    for kw in ({}, {"blocks": 24, "waves": 4}, {"blocks": 8, "waves": 2}):
        got = []
        for heap in (False, True):
            if heap:
                monkeypatch.setenv("MFSGD_LPT_HEAP", "1")
            else:
                monkeypatch.delenv("MFSGD_LPT_HEAP", raising=False)
            with mf.MatrixFactorizationSGD(U, I, 64, LR, LAM, 3, **kw) as m:
                m.set_ratings(u, i, r)
                got.append((m.order(), m.debug_schedule(), m.schedule_info()))
        np.testing.assert_array_equal(got[0][0][0], got[1][0][0])
        np.testing.assert_array_equal(got[0][0][1], got[1][0][1])
        for a, b in zip(got[0][1], got[1][1]):
            np.testing.assert_array_equal(a, b)
        assert got[0][2]["blocks"] * got[0][2]["waves"] >= 16  # (the ring is taken from 16 bins and 4 096 rows up)


def test_partitioned_schedules_cover_everything(mf, oracle):
    w = mf.synth.workload("cfg1_ml100k", scale=0.3)
    G = 3
    with mf.MatrixFactorizationSGD(w["U"], w["I"], w["k"], LR, LAM, 3, n_parts=G) as m:
        m.set_ratings(w["u"], w["i"], w["r"])
        seen = np.zeros(w["nnz"], bool)
        for part in range(G):
            order, cell_ptr = m.order(part)
            info = m.schedule_info(part)
            assert (w["i"][order] % G == part).all()
            assert not seen[order].any()
            seen[order] = True
            # conflict-free within the partition (local item ids)
            sel = order
            remap = {x: j for j, x in enumerate(sel)}
            assert oracle.check_block_schedule(w["u"][sel], w["i"][sel] // G, w["U"], m.part_rows(part),
                                               np.arange(sel.size, dtype=np.int64), cell_ptr, info["rounds"], info["blocks"]) == 0
        assert seen.all()


def test_fuzz_schedules_replay_exactly(mf, oracle):
    from tests.dsgd_common import fuzz_cases

    ok = 0
    for c in fuzz_cases(40, max_ratings=1200):
        try:
            with mf.MatrixFactorizationSGD(c["U"], c["I"], c["k"], c["lr"], c["lam"], 3, blocks=c["blocks"], waves=c["waves"]) as m:
                m.set_ratings(c["u"], c["i"], c["r"])
                info = m.schedule_info()
                order, cell_ptr = m.order()
                sched = m.debug_schedule()
        except mf.MfsgdError as e:
            assert e.code == -7, e  # an explicit B too small for the LDS image is a legal refusal
            continue
        assert oracle.check_block_schedule(c["u"], c["i"], c["U"], c["I"], order, cell_ptr, info["rounds"], info["blocks"]) == 0
        P, Q = oracle.init_factors(c["U"], c["I"], c["k"], 3)
        Pe, Qe = P.copy(), Q.copy()
        oracle.sgd_pass_ordered(P, Q, c["u"], c["i"], c["r"], order, c["lr"], c["lam"])
        a, b = (Qe, Pe) if info["swapped"] else (Pe, Qe)
        replay_epoch(oracle, a, b, c["k"], c["lr"], c["lam"], sched, info["blocks"], info["waves"], info["slots"], info["group_lanes"])
        np.testing.assert_array_equal(Pe, P)
        np.testing.assert_array_equal(Qe, Q)
        ok += 1
    assert ok >= 30


# ---- chunked cells: a cell whose LDS image is too large is cut into chunks ---------------------
@pytest.mark.parametrize("k,B,W,n", [(256, 1, 2, 700), (128, 2, 4, 2500), (64, 1, 4, 2500), (200, 3, 1, 2000)])
def test_oversize_cells_are_chunked(mf, oracle, k, B, W, n):
    rng = np.random.default_rng(k + B)
    U, I = 900, 800
    key = rng.choice(U * I, n, replace=False)
    info = _check(mf, oracle, U, I, k, key // I, key % I, rng.random(n) * 4 + 1, blocks=B, waves=W)
    assert info["blocks"] == B
    assert info["split_cells"] >= 1 and info["chunks"] > B * B
    assert info["lds_bytes"] <= 160 * 1024 - 512


def test_chunk_split_by_items_and_by_users(mf, oracle):
    # one user, many items: only the item side can be cut; and the mirror image
    n = 400
    info = _check(mf, oracle, 1, n, 256, [0] * n, list(range(n)), np.arange(n) * 0.01, blocks=1, waves=1)
    assert info["split_cells"] == 1 and info["chunks"] >= 3
    info = _check(mf, oracle, n, 1, 256, list(range(n)), [0] * n, np.arange(n) * 0.01, blocks=1, waves=1)
    assert info["split_cells"] == 1 and info["chunks"] >= 3


def test_chunks_leave_fitting_schedules_alone(mf):
    # a schedule that fits is not cut, whatever else changed: chunks == cells
    w = mf.synth.workload("cfg1_ml100k", scale=0.5)
    with mf.MatrixFactorizationSGD(w["U"], w["I"], w["k"], LR, LAM, 3) as m:
        m.set_ratings(w["u"], w["i"], w["r"])
        info = m.schedule_info()
    assert info["split_cells"] == 0 and info["chunks"] == info["blocks"] ** 2


def test_chunked_hot_item_at_large_k(mf, oracle):
    # skewed popularity at k = 128 with few blocks: the hot item's cells overflow, the rest do not
    rng = np.random.default_rng(9)
    U, I, n = 3000, 400, 30000
    wgt = 1.0 / (np.arange(I) + 3.0)
    ii = rng.choice(I, n, p=wgt / wgt.sum())
    uu = rng.integers(0, U, n)
    key = np.unique(uu.astype(np.int64) * I + ii)
    info = _check(mf, oracle, U, I, 128, key // I, key % I, rng.random(key.size) * 4 + 1, blocks=8, waves=4)
    assert info["split_cells"] >= 1


def test_fuzz_chunked_schedules_replay_exactly(mf, oracle):
    from tests.dsgd_common import fuzz_chunked_cases

    split = 0
    for c in fuzz_chunked_cases(6, seed=99, max_ratings=5000):
        with mf.MatrixFactorizationSGD(c["U"], c["I"], c["k"], c["lr"], c["lam"], 3, blocks=c["blocks"],
                                       waves=c["waves"]) as m:
            m.set_ratings(c["u"], c["i"], c["r"])
            info = m.schedule_info()
            order, cell_ptr = m.order()
            sched = m.debug_schedule()
        assert oracle.check_block_schedule(c["u"], c["i"], c["U"], c["I"], order, cell_ptr, info["rounds"], info["blocks"]) == 0
        P, Q = oracle.init_factors(c["U"], c["I"], c["k"], 3)
        Pe, Qe = P.copy(), Q.copy()
        oracle.sgd_pass_ordered(P, Q, c["u"], c["i"], c["r"], order, c["lr"], c["lam"])
        args = (c["k"], c["lr"], c["lam"], sched, info["blocks"], info["waves"], info["slots"], info["group_lanes"])
        if info["swapped"]:
            replay_epoch(oracle, Qe, Pe, *args)
        else:
            replay_epoch(oracle, Pe, Qe, *args)
        np.testing.assert_array_equal(Pe, P)
        np.testing.assert_array_equal(Qe, Q)
        split += info["split_cells"] > 0
    assert split >= 3


# ---- solo runs: one item's chain as compact records, stored by a helper wave on the device ------------
def _solo_steps(m):
    cells, rows, subs, entries = m.debug_schedule()
    return int((subs[:, 0] >> 16).sum())


def test_solo_runs_form_for_a_dominant_item_and_replay_exactly(mf, oracle):
    from mfsgd_amd import _lib

    rng = np.random.default_rng(21)
    U, I = 400, 50
    u = list(range(U)) + list(rng.integers(0, U, 900))
    i = [7] * U + list(rng.integers(0, I, 900))
    key = np.unique(np.array(u) * I + np.array(i))
    uu, ii, rr = key // I, key % I, rng.random(key.size) * 4 + 1
    for k, W, B in ((64, 2, 2), (128, 4, 2), (256, 2, 8), (64, 1, 2)):  # k = 256: a cell holds ~140 rows
        _check(mf, oracle, U, I, k, uu, ii, rr, blocks=B, waves=W)
        with mf.MatrixFactorizationSGD(U, I, k, LR, LAM, 3, blocks=B, waves=W) as m:
            m.set_ratings(uu, ii, rr)
            assert _solo_steps(m) >= U // 2, (k, W, "the hot item's ratings should sit in solo runs")
            order_solo = m.order()[0]
        with mf.MatrixFactorizationSGD(U, I, k, LR, LAM, 3, blocks=B, waves=W, flags=_lib.FLAG_NO_SOLO) as m:
            m.set_ratings(uu, ii, rr)
            assert _solo_steps(m) == 0
            assert sorted(m.order()[0].tolist()) == sorted(order_solo.tolist())
    # no solo runs where the kernel has no two-wave loops for them: k <= 32 (fewer than 16 lanes), W = 8
    for k, W in ((32, 2), (64, 8)):
        with mf.MatrixFactorizationSGD(U, I, k, LR, LAM, 3, blocks=2, waves=W) as m:
            m.set_ratings(uu, ii, rr)
            assert _solo_steps(m) == 0


def test_repeated_pairs_never_go_solo(mf, oracle):
    # the hot item's ratings include the same (user, item) pair twice: a solo run stores p rows late,
    # so such a chain must stay an ordinary run
    U, I = 200, 20
    u = list(range(U)) + [5, 5, 9]
    i = [3] * U + [3, 3, 3]
    r = np.linspace(1, 5, len(u))
    _check(mf, oracle, U, I, 64, u, i, r, blocks=1, waves=2)
    with mf.MatrixFactorizationSGD(U, I, 64, LR, LAM, 3, blocks=1, waves=2) as m:
        m.set_ratings(u, i, r)
        cells, rows, subs, entries = m.debug_schedule()
        # user sub-group holding users 5 and 9 has repeats; at most the other sub-group went solo
        assert int((subs[:, 0] >> 16).sum()) < U


def test_item_with_a_tile_of_its_own(mf, oracle):
    """An item whose chain of dependent updates is what the epoch waits for gets its tile to itself
    (schedule.cpp, lpt_assign): the cells of that tile hold one item row and nothing but its ratings.  When
    those cells are the ONLY large ones and have to be chunked (here: 600 users x 512-byte rows per cell),
    the chunk limits must come out of their sizes, not of the small cells' (a regression: 11 219 chunks of
    four rows, no solo run left)."""
    rng = np.random.default_rng(1002)
    U, I, k = 3000, 80, 100
    u = list(range(U)) + list(rng.integers(0, U, 9000))
    i = [7] * U + list(rng.integers(0, I, 9000))
    key = rng.permutation(np.unique(np.array(u) * I + np.array(i)))
    uu, ii, rr = key // I, key % I, rng.random(key.size) * 4 + 1
    info = _check(mf, oracle, U, I, k, uu, ii, rr, blocks=5, waves=2)
    assert info["split_cells"] >= 5 and info["chunks"] < 200, info
    with mf.MatrixFactorizationSGD(U, I, k, LR, LAM, 3, blocks=5, waves=2) as m:
        m.set_ratings(uu, ii, rr)
        assert _solo_steps(m) == U
        cells, rows, subs, entries = m.debug_schedule()
        order, cell_ptr = m.order()
    # the hot item's tile: every rating of its cells is a rating of item 7
    ii = np.asarray(ii)
    B = info["blocks"]
    alone = 0
    for c in range(B * B):
        sel = order[cell_ptr[c]:cell_ptr[c + 1]]
        if sel.size and (ii[sel] == 7).any():
            assert (ii[sel] == 7).all(), c
            alone += 1
    assert alone == B


def test_lone_tiles_are_marked_for_the_mailbox_hand_off(mf, oracle):
    """A tile that is one item row, one chunk, in EVERY cell carries kCellLoneTile (CellDesc.rsv[0] bit 0) in all
    its cells -- the persistent kernel passes such a row on through the tile's mailbox; no other cell carries it,
    and a tile whose cells had to be chunked does not either."""
    rng = np.random.default_rng(77)
    U, I, k, B = 800, 60, 64, 4
    u = list(range(U)) + list(rng.integers(0, U, 3000))
    i = [9] * U + list(rng.integers(0, I, 3000))
    key = rng.permutation(np.unique(np.array(u) * I + np.array(i)))
    uu, ii, rr = key // I, key % I, rng.random(key.size) * 4 + 1
    _check(mf, oracle, U, I, k, uu, ii, rr, blocks=B, waves=2)
    with mf.MatrixFactorizationSGD(U, I, k, LR, LAM, 3, blocks=B, waves=2) as m:
        m.set_ratings(uu, ii, rr)
        cells = m.debug_schedule()[0]
        order, cell_ptr = m.order()
    assert cells.shape[0] == B * B, "nothing chunked here"
    lone = (cells[:, 5] & 1).reshape(B, B)  # [block, tile]
    tiles = np.flatnonzero(lone.all(axis=0))
    assert tiles.size == 1 and lone.sum() == B, lone
    for c in range(B * B):  # canonical order: rounds, then blocks; block b holds tile (b + round) % B
        rnd, b = divmod(c, B)
        sel = order[cell_ptr[c]:cell_ptr[c + 1]]
        if (b + rnd) % B == tiles[0]:
            assert sel.size and (np.asarray(ii)[sel] == 9).all(), c
        else:
            assert not (np.asarray(ii)[sel] == 9).any(), c
    # rows too short for a wave of granules (k <= 32), or a giant whose cells are chunked: not marked
    for kk, UU in ((32, 800), (256, 3000)):
        u2 = list(range(UU)) + list(rng.integers(0, UU, 3000))
        i2 = [9] * UU + list(rng.integers(0, I, 3000))
        key = np.unique(np.array(u2) * I + np.array(i2))
        with mf.MatrixFactorizationSGD(UU, I, kk, LR, LAM, 3, blocks=B, waves=2) as m:
            m.set_ratings(key // I, key % I, rng.random(key.size) * 4 + 1)
            assert (m.debug_schedule()[0][:, 5] & 1).sum() == 0, kk


def _two_items_one_tile():
    """Items 1 and 2 share a tile, each rated by one user of a different block: every cell of that tile holds ONE
    item row -- but not the same one.  (Found by tests/gpu_fuzz_extra.py: such a tile was handed on through the
    mailbox as if it were one row.)"""
    u = np.array([2, 3, 4, 0, 1], np.int32)
    i = np.array([0, 0, 0, 1, 2], np.int32)
    r = np.array([4.0, 3.0, 5.0, 2.0, 1.0], np.float32)
    return 5, 3, u, i, r


def test_two_items_in_a_tile_are_not_a_lone_tile(mf, oracle):
    U, I, u, i, r = _two_items_one_tile()
    info = _check(mf, oracle, U, I, 64, u, i, r, blocks=2, waves=1)  # (_check verifies what the marked tiles hold)
    with mf.MatrixFactorizationSGD(U, I, 64, LR, LAM, 3, blocks=2, waves=1) as m:
        m.set_ratings(u, i, r)
        cells = m.debug_schedule()[0]
        order, cell_ptr = m.order()
    # the situation the test is about did arise: a tile whose two cells hold one item each, different ones
    per_tile = {}
    for c in range(4):
        rnd, b = divmod(c, 2)
        per_tile.setdefault((b + rnd) % 2, []).append(sorted(set(i[order[cell_ptr[c]:cell_ptr[c + 1]]].tolist())))
    shared = [t for t, v in per_tile.items() if sorted(v) == [[1], [2]]]
    assert len(shared) == 1, per_tile
    lone = (cells[:, 5] & 1).reshape(2, 2)  # [block, tile]
    assert not lone[:, shared[0]].any()
    assert lone[:, 1 - shared[0]].all()  # the other tile IS one item (item 0), rated from both blocks


def test_fuzz_dominant_items_replay_exactly(mf, oracle):
    """Random problems with one to three dominant items (tiles of their own, one sub-cell per cell, solo runs,
    chunked when k is large): conflict check, the invariants of the marked tiles and the emulator's replay of the
    device arrays against the oracle's sequential pass (all inside _check)."""
    rng = np.random.default_rng(9090)
    lone = 0
    for _ in range(24):
        k = int(rng.choice([16, 64, 64, 100, 128, 256]))
        B = int(rng.integers(2, 9))
        U = int(rng.integers(20, 60)) * B
        I = int(rng.integers(12, 60))
        u, i = [], []
        for h in rng.choice(I, size=int(rng.integers(1, 4)), replace=False):
            sel = np.flatnonzero(rng.random(U) < rng.uniform(0.5, 1.0))
            u += sel.tolist()
            i += [int(h)] * sel.size
        m = int(rng.integers(1, 4)) * U
        u += rng.integers(0, U, m).tolist()
        i += rng.integers(0, I, m).tolist()
        key = rng.permutation(np.unique(np.array(u, np.int64) * I + np.array(i, np.int64)))
        uu, ii = (key // I).astype(np.int32), (key % I).astype(np.int32)
        rr = (rng.random(key.size) * 4 + 1).astype(np.float32)
        W = int(rng.choice([1, 2, 2, 4]))
        try:
            _check(mf, oracle, U, I, k, uu, ii, rr, blocks=B, waves=W)
        except mf.MfsgdError as e:
            assert e.code == -7, e  # an explicit B too small for the LDS image is a legal refusal
            continue
        with mf.MatrixFactorizationSGD(U, I, k, LR, LAM, 3, blocks=B, waves=W) as m2:
            m2.set_ratings(uu, ii, rr)
            lone += int((m2.debug_schedule()[0][:, 5] & 1).any())
    assert lone >= 2, lone  # (the marked tiles need k >= 64, unchunked cells and an item heavy enough for its B)
