"""The global DSGD partitioner (mfsgd_dsgd_plan / mfsgd_dsgd_plan_ex): one rating set -> G user
ranges balanced by rating count x G item partitions balanced by rating count and CHAIN-AWARE (round 3): the
chain-critical items packed into as few partitions as the balance allows.  Host only."""
import numpy as np
import pytest

from tests.dsgd_common import plan_shards, plan_trainer


def _skewed(U, I, n, seed):
    rng = np.random.default_rng(seed)
    wu, wi = 1.0 / (np.arange(U) + 3.0), 1.0 / (np.arange(I) + 1.5)
    key = np.unique(rng.choice(U, n, p=wu / wu.sum()).astype(np.int64) * I + rng.choice(I, n, p=wi / wi.sum()))
    key = rng.permutation(key)
    return (key // I).astype(np.int32), (key % I).astype(np.int32), (rng.random(key.size) * 4 + 1).astype(np.float32)


@pytest.mark.parametrize("G", [1, 2, 3, 8])
def test_plan_is_a_balanced_partition(mf, G):
    U, I = 4000, 700
    u, i, _ = _skewed(U, I, 120000, G)
    du, di = np.bincount(u, minlength=U), np.bincount(i, minlength=I)
    ub, ip = mf.dsgd_plan(du, di, G)
    assert ub[0] == 0 and ub[-1] == U and (np.diff(ub) >= 1).all()
    assert ip.min() >= 0 and ip.max() < G
    shard = np.array([du[ub[g]:ub[g + 1]].sum() for g in range(G)])
    part = np.bincount(ip, weights=di, minlength=G)
    # a range boundary can be off by at most one user's ratings; LPT by at most one item's
    assert shard.max() - shard.min() <= 2 * du.max()
    assert part.max() - part.min() <= di.max()
    rows = np.bincount(ip, minlength=G)
    assert rows.max() - rows.min() <= max(2, I // (4 * G))
    ub2, ip2 = mf.dsgd_plan(du, di, G)  # deterministic
    np.testing.assert_array_equal(ub, ub2)
    np.testing.assert_array_equal(ip, ip2)


def test_plan_chain_aware_mode(mf):
    """mfsgd_dsgd_plan_ex's chain-aware mode (chain_crit > 0): plain LPT (chain_crit = 0, = mfsgd_dsgd_plan) deals the G
    heaviest items out one per partition -- equal partition times, what a ring wants, but the worst case for the SUM
    over the partitions of their heaviest items' chains; the chain-aware mode packs the chain-critical items together:
    the sum drops, the rating-count balance stays within one item, the heavy partitions are the first ones and are
    filled in descending order."""
    U, I, G = 50000, 9000, 8
    rng = np.random.default_rng(5)
    di = (2.0e6 / (np.arange(I) + 20.0)).astype(np.int64) + 1  # Zipf-Mandelbrot head: ~100 K, 95 K, 91 K, ...
    di = di[rng.permutation(I)]
    di[rng.choice(I, 50, replace=False)] = 0  # a few items nobody rated
    du = np.full(U, di.sum() // U, np.int64)
    ub, ip, info = mf.dsgd_plan_ex(du, di, G, 1, 64, chain_crit=0.3)
    load = np.bincount(ip, weights=di, minlength=G)
    assert load.max() - load.min() <= di.max()
    heaviest = np.array([di[ip == p].max() for p in range(G)])
    assert info["sum_max_chain"] == heaviest.sum() and info["critical_items"] > G and 1 <= info["sequential_parts"] < G
    ub_lpt, ip_lpt, info_lpt = mf.dsgd_plan_ex(du, di, G, 1, 64)  # plain LPT
    ub1, ip1 = mf.dsgd_plan(du, di, G)
    np.testing.assert_array_equal(ip_lpt, ip1)
    np.testing.assert_array_equal(ub_lpt, ub1)
    np.testing.assert_array_equal(ub_lpt, ub)
    assert info_lpt["critical_items"] == 0 and info_lpt["sequential_parts"] == 0
    rows_lpt = np.bincount(ip_lpt, minlength=G)
    assert rows_lpt.max() - rows_lpt.min() <= max(2, I // (4 * G))  # LPT + unrated items dealt out: even row counts
    load_lpt = np.bincount(ip_lpt, weights=di, minlength=G)
    assert load_lpt.max() - load_lpt.min() <= di.max()
    assert info["sum_max_chain"] < 0.5 * info_lpt["sum_max_chain"], (info, info_lpt)
    # sequential partitions: every item of partition p is at least as heavy as every item of partition p + 1
    for p in range(info["sequential_parts"] - 1):
        assert di[(ip == p) & (di > 0)].min() >= di[ip == p + 1].max()  # (unrated items are dealt out by row count)
    assert (ip[di >= info["threshold"]] < info["sequential_parts"]).all()
    # two partitions per rank: the same items, cut into twice as many partitions
    ub2, ip2, info2 = mf.dsgd_plan_ex(du, di, G, 2, 64, chain_crit=0.3)
    assert ub2.size == G + 1 and ip2.max() == 2 * G - 1
    load2 = np.bincount(ip2, weights=di, minlength=2 * G)
    assert load2.max() - load2.min() <= di.max()
    with pytest.raises(mf.MfsgdError):
        mf.dsgd_plan_ex(du, di, G, 1, 257)


def test_plan_edge_cases(mf):
    # more devices than rated users / items; unrated rows everywhere
    du = np.array([0, 5, 0, 0, 7, 0], np.int64)
    di = np.array([12, 0, 0], np.int64)
    ub, ip = mf.dsgd_plan(du, di, 4)
    assert ub[0] == 0 and ub[-1] == 6 and (np.diff(ub) >= 0).all() and (np.diff(ub) >= 1).all()
    assert sorted(np.bincount(ip, minlength=4).tolist()) == [0, 1, 1, 1]
    ub, ip = mf.dsgd_plan(np.array([3], np.int64), np.array([3], np.int64), 3)  # one user, three devices
    assert ub[0] == 0 and ub[-1] == 1 and (np.diff(ub) >= 0).all()
    with pytest.raises(mf.MfsgdError):
        mf.dsgd_plan(np.array([-1], np.int64), np.array([1], np.int64), 2)


def test_planned_handles_cover_every_rating_exactly_once(mf, oracle):
    """G handles built from one global set: the (device, partition) orders together are a permutation
    of all ratings, sub-epochs are conflict-free across devices, and the Q-block seeds of every
    partition reproduce the single-device initialisation."""
    U, I, k, G = 900, 260, 12, 4
    u, i, r = _skewed(U, I, 30000, 1)
    ub, ip, sel = plan_shards(mf, U, I, u, i, G)
    trainers = [plan_trainer(mf, g, ub, ip, sel, I, k, u, i, r, G) for g in range(G)]
    seen = np.zeros(u.size, np.int32)
    for s in range(G):
        users, items = [], []
        for g in range(G):
            part = (g + s) % G
            order, cell_ptr = trainers[g].order(part)
            info = trainers[g].schedule_info(part)
            gi = sel[g][order]
            seen[gi] += 1
            assert (ip[i[gi]] == part).all() and ((u[gi] >= ub[g]) & (u[gi] < ub[g + 1])).all()
            # inside the partition the schedule is the usual conflict-free block schedule (local ids)
            ul, il = u[sel[g]] - ub[g], trainers[g].item_partition()[1][i[sel[g]]]
            assert oracle.check_block_schedule(ul[order], il[order], int(ub[g + 1] - ub[g]), trainers[g].part_rows(part),
                                               np.arange(order.size), cell_ptr, info["rounds"], info["blocks"]) == 0
            users.append(np.unique(u[gi]))
            items.append(np.unique(i[gi]))
        assert np.unique(np.concatenate(users)).size == sum(x.size for x in users)
        assert np.unique(np.concatenate(items)).size == sum(x.size for x in items)
    assert (seen == 1).all()
    _, Q0 = oracle.init_factors(U, I, k, 21)
    part_of, row_of = trainers[0].item_partition()
    np.testing.assert_array_equal(part_of, ip)
    for part in range(G):
        blk = trainers[0].part_init_q(part, 21, U)
        idx = np.flatnonzero(ip == part)
        assert blk.shape[0] == idx.size == trainers[0].part_rows(part)
        np.testing.assert_array_equal(row_of[idx], np.arange(idx.size))
        np.testing.assert_array_equal(blk[:, :k], Q0[idx])
        assert (blk[:, k:] == 0).all()
    P0, _ = oracle.init_factors(U, I, k, 21)
    for g, t in enumerate(trainers):
        np.testing.assert_array_equal(t.get_factors()[0], P0[ub[g]:ub[g + 1]])
        t.close()


def test_set_item_partition_validation(mf):
    with mf.MatrixFactorizationSGD(10, 6, 8, 0.01, 0.05, 1) as m:
        with pytest.raises(mf.MfsgdError):
            m.set_item_partition(np.zeros(6, np.int32))  # single partition handle
    with mf.MatrixFactorizationSGD(10, 6, 8, 0.01, 0.05, 1, n_parts=3) as m:
        with pytest.raises(mf.MfsgdError):
            m.set_item_partition(np.array([0, 1, 2, 3, 0, 0], np.int32))  # partition out of range
        m.set_item_partition(np.array([2, 2, 0, 1, 0, 2], np.int32))
        assert [m.part_rows(p) for p in range(3)] == [2, 1, 3]
        m.set_ratings([0, 1], [0, 3], [1.0, 2.0])
        assert [m.schedule_info(p)["nnz"] for p in range(3)] == [0, 1, 1]
        with pytest.raises(mf.MfsgdError):
            m.set_item_partition(None)  # after set_ratings
    with mf.MatrixFactorizationSGD(10, 6, 8, 0.01, 0.05, 1, n_parts=3) as m:
        m.set_item_partition(np.array([2, 2, 0, 1, 0, 2], np.int32))
        m.set_item_partition(None)
        part, row = m.item_partition()
        np.testing.assert_array_equal(part, np.arange(6) % 3)
        np.testing.assert_array_equal(row, np.arange(6) // 3)
