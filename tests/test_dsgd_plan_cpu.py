"""The global DSGD partitioner (mfsgd_dsgd_plan): one rating set -> G user ranges
balanced by rating count x G item partitions balanced by rating count.  Host only."""
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
