"""Shared pieces of the DSGD tests: per-rank workloads, the sequential
definition of one DSGD epoch, and a CPU stand-in for HipBackend (oracle
arithmetic on the PRODUCT's schedules) used to exercise dsgd.py under gloo."""
import numpy as np

LR, LAM, SEED = 0.01, 0.05, 21


def rank_workload(rank, U_local, I, nnz, seed=100):
    rng = np.random.default_rng(seed + rank)
    key = rng.choice(U_local * I, nnz, replace=False)
    u = (key // I).astype(np.int32)
    i = (key % I).astype(np.int32)
    r = (rng.random(nnz) * 4 + 1).astype(np.float32)
    return u, i, r


def sequential_dsgd(oracle, trainers, data, U_local, I, k, G, epochs, parts_per_rank=1):
    """One process, no communication: sub-epoch s, rank g trains the partitions of group
    (g+s)%G (partitions group*m .. group*m+m-1).  Within a sub-epoch the (rank, partition)
    pairs are disjoint, so any order works."""
    m = parts_per_rank
    P, Q = oracle.init_factors(U_local * G, I, k, SEED)
    sse = []
    for _ in range(epochs):
        for s in range(G):
            for g in range(G):
                for j in range(m):
                    part = ((g + s) % G) * m + j
                    u, i, r = data[g]
                    order, _ = trainers[g].order(part)
                    oracle.sgd_pass_ordered(P, Q, u + g * U_local, i, r, order, LR, LAM)
        tot = 0.0
        for g in range(G):
            u, i, r = data[g]
            tot += oracle.sse(P, Q, u + g * U_local, i, r)
        sse.append(tot)
    return P, Q, sse


# ---- one GLOBAL rating set cut by the product's partitioner (mfsgd_dsgd_plan) -----------------
def plan_shards(mf, U, I, u, i, G):
    """(user_begin, item_part, [indices of shard g's ratings])."""
    ub, ip = mf.dsgd_plan(np.bincount(u, minlength=U), np.bincount(i, minlength=I), G)
    sel = [np.flatnonzero((u >= ub[g]) & (u < ub[g + 1])) for g in range(G)]
    return ub, ip, sel


def plan_trainer(mf, g, ub, ip, sel, I, k, u, i, r, G, lr=LR, lam=LAM, seed=SEED, **kw):
    """Handle of device g: its user range, its users' ratings (local user ids, global item ids)."""
    t = mf.MatrixFactorizationSGD(int(ub[g + 1] - ub[g]), I, k, lr, lam, seed, n_parts=G, **kw)
    t.set_item_partition(ip)
    t.set_ratings(u[sel[g]] - ub[g], i[sel[g]], r[sel[g]])
    t.init_p_offset(seed, int(ub[g]))
    return t


def sequential_dsgd_plan(oracle, trainers, sel, U, I, k, u, i, r, G, epochs, lr=LR, lam=LAM, seed=SEED, mt_threads=0):
    """The sequential definition for a planned problem: global factors, sub-epoch s, device g
    trains partition (g+s)%G in the order its handle exports (indices into the shard)."""
    P, Q = oracle.init_factors(U, I, k, seed)
    sse = []
    orders = {}
    for _ in range(epochs):
        for s in range(G):
            for g in range(G):
                part = (g + s) % G
                if (g, part) not in orders:
                    order, cell_ptr = trainers[g].order(part)
                    info = trainers[g].schedule_info(part)
                    orders[(g, part)] = (sel[g][order], cell_ptr, info["rounds"], info["blocks"])
                order, cell_ptr, rounds, blocks = orders[(g, part)]
                if order.size == 0:
                    continue
                if mt_threads > 1:
                    oracle.sgd_epoch_mt(P, Q, u, i, r, order, cell_ptr, rounds, blocks, lr, lam, mt_threads)
                else:
                    oracle.sgd_pass_ordered(P, Q, u, i, r, order, lr, lam)
        sse.append(oracle.sse(P, Q, u, i, r))
    return P, Q, sse


def assemble_q_plan(blocks, ip, I, k):
    """Dense I x k matrix from {partition: block}: partition rows are its items in ascending id order."""
    Q = np.zeros((I, k), np.float32)
    for part, blk in blocks.items():
        idx = np.flatnonzero(ip == part)
        Q[idx] = blk[: idx.size, :k]
    return Q


def native_problem():
    """The rating set of the multi-process test of the C-ABI DSGD driver: every rank rebuilds it from the seed."""
    U, I, k, nnz, epochs = 2400, 900, 64, 90000, 2
    rng = np.random.default_rng(77)
    wu, wi = 1.0 / (np.arange(U) + 4.0), 1.0 / (np.arange(I) + 2.0)
    key = np.unique(rng.choice(U, nnz, p=wu / wu.sum()).astype(np.int64) * I + rng.choice(I, nnz, p=wi / wi.sum()))
    key = rng.permutation(key)
    u, i = (key // I).astype(np.int32), (key % I).astype(np.int32)
    r = (rng.random(u.size) * 4 + 1).astype(np.float32)
    return U, I, k, u, i, r, epochs


class OracleBackend:
    """CPU stand-in for dsgd.HipBackend (tests only)."""

    def __init__(self, torch, oracle, trainer, u, i, r, k, G):
        self.torch, self.o, self.t = torch, oracle, trainer
        self.u, self.i, self.r, self.k, self.G = u, i, r, k, G
        self.P, _ = trainer.get_factors()
        self.item_row = trainer.item_partition()[1]

    def new_block(self, rows, kp):
        return self.torch.zeros((rows, kp), dtype=self.torch.float32)

    def load_block(self, block, host_array):
        block[: host_array.shape[0]].copy_(self.torch.from_numpy(host_array))

    def block_to_host(self, block):
        return block.numpy().copy()

    def part_rows(self, part):
        return self.t.part_rows(part)

    def part_init_q(self, part, seed, u_total):
        return self.t.part_init_q(part, seed, u_total)

    def _sel(self, part):
        order, _ = self.t.order(part)
        return order

    def part_train(self, part, block):
        order = self._sel(part)
        rows = self.part_rows(part)
        Qb = np.ascontiguousarray(block.numpy()[:rows, : self.k])
        self.o.sgd_pass_ordered(self.P, Qb, self.u, self.item_row[self.i], self.r, order, LR, LAM)
        block[:rows, : self.k] = self.torch.from_numpy(Qb)

    def part_sse(self, part, block):
        order = self._sel(part)
        rows = self.part_rows(part)
        Qb = np.ascontiguousarray(block.numpy()[:rows, : self.k])
        return self.o.sse(self.P, Qb, self.u[order], self.item_row[self.i][order], self.r[order])

    def synchronize(self):
        pass


def fuzz_cases(n_cases, seed=2024, max_ratings=2500):
    """Random small problems: sizes, density, skew, k and explicit geometry all drawn at random."""
    rng = np.random.default_rng(seed)
    for _ in range(n_cases):
        U = int(rng.integers(1, 120))
        I = int(rng.integers(1, 120))
        n = int(min(max_ratings, max(1, rng.integers(1, U * I + 1))))
        if rng.random() < 0.5:  # skewed: a few heavy rows
            wu = 1.0 / (np.arange(U) + 1.0) ** rng.uniform(0.0, 1.5)
            wi = 1.0 / (np.arange(I) + 1.0) ** rng.uniform(0.0, 1.5)
            u = rng.choice(U, n, p=wu / wu.sum())
            i = rng.choice(I, n, p=wi / wi.sum())
        else:
            u = rng.integers(0, U, n)
            i = rng.integers(0, I, n)
        if rng.random() < 0.7:  # usually de-duplicated, sometimes with repeated pairs
            key = np.unique(u.astype(np.int64) * I + i)
            key = rng.permutation(key)
            u, i = key // I, key % I
        r = (rng.standard_normal(u.size) * 2 + 3).astype(np.float32)
        k = int(rng.choice([1, 2, 3, 4, 7, 8, 16, 20, 32, 33, 64, 96, 128, 130, 256]))
        W = int(rng.choice([0, 1, 2, 4, 8]))
        B = int(rng.choice([0, 0, 1, 2, 3, 5]))
        yield dict(U=U, I=I, k=k, u=u.astype(np.int32), i=i.astype(np.int32), r=r, blocks=B, waves=W,
                   lr=float(rng.choice([0.01, 0.05])), lam=float(rng.choice([0.0, 0.05])))


def fuzz_chunked_cases(n_cases, seed=4242, max_ratings=9000):
    """Random problems sized so that cells overflow the LDS image and are chunked: large k, few
    blocks, hundreds of rows per side, skew and repeated pairs included."""
    rng = np.random.default_rng(seed)
    for _ in range(n_cases):
        U = int(rng.integers(150, 1200))
        I = int(rng.integers(150, 1200))
        n = int(rng.integers(max_ratings // 4, max_ratings))
        if rng.random() < 0.6:
            wu = 1.0 / (np.arange(U) + 1.0) ** rng.uniform(0.0, 1.2)
            wi = 1.0 / (np.arange(I) + 1.0) ** rng.uniform(0.0, 1.2)
            u = rng.choice(U, n, p=wu / wu.sum())
            i = rng.choice(I, n, p=wi / wi.sum())
        else:
            u = rng.integers(0, U, n)
            i = rng.integers(0, I, n)
        if rng.random() < 0.7:
            key = rng.permutation(np.unique(u.astype(np.int64) * I + i))
            u, i = key // I, key % I
        r = (rng.standard_normal(u.size) * 2 + 3).astype(np.float32)
        k = int(rng.choice([64, 100, 128, 200, 256]))
        W = int(rng.choice([1, 2, 4, 8]))
        B = int(rng.choice([1, 1, 2, 3, 4]))
        yield dict(U=U, I=I, k=k, u=u.astype(np.int32), i=i.astype(np.int32), r=r, blocks=B, waves=W,
                   lr=float(rng.choice([0.01, 0.05])), lam=float(rng.choice([0.0, 0.05])))
