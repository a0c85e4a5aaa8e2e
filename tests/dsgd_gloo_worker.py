#!/usr/bin/env python3
"""One rank of the gloo DSGD test (launched by tests/test_dsgd_gloo.py through
torch.distributed.run).  Runs dsgd.DSGD with the CPU stand-in backend over gloo
and checks the result against the sequential definition on rank 0."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

import mfsgd_amd as mf
from mfsgd_amd.dsgd import DSGD, TorchDistRing, assemble_q
from tests.dsgd_common import (LAM, LR, SEED, OracleBackend, assemble_q_plan, plan_shards, plan_trainer, rank_workload,
                               sequential_dsgd, sequential_dsgd_plan)
from tests.oracle_bind import Oracle


def planned(rank, world):
    """One GLOBAL skewed rating set, cut by mfsgd_dsgd_plan: every rank computes the same plan and
    keeps its own user range; checked against the sequential definition over global factors."""
    U, I, k, nnz, epochs = 150, 90, 8, 2500, 2
    rng = np.random.default_rng(5)
    wu, wi = 1.0 / (np.arange(U) + 2.0), 1.0 / (np.arange(I) + 1.5)
    key = np.unique(rng.choice(U, nnz, p=wu / wu.sum()).astype(np.int64) * I + rng.choice(I, nnz, p=wi / wi.sum()))
    key = rng.permutation(key)
    u, i = (key // I).astype(np.int32), (key % I).astype(np.int32)
    r = (rng.random(u.size) * 4 + 1).astype(np.float32)
    orc = Oracle()
    ub, ip, sel = plan_shards(mf, U, I, u, i, world)
    t = plan_trainer(mf, rank, ub, ip, sel, I, k, u, i, r, world)
    ul, il, rl = u[sel[rank]] - ub[rank], i[sel[rank]], r[sel[rank]]
    backend = OracleBackend(torch, orc, t, ul, il, rl, k, world)
    ring = TorchDistRing(dist, rank, world)
    d = DSGD(backend, ring, rank, world, I, t.kp, SEED, U, ul.size, parts_per_rank=1)
    sse = []
    for _ in range(epochs):
        d.epoch()
        tot, cnt = ring.sum_f64([d.sse(), float(ul.size)], torch, torch.device("cpu"))
        assert cnt == u.size
        sse.append(tot)
    gathered = [None] * world
    dist.all_gather_object(gathered, (rank, backend.P, d.home_blocks()))
    if rank == 0:
        P = np.concatenate([g[1] for g in sorted(gathered, key=lambda x: x[0])])
        blocks = {}
        for g in gathered:
            blocks.update(g[2])
        Q = assemble_q_plan(blocks, ip, I, k)
        trainers = [plan_trainer(mf, g, ub, ip, sel, I, k, u, i, r, world) for g in range(world)]
        Ps, Qs, sse_s = sequential_dsgd_plan(orc, trainers, sel, U, I, k, u, i, r, world, epochs)
        assert np.array_equal(P, Ps), "P differs from the sequential DSGD definition"
        assert np.array_equal(Q, Qs), "Q differs from the sequential DSGD definition"
        assert np.allclose(sse, sse_s, rtol=1e-12), (sse, sse_s)
        print("dsgd gloo ok", world, "planned", sse)
    dist.barrier()
    dist.destroy_process_group()


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if os.environ.get("MFSGD_TEST_PLANNED") == "1":
        return planned(rank, world)
    U_local, I, k, nnz, epochs = 40, 37, 8, 500, 2
    m = int(os.environ.get("MFSGD_TEST_PARTS_PER_RANK", "1"))
    n_parts = world * m
    orc = Oracle()
    u, i, r = rank_workload(rank, U_local, I, nnz)
    t = mf.MatrixFactorizationSGD(U_local, I, k, LR, LAM, SEED, n_parts=n_parts)
    t.set_ratings(u, i, r)
    t.init_p_offset(SEED, rank * U_local)
    backend = OracleBackend(torch, orc, t, u, i, r, k, n_parts)
    ring = TorchDistRing(dist, rank, world)
    d = DSGD(backend, ring, rank, world, I, t.kp, SEED, U_local * world, nnz, parts_per_rank=m)
    sse = []
    for _ in range(epochs):
        d.epoch()
        assert d.group == rank, "blocks are not home after an epoch"
        tot, cnt = ring.sum_f64([d.sse(), float(nnz)], torch, torch.device("cpu"))
        assert cnt == nnz * world
        sse.append(tot)
    gathered = [None] * world
    dist.all_gather_object(gathered, (rank, backend.P, d.home_blocks()))
    if rank == 0:
        P = np.concatenate([g[1] for g in sorted(gathered, key=lambda x: x[0])])
        blocks = {}
        for g in gathered:
            blocks.update(g[2])
        assert sorted(blocks) == list(range(n_parts))
        Q = assemble_q(blocks, I, k, n_parts)
        trainers, data = [], []
        for g in range(world):
            ug, ig, rg = rank_workload(g, U_local, I, nnz)
            tg = mf.MatrixFactorizationSGD(U_local, I, k, LR, LAM, SEED, n_parts=n_parts)
            tg.set_ratings(ug, ig, rg)
            trainers.append(tg)
            data.append((ug, ig, rg))
        Ps, Qs, sse_s = sequential_dsgd(orc, trainers, data, U_local, I, k, world, epochs, parts_per_rank=m)
        assert np.array_equal(P, Ps), "P differs from the sequential DSGD definition"
        assert np.array_equal(Q, Qs), "Q differs from the sequential DSGD definition"
        assert np.allclose(sse, sse_s, rtol=1e-12), (sse, sse_s)
        assert sse[-1] < sse[0]
        print("dsgd gloo ok", world, sse)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
