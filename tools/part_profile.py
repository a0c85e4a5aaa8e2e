#!/usr/bin/env python3
"""Per-partition timing of ONE rank of an N-rank DSGD job on one GPU (no communication): for every item partition of
the rank's handle the time of one training launch (mean of several, HIP events through the stream), next to what the
scheduler built for it.  The numbers behind DESIGN.md section 6.

    python tools/part_profile.py [--world 8] [--workload cfg2_ml20m] [--scaling weak|strong] [--parts-per-rank 1]
                                 [--blocks B] [--waves W] [--reps 5]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--workload", default="cfg2_ml20m")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--scaling", default="weak")
    ap.add_argument("--parts-per-rank", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=0)
    ap.add_argument("--waves", type=int, default=0)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--round-launch", action="store_true")
    ap.add_argument("--plan-crit", type=float, default=0.0)
    a = ap.parse_args()
    import torch

    import mfsgd_amd
    from mfsgd_amd import _lib, synth

    N, ppr = a.world, a.parts_per_rank
    if a.scaling == "strong":
        w = synth.workload(a.workload, a.scale)
        du = np.bincount(w["u"], minlength=w["U"]).astype(np.int64)
        di = np.bincount(w["i"], minlength=w["I"]).astype(np.int64)
        ub, ip, info = mfsgd_amd.dsgd_plan_ex(du, di, N, ppr, w["k"], a.plan_crit)
        lo, hi = int(ub[a.rank]), int(ub[a.rank + 1])
        sel = np.flatnonzero((w["u"] >= lo) & (w["u"] < hi))
        U, u, i, r = hi - lo, (w["u"][sel] - lo).astype(np.int32), w["i"][sel], w["r"][sel]
        u_total, u_off = w["U"], lo
    else:
        w = synth.workload(a.workload, a.scale, seed_offset=1000 * a.rank, item_mult=N)
        di = np.bincount(w["i"], minlength=w["I"]).astype(np.int64) * N
        _, ip, info = mfsgd_amd.dsgd_plan_ex(np.ones(N, np.int64), di, N, ppr, w["k"], a.plan_crit)
        U, u, i, r = w["U"], w["u"], w["i"], w["r"]
        u_total, u_off = U * N, a.rank * U
    print(f"{a.workload} x{a.scale} {a.scaling} world {N} rank {a.rank}: {u.size} ratings, {U} users, {w['I']} items, k={w['k']}; plan {info}")
    flags = _lib.FLAG_ROUND_LAUNCH if a.round_launch else 0
    n_parts = N * ppr
    dev = torch.device("cuda", 0)
    with mfsgd_amd.MatrixFactorizationSGD(U, w["I"], w["k"], 0.01, 0.05, 3, n_parts=n_parts, blocks=a.blocks, waves=a.waves,
                                          flags=flags) as m:
        m.set_item_partition(ip)
        m.set_ratings(u, i, r)
        m.init_p_offset(3, u_off)
        stream = torch.cuda.current_stream(dev)
        tot = 0.0
        deg_local = np.bincount(i, minlength=w["I"])
        for p in range(n_parts):
            inf = m.schedule_info(p)
            blk = torch.from_numpy(m.part_init_q(p, 3, u_total)).to(dev)
            if inf["nnz"] == 0:
                continue
            for _ in range(2):
                m.part_train(p, blk.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(a.reps):
                m.part_train(p, blk.data_ptr(), stream.cuda_stream)
            e1.record(stream)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.reps
            tot += ms
            mx = int(deg_local[ip == p].max())
            print(f"  part {p:2d}: nnz {inf['nnz']:9d} rows {m.part_rows(p):7d} B {inf['blocks']:3d} W {inf['waves']} lds {inf['lds_bytes']:6d} "
                  f"steps {inf['total_steps']:8d} rows/epoch {inf['total_rows']:8d} max_cell_steps {inf['max_cell_steps']:5d} "
                  f"sum_round_steps {inf['sum_round_steps']:6d} split {inf['split_cells']:4d} heaviest item {mx:6d}: {ms:7.3f} ms "
                  f"({inf['nnz'] / ms / 1e6:6.2f} G/s)")
        print(f"  sum {tot:.3f} ms per epoch -> {u.size / tot / 1e6:.3f} G updates/s per rank, "
              f"frac {u.size * (16 * w['k'] + 12) / (tot * 1e-3) / 8e12:.3f}")


if __name__ == "__main__":
    main()
