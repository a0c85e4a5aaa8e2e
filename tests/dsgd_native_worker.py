#!/usr/bin/env python3
"""One rank of the multi-process test of the DSGD driver UNDER THE C-ABI (csrc/dsgd.cpp), launched by
tests/test_gpu_parity.py::test_native_dsgd_multi_process_shm.  Several real processes share the one GPU;
the blocks travel through the driver's shared-memory rehearsal transport (RCCL cannot put two ranks on
one GPU), everything else -- groups, slots, double buffers, event ordering, the RMSE reduction -- is the
code the RCCL ring runs.

    python tests/dsgd_native_worker.py RANK WORLD PARTS_PER_RANK OUT_DIR

The rehearsal transport lives in lib/libmfsgd_rehearsal.so only (MFSGD_LIBRARY points at it, MFSGD_DSGD_TRANSPORT=shm
selects it); rank 0 makes the ring id and leaves it in OUT_DIR/ring.id for the others.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import mfsgd_amd as mf  # noqa: E402
from mfsgd_amd import _lib  # noqa: E402
from mfsgd_amd.dsgd import NativeDSGD  # noqa: E402
from tests.dsgd_common import SEED, native_problem, plan_shards, plan_trainer  # noqa: E402


def main():
    rank, world, m = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    out_dir = sys.argv[4]
    id_path = os.path.join(out_dir, "ring.id")
    if rank == 0:
        uid = NativeDSGD.unique_id()
        assert uid[:8] == b"MFSGDSHM"
        with open(id_path + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(id_path + ".tmp", id_path)
    else:
        import time

        t0 = time.time()
        while not os.path.exists(id_path):
            if time.time() - t0 > 300:
                raise SystemExit("rank 0 never wrote the ring id")
            time.sleep(0.05)
        with open(id_path, "rb") as f:
            uid = f.read()
    U, I, k, u, i, r, epochs = native_problem()
    n_parts = world * m
    ub, ip, sel = plan_shards(mf, U, I, u, i, world)  # users over the ranks ...
    _, ip = mf.dsgd_plan(np.bincount(u, minlength=U), np.bincount(i, minlength=I), n_parts)  # ... items over all partitions
    # several processes share this GPU: round launches (a persistent kernel wants its workgroups co-resident)
    t = plan_trainer(mf, rank, ub, ip, sel, I, k, u, i, r, n_parts, flags=_lib.FLAG_ROUND_LAUNCH)
    with NativeDSGD(t, rank, world, uid) as d:
        assert d.m == m
        d.init_q(SEED, U)
        rm0 = d.rmse()
        rm = d.train(epochs)
        tot, cnt = d.allreduce(float(rank + 1), 1.0)
        assert (tot, cnt) == (world * (world + 1) / 2, float(world)), (tot, cnt)
        blocks = d.home_blocks()
        P, _ = t.get_factors()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), P=P, rm0=rm0, rm=rm, parts=np.array(sorted(blocks), np.int32),
             **{f"q{p}": b for p, b in blocks.items()})
    t.close()
    print(f"rank {rank} ok", flush=True)


if __name__ == "__main__":
    main()
