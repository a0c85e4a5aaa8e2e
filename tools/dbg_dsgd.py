import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mfsgd_amd as mf
from tests.oracle_bind import Oracle
from tests.dsgd_common import LAM, LR, SEED, rank_workload
from mfsgd_amd import _lib
orc = Oracle()
G, U_local, I, k, nnz = 2, 500, 333, 64, 20000
dev = torch.device("cuda", 0)
for flags in (0, _lib.FLAG_NO_GRAPH):
    u, i, r = rank_workload(0, U_local, I, nnz)
    t = mf.MatrixFactorizationSGD(U_local, I, k, LR, LAM, SEED, n_parts=G, flags=flags)
    t.set_ratings(u, i, r); t.init_p_offset(SEED, 0)
    P0, _ = t.get_factors()
    for part in range(G):
        info = t.schedule_info(part)
        blk_h = t.part_init_q(part, SEED, U_local * G)
        blk = torch.from_numpy(blk_h).to(dev)
        t.part_train(part, blk.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize()
        P1, _ = t.get_factors()
        order, _ = t.order(part)
        Po = P0.copy(); Qo = np.ascontiguousarray(blk_h[:, :k])
        orc.sgd_pass_ordered(Po, Qo, u, i // G, r, order, LR, LAM)
        print("flags", flags, "part", part, "B", info["blocks"], "nnz", info["nnz"], "P eq", np.array_equal(P1, Po), "Q eq", np.array_equal(blk.cpu().numpy()[:, :k], Qo),
              "maxdiff", np.abs(P1-Po).max())
        P0 = P1
    t.close()
