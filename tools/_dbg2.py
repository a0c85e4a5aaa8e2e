import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import mfsgd_amd as mf
from mfsgd_amd import _lib
from tests.oracle_bind import Oracle
orc = Oracle()
LR, LAM = 0.01, 0.05
k, W = 64, 4
rng = np.random.default_rng(k * 10 + W)
U, I = 3000, 80
u = list(range(U)) + list(rng.integers(0, U, 9000))
i = [7] * U + list(rng.integers(0, I, 9000))
key = rng.permutation(np.unique(np.array(u) * I + np.array(i)))
u, i, r = (key // I).astype(np.int32), (key % I).astype(np.int32), (rng.random(key.size) * 4 + 1).astype(np.float32)
tag = sys.argv[1] if len(sys.argv) > 1 else ""
flags = _lib.FLAG_NO_GRAPH if "nograph" in tag else 0
res = []
for rep in range(4):
    with mf.MatrixFactorizationSGD(U, I, k, LR, LAM, 3, blocks=5, waves=W, flags=flags) as m:
        m.set_ratings(u, i, r)
        m.init_factors(3)
        lone = int((m.debug_schedule()[0][:, 5] & 1).sum())
        m.fit(1, rmse=False)
        P, Q = m.get_factors()
        order, cell_ptr = m.order()
    Po, Qo = orc.init_factors(U, I, k, 3)
    orc.sgd_pass_ordered(Po, Qo, u, i, r, order, LR, LAM)
    badp = np.flatnonzero((P != Po).any(axis=1)); badq = np.flatnonzero((Q != Qo).any(axis=1))
    # which cells hold the first wrong user?
    first = None
    if badp.size:
        pos = {int(x): j for j, x in enumerate(order)}
        js = sorted(pos[int(j)] for j in np.flatnonzero(np.isin(u, badp[:1])))
        first = [int(np.searchsorted(cell_ptr, j, side="right") - 1) for j in js[:4]]
    res.append((badp.size, badq.size, badq[:6].tolist(), first))
print(tag, "lone", lone, res, flush=True)
