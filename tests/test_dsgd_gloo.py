"""The N > 1 path on CPU: dsgd.py's ring rotation over torch.distributed (gloo),
world_size 2 and 3, against the sequential definition of a DSGD epoch."""
import os
import socket
import subprocess
import sys

import pytest

from tests.conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,m", [(2, 1), (3, 1), (2, 3), (2, 0), (3, 0)])  # m = 0: one global set cut by mfsgd_dsgd_plan
def test_dsgd_ring_over_gloo(mf, world, m):
    env = dict(os.environ, OMP_NUM_THREADS="1", MFSGD_TEST_PARTS_PER_RANK=str(max(m, 1)),
               MFSGD_TEST_PLANNED="1" if m == 0 else "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dsgd_gloo_worker.py")]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:]
    assert "dsgd gloo ok" in p.stdout
