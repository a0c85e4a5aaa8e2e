import sys, ctypes as C
sys.path.insert(0, '/root/repo')
order = sys.argv[1]
import mfsgd_amd
from mfsgd_amd import _lib
lib = _lib.load_library()
def mine():
    n = C.c_int32(-1)
    rc = lib.mfsgd_device_count(C.byref(n))
    print("mfsgd_device_count rc", rc, "n", n.value, flush=True)
def tor():
    import torch
    print("torch.cuda.is_available", torch.cuda.is_available(), torch.cuda.device_count(), flush=True)
    if torch.cuda.is_available():
        x = torch.ones(4, device="cuda"); print("torch tensor ok", float(x.sum()), flush=True)
if order == "mine_first":
    mine(); tor(); mine()
else:
    tor(); mine()
