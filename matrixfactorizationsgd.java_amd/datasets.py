"""Rating files -> (u, i, r) arrays, through the library's readers (csrc/io.cpp)."""
import ctypes as C

import numpy as np

from . import _lib
from .trainer import MfsgdError

FORMATS = {"auto": 0, "ml-tsv": 1, "ml-dat": 2, "ml-csv": 3, "netflix": 4}


def load_ratings(path, fmt="auto"):
    """Returns dict(u, i, r, user_ids, item_ids, n_users, n_items): dense int32 indices,
    float32 ratings, and the original ids of every dense index (int64)."""
    lib = _lib.load_library()
    f = C.c_void_p()
    rc = lib.mfsgd_ratings_file_open(str(path).encode(), FORMATS[fmt], C.byref(f))
    if rc != 0:
        raise MfsgdError(rc, lib.mfsgd_io_last_error().decode())
    try:
        nnz, nu, ni = C.c_int64(), C.c_int32(), C.c_int32()
        lib.mfsgd_ratings_file_info(f, C.byref(nnz), C.byref(nu), C.byref(ni))
        u = np.empty(nnz.value, np.int32)
        i = np.empty(nnz.value, np.int32)
        r = np.empty(nnz.value, np.float32)
        uid = np.empty(nu.value, np.int64)
        iid = np.empty(ni.value, np.int64)
        p = lambda a, t: a.ctypes.data_as(C.POINTER(t))  # noqa: E731
        lib.mfsgd_ratings_file_read(f, p(u, C.c_int32), p(i, C.c_int32), p(r, C.c_float), p(uid, C.c_int64), p(iid, C.c_int64))
    finally:
        lib.mfsgd_ratings_file_close(f)
    return dict(u=u, i=i, r=r, user_ids=uid, item_ids=iid, n_users=nu.value, n_items=ni.value)
