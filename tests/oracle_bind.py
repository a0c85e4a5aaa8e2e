"""ctypes binding of oracle/libmfsgd_oracle.so (the CPU checker).  Test
infrastructure: imported only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "libmfsgd_oracle.so")

_f = C.POINTER(C.c_float)
_i = C.POINTER(C.c_int32)
_l = C.POINTER(C.c_int64)


def _fp(a):
    return a.ctypes.data_as(_f)


def _ip(a):
    return a.ctypes.data_as(_i)


def _lp(a):
    return a.ctypes.data_as(_l)


class _JR(C.Structure):
    _fields_ = [("seed", C.c_uint64)]


class Oracle:
    def __init__(self):
        if not os.path.exists(_SO):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=subprocess.DEVNULL)
        L = self.L = C.CDLL(_SO)
        L.mfo_jrandom_next_int.restype = C.c_int32
        L.mfo_jrandom_next_float.restype = C.c_float
        L.mfo_jrandom_next_double.restype = C.c_double
        L.mfo_dot.restype = C.c_float
        L.mfo_dot.argtypes = [_f, _f, C.c_int32]
        L.mfo_sgd_update.restype = C.c_float
        L.mfo_sgd_update.argtypes = [_f, _f, C.c_int32, C.c_float, C.c_float, C.c_float]
        L.mfo_init_factors.argtypes = [_f, _f, C.c_int32, C.c_int32, C.c_int32, C.c_int64]
        L.mfo_sgd_pass.argtypes = [_f, _f, C.c_int32, _i, _i, _f, C.c_int64, C.c_float, C.c_float]
        L.mfo_sgd_pass_ordered.argtypes = [_f, _f, C.c_int32, _i, _i, _f, _l, C.c_int64, C.c_float, C.c_float]
        L.mfo_sgd_epoch_mt.argtypes = [_f, _f, C.c_int32, _i, _i, _f, _l, _l, C.c_int32, C.c_int32,
                                       C.c_float, C.c_float, C.c_int32]
        L.mfo_textbook_pass_ordered.argtypes = L.mfo_sgd_pass_ordered.argtypes
        L.mfo_textbook_epoch_mt.argtypes = L.mfo_sgd_epoch_mt.argtypes
        L.mfo_sse.restype = C.c_double
        L.mfo_sse.argtypes = [_f, _f, C.c_int32, _i, _i, _f, C.c_int64]
        L.mfo_rmse.restype = C.c_double
        L.mfo_rmse.argtypes = [_f, _f, C.c_int32, _i, _i, _f, C.c_int64]
        L.mfo_predict.argtypes = [_f, _f, C.c_int32, _i, _i, _f, C.c_int64]
        L.mfo_check_block_schedule.argtypes = [_i, _i, C.c_int64, C.c_int32, C.c_int32, _l, _l, C.c_int32, C.c_int32]

    # -- java.util.Random -------------------------------------------------------
    def jrandom_ints(self, seed, n):
        g = _JR()
        self.L.mfo_jrandom_init(C.byref(g), C.c_int64(seed))
        return [self.L.mfo_jrandom_next_int(C.byref(g)) for _ in range(n)]

    def jrandom_floats(self, seed, n):
        g = _JR()
        self.L.mfo_jrandom_init(C.byref(g), C.c_int64(seed))
        return [self.L.mfo_jrandom_next_float(C.byref(g)) for _ in range(n)]

    def jrandom_doubles(self, seed, n):
        g = _JR()
        self.L.mfo_jrandom_init(C.byref(g), C.c_int64(seed))
        return [self.L.mfo_jrandom_next_double(C.byref(g)) for _ in range(n)]

    # -- factors / arithmetic -----------------------------------------------------
    def init_factors(self, U, I, k, seed):
        P = np.empty((U, k), np.float32)
        Q = np.empty((I, k), np.float32)
        self.L.mfo_init_factors(_fp(P), _fp(Q), U, I, k, seed)
        return P, Q

    def dot(self, p, q):
        p = np.ascontiguousarray(p, np.float32)
        q = np.ascontiguousarray(q, np.float32)
        return float(self.L.mfo_dot(_fp(p), _fp(q), p.size))

    def sgd_update(self, p, q, r, lr, lam):
        return float(self.L.mfo_sgd_update(_fp(p), _fp(q), p.size, r, lr, lam))

    @staticmethod
    def _chk(P, Q, u, i, r):
        assert P.dtype == np.float32 and Q.dtype == np.float32 and P.flags.c_contiguous and Q.flags.c_contiguous
        assert P.shape[1] == Q.shape[1]
        return (np.ascontiguousarray(u, np.int32), np.ascontiguousarray(i, np.int32),
                np.ascontiguousarray(r, np.float32))

    def sgd_pass(self, P, Q, u, i, r, lr, lam):
        u, i, r = self._chk(P, Q, u, i, r)
        self.L.mfo_sgd_pass(_fp(P), _fp(Q), P.shape[1], _ip(u), _ip(i), _fp(r), u.size, lr, lam)

    def sgd_pass_ordered(self, P, Q, u, i, r, order, lr, lam):
        u, i, r = self._chk(P, Q, u, i, r)
        order = np.ascontiguousarray(order, np.int64)
        self.L.mfo_sgd_pass_ordered(_fp(P), _fp(Q), P.shape[1], _ip(u), _ip(i), _fp(r), _lp(order),
                                    order.size, lr, lam)

    def sgd_epoch_mt(self, P, Q, u, i, r, order, cell_ptr, n_rounds, n_cells, lr, lam, threads):
        u, i, r = self._chk(P, Q, u, i, r)
        order = np.ascontiguousarray(order, np.int64)
        cell_ptr = np.ascontiguousarray(cell_ptr, np.int64)
        rc = self.L.mfo_sgd_epoch_mt(_fp(P), _fp(Q), P.shape[1], _ip(u), _ip(i), _fp(r), _lp(order),
                                     _lp(cell_ptr), n_rounds, n_cells, lr, lam, threads)
        if rc != 0:
            raise RuntimeError("mfo_sgd_epoch_mt failed")

    def textbook_pass_ordered(self, P, Q, u, i, r, order, lr, lam):
        """Plain left-to-right fp32 loop (second CPU baseline; not the bit-exact contract)."""
        u, i, r = self._chk(P, Q, u, i, r)
        order = np.ascontiguousarray(order, np.int64)
        self.L.mfo_textbook_pass_ordered(_fp(P), _fp(Q), P.shape[1], _ip(u), _ip(i), _fp(r), _lp(order),
                                         order.size, lr, lam)

    def textbook_epoch_mt(self, P, Q, u, i, r, order, cell_ptr, n_rounds, n_cells, lr, lam, threads):
        u, i, r = self._chk(P, Q, u, i, r)
        order = np.ascontiguousarray(order, np.int64)
        cell_ptr = np.ascontiguousarray(cell_ptr, np.int64)
        rc = self.L.mfo_textbook_epoch_mt(_fp(P), _fp(Q), P.shape[1], _ip(u), _ip(i), _fp(r), _lp(order),
                                          _lp(cell_ptr), n_rounds, n_cells, lr, lam, threads)
        if rc != 0:
            raise RuntimeError("mfo_textbook_epoch_mt failed")

    def sse(self, P, Q, u, i, r):
        u, i, r = self._chk(P, Q, u, i, r)
        return float(self.L.mfo_sse(_fp(P), _fp(Q), P.shape[1], _ip(u), _ip(i), _fp(r), u.size))

    def rmse(self, P, Q, u, i, r):
        u, i, r = self._chk(P, Q, u, i, r)
        return float(self.L.mfo_rmse(_fp(P), _fp(Q), P.shape[1], _ip(u), _ip(i), _fp(r), u.size))

    def predict(self, P, Q, u, i):
        u = np.ascontiguousarray(u, np.int32)
        i = np.ascontiguousarray(i, np.int32)
        out = np.empty(u.size, np.float32)
        self.L.mfo_predict(_fp(P), _fp(Q), P.shape[1], _ip(u), _ip(i), _fp(out), u.size)
        return out

    def check_block_schedule(self, u, i, U, I, order, cell_ptr, n_rounds, n_cells):
        u = np.ascontiguousarray(u, np.int32)
        i = np.ascontiguousarray(i, np.int32)
        order = np.ascontiguousarray(order, np.int64)
        cell_ptr = np.ascontiguousarray(cell_ptr, np.int64)
        return int(self.L.mfo_check_block_schedule(_ip(u), _ip(i), u.size, U, I, _lp(order), _lp(cell_ptr),
                                                   n_rounds, n_cells))
