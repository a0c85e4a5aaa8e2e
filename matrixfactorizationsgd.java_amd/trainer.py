"""Python mirror of the Java surface ``MatrixFactorizationSGD`` (train/predict).

The reference repository contains no source (/root/reference/README.md:1-2), so
the surface is the one SURVEY.md section 8b derives from BASELINE.json:
``new MatrixFactorizationSGD(users, items, k, lr, lambda, seed)``,
``double[] train(int[] u, int[] i, float[] r, int epochs)``,
``float predict(int u, int i)`` / ``float[] predict(int[] u, int[] i)``,
``close()``.  Every method is a thin call into the C-ABI (include/mfsgd.h).
"""
import ctypes as C

import numpy as np

from . import _lib


class MfsgdError(RuntimeError):
    """A C-ABI call returned a non-zero status (the JNI shim throws
    RuntimeException in the same place)."""

    def __init__(self, code, message):
        super().__init__(f"mfsgd error {code}: {message}")
        self.code = code


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


def dsgd_plan(deg_user, deg_item, n_parts):
    """The global DSGD partitioner (mfsgd_dsgd_plan): (user_begin[n_parts + 1], item_part[n_items])."""
    du = np.ascontiguousarray(deg_user, np.int64)
    di = np.ascontiguousarray(deg_item, np.int64)
    ub = np.empty(int(n_parts) + 1, np.int32)
    ip = np.empty(di.size, np.int32)
    rc = _lib.load_library().mfsgd_dsgd_plan(_p(du, C.c_int64), _p(di, C.c_int64), du.size, di.size, int(n_parts),
                                             _p(ub, C.c_int32), _p(ip, C.c_int32))
    if rc != 0:
        raise MfsgdError(rc, "mfsgd_dsgd_plan: bad argument")
    return ub, ip


def dsgd_plan_ex(deg_user, deg_item, world, parts_per_rank=1, k=64, chain_crit=0.0):
    """mfsgd_dsgd_plan_ex: (user_begin[world + 1], item_part[n_items] over world * parts_per_rank partitions, info);
    chain_crit = 0: plain LPT (what a ring wants), > 0: chain-aware packing (include/mfsgd.h);
    info = dict(sum_max_chain, critical_items, sequential_parts, threshold)."""
    du = np.ascontiguousarray(deg_user, np.int64)
    di = np.ascontiguousarray(deg_item, np.int64)
    ub = np.empty(int(world) + 1, np.int32)
    ip = np.empty(di.size, np.int32)
    info = np.zeros(4, np.int64)
    rc = _lib.load_library().mfsgd_dsgd_plan_ex(_p(du, C.c_int64), _p(di, C.c_int64), du.size, di.size, int(world),
                                                int(parts_per_rank), int(k), float(chain_crit), _p(ub, C.c_int32), _p(ip, C.c_int32),
                                                _p(info, C.c_int64))
    if rc != 0:
        raise MfsgdError(rc, "mfsgd_dsgd_plan_ex: bad argument")
    return ub, ip, dict(sum_max_chain=int(info[0]), critical_items=int(info[1]), sequential_parts=int(info[2]),
                        threshold=int(info[3]))


class MatrixFactorizationSGD:
    def __init__(self, users, items, k, lr, lam, seed, *, device=0, blocks=0, waves=0,
                 n_parts=0, host_threads=0, flags=0):
        self._lib = _lib.load_library()
        self._h = C.c_void_p()
        self.users, self.items, self.k = int(users), int(items), int(k)
        self.lr, self.lam, self.seed = float(lr), float(lam), int(seed)
        cfg = _lib.Config(n_users=self.users, n_items=self.items, k=self.k, lr=self.lr,
                          lambda_=self.lam, device=device, blocks=blocks, waves=waves,
                          n_parts=n_parts, host_threads=host_threads, flags=flags)
        rc = self._lib.mfsgd_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            raise MfsgdError(rc, self._lib.mfsgd_last_error(None).decode())
        self.n_parts = max(1, int(n_parts))
        self._initialised = False

    # -- plumbing ---------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise MfsgdError(rc, self._lib.mfsgd_last_error(self._h).decode())

    def _handle(self):
        if not self._h:
            raise MfsgdError(-5, "handle is closed")
        return self._h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mfsgd_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- ratings / factors ------------------------------------------------------
    def set_ratings(self, u, i, r):
        u, i, r = _i32(u), _i32(i), _f32(r)
        if not (u.shape == i.shape == r.shape and u.ndim == 1):
            raise ValueError("u, i, r must be 1-d arrays of equal length")
        # the library itself recognises the same triples again (exact: every byte is hashed) and keeps the
        # schedules, so there is nothing to remember here
        self._check(self._lib.mfsgd_set_ratings(self._handle(), _p(u, C.c_int32), _p(i, C.c_int32),
                                                _p(r, C.c_float), u.size))

    def init_factors(self, seed=None):
        self._check(self._lib.mfsgd_init_factors(self._handle(), self.seed if seed is None else int(seed)))
        self._initialised = True

    def set_factors(self, P, Q=None):
        P = _f32(P)
        if P.shape != (self.users, self.k):
            raise ValueError("P must be users x k")
        qp = None
        if Q is not None:
            Q = _f32(Q)
            if Q.shape != (self.items, self.k):
                raise ValueError("Q must be items x k")
            qp = _p(Q, C.c_float)
        self._check(self._lib.mfsgd_set_factors(self._handle(), _p(P, C.c_float), qp))
        self._initialised = True

    def get_factors(self):
        P = np.empty((self.users, self.k), np.float32)
        if self.n_parts > 1:
            self._check(self._lib.mfsgd_get_factors(self._handle(), _p(P, C.c_float), None))
            return P, None
        Q = np.empty((self.items, self.k), np.float32)
        self._check(self._lib.mfsgd_get_factors(self._handle(), _p(P, C.c_float), _p(Q, C.c_float)))
        return P, Q

    def save_factors(self, path):
        rc = self._lib.mfsgd_save_factors(self._handle(), str(path).encode())
        if rc != 0:
            raise MfsgdError(rc, self._lib.mfsgd_io_last_error().decode() or self._lib.mfsgd_last_error(self._h).decode())

    def load_factors(self, path):
        rc = self._lib.mfsgd_load_factors(self._handle(), str(path).encode())
        if rc != 0:
            raise MfsgdError(rc, self._lib.mfsgd_io_last_error().decode() or self._lib.mfsgd_last_error(self._h).decode())
        self._initialised = True

    # -- the Java surface -------------------------------------------------------
    def train(self, u, i, r, epochs, *, rmse=True):
        """Runs `epochs` SGD passes over the ratings; returns the RMSE after each
        epoch (float64 array), as the Java ``double[] train(...)`` does."""
        self.set_ratings(u, i, r)
        if not self._initialised:
            self.init_factors()
        return self.fit(epochs, rmse=rmse)

    def fit(self, epochs, *, rmse=True):
        out = np.zeros(int(epochs), np.float64)
        self._check(self._lib.mfsgd_train(self._handle(), int(epochs),
                                          _p(out, C.c_double) if rmse else None))
        return out if rmse else None

    def train_timed(self, epochs):
        """(elapsed device milliseconds, kernel launches) for `epochs` passes."""
        ms = C.c_double()
        launches = C.c_int64()
        self._check(self._lib.mfsgd_train_timed(self._handle(), int(epochs), C.byref(ms), C.byref(launches)))
        return ms.value, launches.value

    def rmse(self):
        out = C.c_double()
        self._check(self._lib.mfsgd_rmse(self._handle(), C.byref(out)))
        return out.value

    def predict(self, u, i):
        scalar = np.isscalar(u) and np.isscalar(i)
        uu, ii = _i32(np.atleast_1d(u)), _i32(np.atleast_1d(i))
        if uu.shape != ii.shape or uu.ndim != 1:
            raise ValueError("u and i must have the same 1-d shape")
        out = np.empty(uu.size, np.float32)
        self._check(self._lib.mfsgd_predict(self._handle(), _p(uu, C.c_int32), _p(ii, C.c_int32),
                                            _p(out, C.c_float), uu.size))
        return float(out[0]) if scalar else out

    def recommend(self, users, topn):
        """(items, scores), each [len(users), topn]: best items per user, best first."""
        uu = _i32(np.atleast_1d(users))
        items = np.empty((uu.size, int(topn)), np.int32)
        scores = np.empty((uu.size, int(topn)), np.float32)
        self._check(self._lib.mfsgd_recommend(self._handle(), _p(uu, C.c_int32), uu.size, int(topn),
                                              _p(items, C.c_int32), _p(scores, C.c_float)))
        return items, scores

    # -- schedule introspection (tests, bench) -----------------------------------
    def schedule_info(self, part=0):
        info = _lib.ScheduleInfo()
        self._check(self._lib.mfsgd_get_schedule_info(self._handle(), int(part), C.byref(info)))
        return info.as_dict()

    def order(self, part=0):
        """Canonical sequential order of a partition and its cell boundaries."""
        info = self.schedule_info(part)
        order = np.empty(info["nnz"], np.int64)
        cell_ptr = np.empty(info["rounds"] * info["blocks"] + 1, np.int64)
        self._check(self._lib.mfsgd_get_order(self._handle(), int(part), _p(order, C.c_int64),
                                              _p(cell_ptr, C.c_int64)))
        return order, cell_ptr

    def debug_schedule(self, part=0):
        """Device-facing schedule arrays (chunk descriptors, rows, subs, entries) as uint32 arrays."""
        n = [C.c_int64() for _ in range(4)]
        self._check(self._lib.mfsgd_debug_schedule_sizes(self._handle(), int(part), *[C.byref(x) for x in n]))
        cells = np.zeros((n[0].value, 8), np.uint32)
        rows = np.zeros(n[1].value, np.uint32)
        subs = np.zeros((n[2].value, 2), np.uint32)
        entries = np.zeros((n[3].value, 4), np.uint32)
        self._check(self._lib.mfsgd_debug_get_schedule(self._handle(), int(part), _p(cells, C.c_uint32),
                                                       _p(rows, C.c_uint32), _p(subs, C.c_uint32),
                                                       _p(entries, C.c_uint32)))
        return cells, rows, subs, entries

    def debug_epoch_profile(self):
        """[workgroups, 7] shader cycles per phase of one persistent epoch (diagnostic)."""
        info = self.schedule_info()
        out = np.zeros(info["blocks"] * 16, np.uint64)
        n = C.c_int32()
        self._check(self._lib.mfsgd_debug_epoch_profile(self._handle(), _p(out, C.c_uint64), C.byref(n)))
        out = out[: n.value * 16].reshape(n.value, 16)
        self.last_slowest_cell = out[:, 7]  # longest single "ratings" phase per workgroup
        self.last_slowest_pass = out[:, 8:15]  # the seven phases of the pass that contained it
        return out[:, :7]

    def debug_counters(self):
        out = np.zeros(4, np.int64)
        self._check(self._lib.mfsgd_debug_counters(self._handle(), _p(out, C.c_int64)))
        return dict(not_resident=int(out[0]), persistent_parts=int(out[1]), graphs=int(out[2]), schedule_builds=int(out[3]))

    def debug_occupy(self, milliseconds):
        """Holds every CU's LDS for a while on a side stream (diagnostic; asynchronous)."""
        self._check(self._lib.mfsgd_debug_occupy(self._handle(), int(milliseconds)))

    def debug_round_stamps(self, rnd):
        """[blocks, 6] stamps of one training round (diagnostic): 4 shader-clock phase
        stamps, then the 100 MHz constant clock at start and end."""
        info = self.schedule_info()
        B, W = info["blocks"], info["waves"]
        out = np.zeros(B * (6 + W * W * 4), np.uint64)
        self._check(self._lib.mfsgd_debug_round_stamps(self._handle(), 0, int(rnd), _p(out, C.c_uint64)))
        self.last_loop_timers = out[B * 6:].reshape(B, W, W, 4)  # [block, wave, sub-round, (gen cyc, run cyc, gen steps, run steps)]
        return out[:B * 6].reshape(B, 6)

    # -- DSGD building blocks (n_parts > 1); see dsgd.py ---------------------------
    def set_item_partition(self, item_part):
        """Item -> partition map (mfsgd_dsgd_plan's), before set_ratings; None = i % n_parts."""
        if item_part is None:
            self._check(self._lib.mfsgd_set_item_partition(self._handle(), None))
            return
        ip = _i32(item_part)
        if ip.shape != (self.items,):
            raise ValueError("item_part must have one entry per item")
        self._check(self._lib.mfsgd_set_item_partition(self._handle(), _p(ip, C.c_int32)))

    def item_partition(self):
        part = np.empty(self.items, np.int32)
        row = np.empty(self.items, np.int32)
        self._check(self._lib.mfsgd_get_item_partition(self._handle(), _p(part, C.c_int32), _p(row, C.c_int32)))
        return part, row

    def part_rows(self, part):
        rows = C.c_int32()
        self._check(self._lib.mfsgd_part_rows(self._handle(), int(part), C.byref(rows)))
        return rows.value

    def part_init_q(self, part, seed, u_total):
        kp = 4 * self._group_lanes()
        buf = np.zeros((self.part_rows(part), kp), np.float32)
        self._check(self._lib.mfsgd_part_init_q(self._handle(), int(part), int(seed), int(u_total),
                                                _p(buf, C.c_float)))
        return buf

    def init_p_offset(self, seed, u_offset):
        self._check(self._lib.mfsgd_init_p_offset(self._handle(), int(seed), int(u_offset)))
        self._initialised = True

    def part_train(self, part, q_block_ptr, stream_ptr=0):
        self._check(self._lib.mfsgd_part_train(self._handle(), int(part), C.c_void_p(q_block_ptr),
                                               C.c_void_p(stream_ptr)))

    def part_sse(self, part, q_block_ptr, stream_ptr=0):
        out = C.c_double()
        self._check(self._lib.mfsgd_part_sse(self._handle(), int(part), C.c_void_p(q_block_ptr),
                                             C.c_void_p(stream_ptr), C.byref(out)))
        return out.value

    def _group_lanes(self):
        need, L = (self.k + 3) // 4, 1
        while L < need:
            L <<= 1
        return L

    @property
    def kp(self):
        return 4 * self._group_lanes()
