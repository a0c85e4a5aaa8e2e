"""DSGD over the GPUs of one node: user x item blocks, the item-factor blocks
rotated between ranks (BASELINE.json north_star).

One process per GPU.  Rank g owns the P rows of its users for the whole run and
a schedule for each of the G item partitions (item i -> partition i % G, row
i // G of that partition's Q block).  Sub-epoch s: rank g trains partition
(g + s) % G against the Q block it currently holds -- the G active (user shard,
item partition) pairs share no user and no item -- then every rank passes its
block to rank g-1 and receives rank g+1's (a ring shift: one point-to-point
message per rank over xGMI, RCCL send/recv through torch.distributed; no
all-to-all, no data-path all-reduce).  After G sub-epochs every block is home.

The compute backend is an object with part_rows/part_init_q/part_train/part_sse;
the product backend (HipBackend) drives libmfsgd.so.  Tests inject a CPU
stand-in to exercise this file's rotation logic under gloo.
"""
import numpy as np


class HipBackend:
    """libmfsgd.so on one MI355X; Q blocks are torch CUDA tensors (device memory
    and streams are torch's: plumbing only)."""

    def __init__(self, trainer, torch_device):
        import torch

        self.t = trainer
        self.torch = torch
        self.device = torch_device

    def new_block(self, rows, kp):
        return self.torch.zeros((rows, kp), dtype=self.torch.float32, device=self.device)

    def load_block(self, block, host_array):
        block[: host_array.shape[0]].copy_(self.torch.from_numpy(host_array))

    def block_to_host(self, block):
        return block.cpu().numpy()

    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def part_rows(self, part):
        return self.t.part_rows(part)

    def part_init_q(self, part, seed, u_total):
        return self.t.part_init_q(part, seed, u_total)

    def part_train(self, part, block):
        self.t.part_train(part, block.data_ptr(), self._stream())

    def part_sse(self, part, block):
        return self.t.part_sse(part, block.data_ptr(), self._stream())

    def synchronize(self):
        self.torch.cuda.synchronize(self.device)


class NativeDSGD:
    """The product path for N GPUs: the ring under the C-ABI (csrc/dsgd.cpp, mfsgd_dsgd_*): RCCL
    ncclSend / ncclRecv in a group on a communication stream, ordered against the training stream
    with events.  This class is the thin binding a Java / C++ host would write (INTEGRATION.md
    section 5); `trainer` is a MatrixFactorizationSGD created with n_parts = world * m that already
    has its ratings and its P seed.  `unique_id` (128 bytes from NativeDSGD.unique_id() on one
    rank) reaches the other ranks by whatever transport the host has."""

    def __init__(self, trainer, rank, world, unique_id):
        import ctypes as C

        from . import _lib

        self._C = C
        self._lib = _lib.load_library()
        _lib.share_torch_rccl()
        self.t = trainer
        self.rank, self.world = int(rank), int(world)
        self.m = trainer.n_parts // self.world
        self._d = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), 128)
        rc = self._lib.mfsgd_dsgd_create(trainer._handle(), self.rank, self.world, buf, C.byref(self._d))
        if rc != 0:
            from .trainer import MfsgdError

            raise MfsgdError(rc, self._lib.mfsgd_dsgd_last_error(None).decode())

    @staticmethod
    def unique_id():
        import ctypes as C

        from . import _lib
        from .trainer import MfsgdError

        lib = _lib.load_library()
        _lib.share_torch_rccl()
        buf = C.create_string_buffer(128)
        rc = lib.mfsgd_dsgd_unique_id(buf)
        if rc != 0:
            raise MfsgdError(rc, lib.mfsgd_dsgd_last_error(None).decode())
        return buf.raw

    def _check(self, rc):
        if rc != 0:
            from .trainer import MfsgdError

            raise MfsgdError(rc, self._lib.mfsgd_dsgd_last_error(self._d).decode())

    def close(self):
        if getattr(self, "_d", None):
            self._lib.mfsgd_dsgd_destroy(self._d)
            self._d = self._C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def init_q(self, seed, u_total):
        self._check(self._lib.mfsgd_dsgd_init_q(self._d, int(seed), int(u_total)))

    def train(self, epochs, rmse=True):
        C = self._C
        out = np.zeros(int(epochs), np.float64)
        self._check(self._lib.mfsgd_dsgd_train(self._d, int(epochs), out.ctypes.data_as(C.POINTER(C.c_double)) if rmse else None))
        return out if rmse else None

    def train_timed(self, epochs):
        ms = self._C.c_double()
        self._check(self._lib.mfsgd_dsgd_train_timed(self._d, int(epochs), self._C.byref(ms)))
        return ms.value

    def rmse(self):
        out = self._C.c_double()
        self._check(self._lib.mfsgd_dsgd_rmse(self._d, self._C.byref(out)))
        return out.value

    def stats(self):
        """Counters of this rank's ring: sub-epoch trainings, of those with the recovery point, re-run as round
        launches after a failed residency check, bytes sent."""
        out = np.zeros(4, np.int64)
        self._check(self._lib.mfsgd_dsgd_stats(self._d, out.ctypes.data_as(self._C.POINTER(self._C.c_int64))))
        return dict(trained=int(out[0]), checked=int(out[1]), rerun_as_round_launches=int(out[2]), bytes_sent=int(out[3]))

    def allreduce(self, a, b, op="sum"):
        C = self._C
        v = np.array([a, b], np.float64)
        self._check(self._lib.mfsgd_dsgd_allreduce(self._d, v.ctypes.data_as(C.POINTER(C.c_double)), 0 if op == "sum" else 1))
        return float(v[0]), float(v[1])

    def home_blocks(self):
        """{partition: rows x k host copy of its Q block} of the group held (home between epochs)."""
        C = self._C
        out = {}
        for j in range(self.m):
            part, rows = C.c_int32(), C.c_int32()
            self._check(self._lib.mfsgd_dsgd_get_q(self._d, j, C.byref(part), C.byref(rows), None))
            blk = np.empty((rows.value, self.t.k), np.float32)
            self._check(self._lib.mfsgd_dsgd_get_q(self._d, j, C.byref(part), C.byref(rows),
                                                   blk.ctypes.data_as(C.POINTER(C.c_float))))
            out[part.value] = blk
        return out


class TorchDistRing:
    """Ring shift over torch.distributed (backend "nccl" = RCCL on ROCm, or gloo on CPU)."""

    def __init__(self, dist, rank, world, group=None):
        self.dist, self.rank, self.world, self.group = dist, rank, world, group  # group: the P2P group (None = default)

    def shift(self, send_block, recv_block):
        """send to rank-1, receive from rank+1."""
        if self.world == 1:
            recv_block.copy_(send_block)
            return
        d = self.dist
        if send_block.is_cuda and d.get_backend(self.group) == "gloo":
            # rehearsal mode (several ranks sharing one GPU, where RCCL cannot run):
            # stage through host memory.  The production path is the branch below.
            hs, hr = send_block.cpu(), recv_block.cpu()
            for req in d.batch_isend_irecv([d.P2POp(d.isend, hs, (self.rank - 1) % self.world),
                                            d.P2POp(d.irecv, hr, (self.rank + 1) % self.world)]):
                req.wait()
            recv_block.copy_(hr)
            return
        ops = [d.P2POp(d.isend, send_block, (self.rank - 1) % self.world, group=self.group),
               d.P2POp(d.irecv, recv_block, (self.rank + 1) % self.world, group=self.group)]
        for req in d.batch_isend_irecv(ops):
            req.wait()

    def _dev(self, device):
        return "cpu" if self.world > 1 and self.dist.get_backend() == "gloo" else device

    def sum_f64(self, values, torch, device):
        device = self._dev(device)
        t = torch.tensor(values, dtype=torch.float64, device=device)
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().tolist()

    def max_f64(self, value, torch, device):
        device = self._dev(device)
        t = torch.tensor([value], dtype=torch.float64, device=device)
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.cpu()[0])

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()


class DSGD:
    """The rotation schedule.  `backend` computes, `ring` moves blocks.

    Items are cut into world * m partitions; a rank holds m of them at a time (its
    "group": partitions group*m .. group*m + m - 1), trains them ONE AFTER ANOTHER -- they
    share the rank's users, so they cannot run concurrently -- and passes the whole group
    along the ring.  m = 1 is the normal case; m > 1 only makes the schedules smaller."""

    def __init__(self, backend, ring, rank, world, n_items, kp, seed, u_total, nnz_local, parts_per_rank=1):
        self.b, self.ring, self.rank, self.world = backend, ring, rank, world
        self.seed, self.u_total, self.nnz_local = seed, u_total, nnz_local
        self.m = int(parts_per_rank)
        self.n_parts = world * self.m
        # every block buffer can hold any partition (a planned item map gives them different row counts)
        self.max_rows = max(backend.part_rows(p) for p in range(self.n_parts))
        self.kp = kp
        # two buffers of m blocks each: the group being trained and the landing zone of the next
        self.cur = backend.new_block(self.m * self.max_rows, kp)
        self.nxt = backend.new_block(self.m * self.max_rows, kp)
        self.group = rank  # group currently held
        for j, part in enumerate(self.parts()):
            backend.load_block(self.block(j), backend.part_init_q(part, seed, u_total))

    def parts(self):
        return [self.group * self.m + j for j in range(self.m)]

    def block(self, j, buf=None):
        buf = self.cur if buf is None else buf
        return buf[j * self.max_rows:(j + 1) * self.max_rows]

    def _rotate(self):
        self.ring.shift(self.cur, self.nxt)
        self.cur, self.nxt = self.nxt, self.cur
        self.group = (self.group + 1) % self.world

    def epoch(self):
        for _ in range(self.world):
            for j, part in enumerate(self.parts()):
                self.b.part_train(part, self.block(j))
            self._rotate()

    def sse(self):
        """Sum of squared errors of this rank's ratings (one read-only rotation)."""
        total = 0.0
        for _ in range(self.world):
            for j, part in enumerate(self.parts()):
                total += self.b.part_sse(part, self.block(j))
            self._rotate()
        return total

    def home_blocks(self):
        """{partition: host copy of its Q block} -- valid between epochs."""
        out = {}
        for j, part in enumerate(self.parts()):
            rows = self.b.part_rows(part)
            out[part] = self.b.block_to_host(self.block(j))[:rows]
        return out

    def home_block(self):
        part = self.parts()[0]
        return part, self.home_blocks()[part]


def assemble_q(blocks, n_items, k, world):
    """Rebuilds the dense I x k matrix from {partition: block} (tests)."""
    Q = np.zeros((n_items, k), np.float32)
    for part, blk in blocks.items():
        idx = np.arange(part, n_items, world)
        Q[idx] = blk[: idx.size, :k]
    return Q
