"""Multi-GPU sharding of the candidate index space + the all-gather of final tracks.

The reference parallelises with ``dask.delayed(run_tracks)`` — one *process per
simulated year*, results pickled back (`util/compute.py:223-230`).  Storms are
independent, so here the **candidate index space of a year** is sharded instead:
one process per GPU (``torch.distributed``; backend "nccl" is RCCL on ROCm, "gloo"
on CPU for the tests), every rank holding a replica of the read-only fields.

Determinism across world sizes comes from two rules:
  * a candidate's random stream is keyed by its *global* index (tcr_seed.hip), and
  * a round of ``world * C`` candidates is split into contiguous blocks in rank
    order, so concatenating per-rank survivors in rank order *is* candidate order —
    the order in which the reference's sequential loop would have met them.

The only data-path collective is the all-gather of survivor records (fixed-size
rows of 9*n_steps fp64 = 26 kB); counts and ``n_seeds`` travel as tiny
all-gathers / all-reduces.  xGMI is point-to-point, so one large all-gather per
round (not one per storm or per variable) is the shape that suits it.
"""
import os

import torch
import torch.distributed as dist


def forced():
    """TCR_FORCE_COLLECTIVES=1: a one-rank run takes the world > 1 branches — a one-rank process group is created and every
    collective below goes through the backend (RCCL on a GPU box) instead of short-circuiting.  This is how the code RCCL
    and its stream ordering run on an eight-GPU node is executed on the one GPU a test box has."""
    return os.environ.get('TCR_FORCE_COLLECTIVES', '0') == '1'


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment (no-op when WORLD_SIZE is 1 or unset, unless
    TCR_FORCE_COLLECTIVES=1).  Returns (rank, world, local_rank).  Fails loudly when the backend is RCCL and the node has
    fewer GPUs than local ranks (RCCL cannot put two ranks on one device; TCR_DIST_BACKEND=gloo can, for functional tests)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if (world > 1 or forced()) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if world == 1:
            os.environ.setdefault('MASTER_PORT', str(_free_port()))
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            # TCR_DIST_BACKEND=gloo lets several ranks share one GPU for functional tests
            backend = os.environ.get('TCR_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            n_local = int(os.environ.get('LOCAL_WORLD_SIZE', str(world)))
            if torch.cuda.device_count() < n_local:
                raise RuntimeError('%d local ranks but %d GPU(s) visible: RCCL needs one device per rank '
                                   '(TCR_DIST_BACKEND=gloo lets ranks share a GPU for functional tests)'
                                   % (n_local, torch.cuda.device_count()))
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def collective():
    """True when collectives go through the process group (several ranks, or TCR_FORCE_COLLECTIVES=1 with one)."""
    return dist.is_initialized() and (dist.get_world_size() > 1 or forced())


def backend_name():
    b = dist.get_backend() if dist.is_initialized() else 'none'
    return {'nccl': 'nccl = RCCL'}.get(b, b)


def local_device(local):
    """GPU index of this rank: LOCAL_RANK, folded onto the visible devices when ranks
    deliberately share GPUs (functional tests on a 1-GPU box)."""
    n = torch.cuda.device_count()
    return int(local) % n if n else 0


def world():
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def round_block(round_idx, per_rank, rank_=None, world_=None):
    """First global candidate index of this rank's block in round ``round_idx``."""
    r = rank() if rank_ is None else rank_
    w = world() if world_ is None else world_
    return (int(round_idx) * w + r) * int(per_rank)


def allgather_counts(count):
    """count: int64 tensor [1] on the compute device -> list of python ints per rank."""
    if not collective():
        return [int(count.item())]
    buf = torch.empty(world(), dtype=count.dtype, device=count.device)
    dist.all_gather_into_tensor(buf, count.reshape(1))
    return [int(x) for x in buf.tolist()]


def allgather_ints(vec):
    """vec: int64 tensor [k] on the compute device -> list (per rank) of lists of k python ints.  This is the
    accept loop's one host synchronisation per round."""
    if not collective():
        return [[int(x) for x in vec.tolist()]]
    buf = torch.empty(world() * vec.numel(), dtype=vec.dtype, device=vec.device)
    dist.all_gather_into_tensor(buf, vec.contiguous())
    flat = buf.tolist()
    k = vec.numel()
    return [[int(x) for x in flat[r * k:(r + 1) * k]] for r in range(world())]


def allgather_rows(rows, count, counts=None, async_op=False, concat=True):
    """All-gather variable-length row blocks.

    rows  : [cap, width] tensor whose first ``count`` rows are valid (cap may differ
            per rank; only ``max(counts)`` rows are sent)
    Returns (gathered [sum(counts), width] in rank order, counts) — or, with
    ``async_op=True``, (work handle, finish()) where finish() returns that pair.
    ``concat=False`` skips the compaction copy: the first element is then the list of
    per-rank views into the receive buffer.
    """
    counts = allgather_counts(count) if counts is None else counts
    w = world()
    if not collective():
        out = rows[:counts[0]] if concat else [rows[:counts[0]]]
        return ((None, lambda: (out, counts)) if async_op else (out, counts))
    m = max(counts)
    width = rows.shape[1]
    if rows.shape[0] < m:       # pad so every rank contributes the same shape
        pad = torch.empty(m, width, dtype=rows.dtype, device=rows.device)
        pad[:rows.shape[0]] = rows
        rows = pad
    send = rows[:m].contiguous()
    recv = torch.empty(w * m, width, dtype=rows.dtype, device=rows.device)
    work = dist.all_gather_into_tensor(recv, send, async_op=async_op) if m > 0 else None

    def finish():
        if work is not None and async_op:
            work.wait()
        parts = [recv[r * m: r * m + counts[r]] for r in range(w)]
        return (torch.cat(parts, dim=0) if concat else parts), counts
    return (work, finish) if async_op else finish()


class DeferredRowGather:
    """All-gather of a stream of row batches without a host sync in the producer's step.

    Batch k's row count (a device scalar) is all-gathered asynchronously when the batch is submitted;
    it is read on the host only when batch k + lag is submitted — by then that tiny collective has long
    finished — and only then is the row all-gather of batch k launched (on the collective backend's own
    stream).  The producer fills `buffer(k)` (one of lag + 2 rotating [cap, width] buffers); before a
    buffer is handed out again the gather that read it is waited for on the *caller's stream*, not on
    the host.  `on_rows(parts, counts)` (optional) receives the per-rank views of each finished gather.
    """

    def __init__(self, cap, width, device, lag, dtype=torch.float64, on_rows=None):
        import collections
        self.cap, self.lag, self.depth = int(cap), int(lag), int(lag) + 2
        self.w = world()
        self.packed = [torch.empty(self.cap, width, dtype=dtype, device=device) for _ in range(self.depth)]
        self.cnt = [torch.zeros(self.w, dtype=torch.int64, device=device) for _ in range(self.depth)]
        self.row_work = [None] * self.depth
        self.inflight = collections.deque()
        self.launched = collections.deque()     # slots with a row gather in flight, oldest first
        self.side = torch.cuda.Stream(device=device) if torch.device(device).type == 'cuda' else None
        self.rows_gathered = 0
        self.rows_clipped = 0
        self.on_rows = on_rows
        self.k = 0

    def buffer(self):
        """The buffer for the next batch; waits (stream-side) for the gather that last read it."""
        slot = self.k % self.depth
        self._retire(slot)
        return self.packed[slot]

    def submit(self, count):
        """The buffer returned by the last buffer() call holds `count` (device int64 [1]) valid rows."""
        slot = self.k % self.depth
        if collective():
            wc = dist.all_gather_into_tensor(self.cnt[slot], count.reshape(1), async_op=True)
        else:
            self.cnt[slot].copy_(count.reshape(1))
            wc = None
        # the producer's stream has written the rows and the count up to here: the side stream that later reads
        # them must wait for exactly this point (with world == 1 there is no collective to order them)
        ev = None
        if self.side is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.packed[slot].device))
        self.inflight.append((slot, wc, ev))
        self.k += 1
        if len(self.inflight) > self.lag:
            self._launch_oldest()

    def _launch_oldest(self):
        slot, wc, ev = self.inflight.popleft()
        ctx = torch.cuda.stream(self.side) if self.side is not None else _null()
        with ctx:
            if ev is not None:
                self.side.wait_event(ev)
            if wc is not None:
                wc.wait()
            counts = [int(c) for c in self.cnt[slot].tolist()]
            self.rows_clipped += sum(max(0, c - self.cap) for c in counts)
            counts = [min(c, self.cap) for c in counts]
            _, fin = allgather_rows(self.packed[slot], None, counts=counts, async_op=True, concat=False)
        self.row_work[slot] = (fin, counts)
        self.launched.append(slot)

    def _retire(self, slot):
        if self.row_work[slot] is not None:
            fin, counts = self.row_work[slot]
            parts, _ = fin()                                   # stream-side wait, no copy
            self.rows_gathered += sum(counts)
            if self.on_rows is not None:
                self.on_rows(parts, counts)
            self.row_work[slot] = None
            self.launched.remove(slot)

    def drain(self):
        while self.inflight:
            self._launch_oldest()
        while self.launched:
            self._retire(self.launched[0])


class Local:
    """The collectives of the accept loop for a rank that works a whole year on its own (years sharded over the ranks:
    `compute.run_downscaling`): world 1, rank 0, nothing leaves the rank."""

    @staticmethod
    def world():
        return 1

    @staticmethod
    def rank():
        return 0

    @staticmethod
    def round_block(round_idx, per_rank, rank_=None, world_=None):
        return int(round_idx) * int(per_rank)

    @staticmethod
    def allgather_ints(vec):
        return [[int(x) for x in vec.tolist()]]

    @staticmethod
    def allgather_rows(rows, count, counts=None, async_op=False, concat=True):
        return rows[:counts[0]], counts

    @staticmethod
    def allreduce_sum_(t):
        return t


def allgather_year_blocks(block, n_mine, device):
    """All-gather of the final tracks when the YEARS are sharded over the ranks: `block` [k_max, ...] holds this rank's
    `n_mine` finished years (k_max = the most any rank has, so every rank contributes the same shape).  Returns
    (gathered [world, k_max, ...], counts per rank) — one collective over the whole result, device to device."""
    w = world()
    cnt = torch.tensor([int(n_mine)], dtype=torch.int64, device=device)
    if collective() and use_lib_collectives() and block.is_cuda:
        comm = lib_comm(device)
        counts = [int(x) for x in comm.allgather_counts(cnt).tolist()]
        return comm.allgather(block.contiguous()), counts
    counts = allgather_counts(cnt)
    if not collective():
        return block.unsqueeze(0), counts
    recv = torch.empty((w,) + tuple(block.shape), dtype=block.dtype, device=block.device)
    dist.all_gather_into_tensor(recv.view(w * block.shape[0], -1), block.contiguous().view(block.shape[0], -1))
    return recv, counts


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def allreduce_sum_(t):
    if collective():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def barrier():
    if collective():
        dist.barrier()


def max_over_ranks(x, device):
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    if collective():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x, device):
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    if collective():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


class LibComm:
    """The library's own RCCL communicator (include/tcrisk_hip.h, multi-GPU section: tcr_comm_* / tcr_allgather_* /
    tcr_concat_rows_dev) — the exchange of final tracks behind the C ABI, for hosts that do not bring torch.distributed.
    Here it is bootstrapped from the process group that is already up: rank 0 makes the 128-byte id, everybody receives it
    through `broadcast_object_list` (any backend); a host without torch hands the id over by file / MPI instead.
    One rank per GPU (RCCL); every method is collective and asynchronous on `stream` (None: torch's current stream)."""

    def __init__(self, engine, rank_=None, world_=None, unique_id=None):
        import ctypes as C
        from . import _lib
        self.L = _lib.lib()
        self.eng = engine
        self.rank = rank() if rank_ is None else int(rank_)
        self.world = world() if world_ is None else int(world_)
        if unique_id is None:
            ident = [None]
            if self.rank == 0:
                buf = (C.c_uint8 * _lib.TCR_COMM_ID_BYTES)()
                if self.L.tcr_comm_unique_id(buf) != 0:
                    raise _lib.TcrError(self.L.tcr_last_error(None).decode())
                ident = [bytes(buf)]
            if dist.is_initialized() and dist.get_world_size() > 1:
                dist.broadcast_object_list(ident, src=0)
            unique_id = ident[0]
        idb = (C.c_uint8 * _lib.TCR_COMM_ID_BYTES).from_buffer_copy(unique_id)
        h = C.c_void_p()
        if self.L.tcr_comm_create(engine.h, idb, self.rank, self.world, C.byref(h)) != 0:
            raise _lib.TcrError(self.L.tcr_last_error(engine.h).decode())
        self.h = h

    @classmethod
    def for_device(cls, device, rank_=None, world_=None):
        """A communicator on a context of its own (the exchange needs no staged fields): `device` is the GPU index."""
        import ctypes as C
        from . import _lib
        L = _lib.lib()

        class _Ctx:
            pass
        holder = _Ctx()
        h = C.c_void_p()
        if L.tcr_ctx_create(int(device), C.byref(h)) != 0:
            raise _lib.TcrError(L.tcr_last_error(None).decode())
        holder.h = h
        comm = cls(holder, rank_, world_)
        comm._own_ctx = True
        return comm

    def _st(self, stream):
        import ctypes as C
        return C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)

    def allgather(self, t, stream=None):
        """t: contiguous device tensor, the same shape on every rank -> [world, *t.shape] (tcr_allgather_dev)."""
        assert t.is_contiguous()
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        self._chk(self.L.tcr_allgather_dev(self.h, t.data_ptr(), out.data_ptr(), t.numel() * t.element_size(), self._st(stream)))
        return out

    def _chk(self, rc):
        if rc != 0:
            from . import _lib
            raise _lib.TcrError(self.L.tcr_last_error(self.eng.h).decode())

    def allgather_counts(self, count, stream=None):
        """count: int64 [1] device tensor -> int64 [world] device tensor."""
        out = torch.empty(self.world, dtype=torch.int64, device=count.device)
        self._chk(self.L.tcr_allgather_counts_dev(self.h, count.data_ptr(), out.data_ptr(), self._st(stream)))
        return out

    def allgather_rows(self, rows, counts_dev, out_cap=None, stream=None):
        """rows: [cap, width] float64 (the same cap on every rank) with this rank's valid rows in front; counts_dev: int64 [world]
        (allgather_counts).  Returns (packed [out_cap, width] in rank order, n_out int64 [1] on the device) — no host sync."""
        cap, width = rows.shape
        out_cap = self.world * cap if out_cap is None else int(out_cap)
        recv = torch.empty(self.world * cap, width, dtype=torch.float64, device=rows.device)
        out = torch.empty(out_cap, width, dtype=torch.float64, device=rows.device)
        n_out = torch.zeros(1, dtype=torch.int64, device=rows.device)
        st = self._st(stream)
        self._chk(self.L.tcr_allgather_rows_dev(self.h, rows.data_ptr(), cap, width, recv.data_ptr(), st))
        self._chk(self.L.tcr_concat_rows_dev(self.eng.h, self.world, recv.data_ptr(), counts_dev.data_ptr(), cap, width, out.data_ptr(), out_cap,
                                             n_out.data_ptr(), st))
        return out, n_out

    def allreduce_sum_(self, t, stream=None):
        assert t.dtype == torch.int64 and t.is_contiguous()
        self._chk(self.L.tcr_allreduce_sum_i64_dev(self.h, t.data_ptr(), t.numel(), self._st(stream)))
        return t

    def close(self):
        if getattr(self, 'h', None) is not None:
            self.L.tcr_comm_destroy(self.h)
            self.h = None
            if getattr(self, '_own_ctx', False):
                self.L.tcr_ctx_destroy(self.eng.h)


_LIB_COMM = {}


def use_lib_collectives():
    """TCR_COLLECTIVES=lib: the data-path exchange (the all-gather of final tracks) goes through the library's own RCCL
    communicator (tcr_comm_* of include/tcrisk_hip.h) instead of torch.distributed's; the process group only bootstraps it."""
    return os.environ.get('TCR_COLLECTIVES', '') == 'lib'


def lib_comm(device):
    """The process's LibComm on `device` (created on first use; collective: every rank reaches this together)."""
    idx = torch.device(device).index if not isinstance(device, int) else device
    if idx not in _LIB_COMM:
        _LIB_COMM[idx] = LibComm.for_device(idx)
    return _LIB_COMM[idx]
