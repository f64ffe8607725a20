"""Batch sharding across the GPUs of one node: one process per GPU, clips are independent end to end, the only
exchange step is one all-gather of the emitted codes (RCCL over xGMI via torch.distributed, backend "nccl").

The reference has no inference-time collective; this is the BASELINE.json config-4 path.
Codes are 10-bit values carried as int64 by the API; they travel as int16 (4x fewer bytes on the links).
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(total: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, near-equal split of `total` clips; earlier ranks take the remainder."""
    base, rem = divmod(total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def narrow_codes(codes: torch.Tensor) -> torch.Tensor:
    """int64 (values < 2^15) -> int16; HIP kernel on device tensors, torch cast for host tensors (gloo tests)."""
    if codes.is_cuda:
        from . import _native
        lib = _native.load()          # codes < 2^15: ESC.__init__ refuses codebook_size > 32768
        out = torch.empty(codes.shape, dtype=torch.int16, device=codes.device)
        c = codes.contiguous()
        with torch.cuda.device(codes.device):
            _native.check(lib.escx_codes_narrow(ctypes.c_void_p(c.data_ptr()), ctypes.c_void_p(out.data_ptr()), c.numel(),
                                                ctypes.c_void_p(torch.cuda.current_stream(codes.device).cuda_stream)))
        return out
    return codes.to(torch.int16)


def widen_codes(codes16: torch.Tensor) -> torch.Tensor:
    if codes16.is_cuda:
        from . import _native
        lib = _native.load()
        out = torch.empty(codes16.shape, dtype=torch.int64, device=codes16.device)
        c = codes16.contiguous()
        with torch.cuda.device(codes16.device):
            _native.check(lib.escx_codes_widen(ctypes.c_void_p(c.data_ptr()), ctypes.c_void_p(out.data_ptr()), c.numel(),
                                               ctypes.c_void_p(torch.cuda.current_stream(codes16.device).cuda_stream)))
        return out
    return codes16.to(torch.int64)


class PendingCodes:
    """An all-gather of codes in flight (`all_gather_codes(..., async_op=True)`): the collective runs on the communicator's own stream,
    the caller's stream carries on (the local decode does not depend on the other ranks' codes).  `wait()` orders the caller's CURRENT
    stream behind the collective (no host synchronisation) and returns the gathered int64 codes."""

    def __init__(self, finish):
        self._finish, self._out = finish, None

    def wait(self) -> torch.Tensor:
        if self._finish is not None:
            self._out, self._finish = self._finish(), None
        return self._out


def all_gather_codes(codes_local: torch.Tensor, group: Optional[dist.ProcessGroup] = None, force: bool = False,
                     counts: Optional[Sequence[int]] = None, async_op: bool = False):
    """(B_local, S, G, T) int64 on every rank -> (sum B_local, S, G, T) int64 on every rank, rank order.

    No host synchronisation: shard sizes are never exchanged.  Equal shards (the default) use one
    all_gather_into_tensor -- a single direct collective: the payload is <= 0.2 MB per rank, latency-bound, nowhere near
    the per-link xGMI bandwidth.  Ragged shards pass `counts` (every rank can compute them with `shard_bounds`) and
    gather zero-padded shards.  Collectives move bytes: RCCL/gloo have no int16 type, so the payload is viewed as uint8."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return PendingCodes(lambda: codes_local) if async_op else codes_local
    world = dist.get_world_size(group)
    small = narrow_codes(codes_local)
    if counts is None or len(set(counts)) == 1:
        out = torch.empty((world * small.shape[0],) + tuple(small.shape[1:]), dtype=torch.int16, device=small.device)
        src8 = small.contiguous().view(torch.uint8)
        # the collective variant is chosen from the backend UP FRONT (never by catching an exception around a collective: ranks
        # that disagree on the fallback would deadlock)
        if str(dist.get_backend(group)).lower() == "nccl":
            if async_op:            # RCCL runs on the process group's stream behind an event of ours; our stream waits only in wait()
                work = dist.all_gather_into_tensor(out.view(torch.uint8), src8, group=group, async_op=True)

                def finish(work=work, out=out, keep=(small, src8)):
                    work.wait()     # stream-orders the current stream behind the collective, does not block the host
                    return widen_codes(out)
                return PendingCodes(finish)
            dist.all_gather_into_tensor(out.view(torch.uint8), src8, group=group)
        else:
            parts = [torch.empty_like(src8) for _ in range(world)]
            dist.all_gather(parts, src8, group=group)
            out = torch.cat(parts, dim=0).view(torch.int16)
        res = widen_codes(out)
        return PendingCodes(lambda: res) if async_op else res
    assert len(counts) == world and counts[dist.get_rank(group)] == small.shape[0]
    mx = max(counts)
    pad = torch.zeros((mx,) + tuple(small.shape[1:]), dtype=torch.int16, device=small.device)
    pad[: small.shape[0]] = small
    parts = [torch.empty_like(pad.view(torch.uint8)) for _ in range(world)]
    dist.all_gather(parts, pad.view(torch.uint8), group=group)
    res = widen_codes(torch.cat([p.view(torch.int16)[:c] for p, c in zip(parts, counts)], dim=0))
    return PendingCodes(lambda: res) if async_op else res


def _mapped_path(cdll) -> Optional[str]:
    """Real path of a loaded shared object, None when it cannot be told.  First choice: dladdr() on a symbol resolved THROUGH THE HANDLE,
    which names the object that handle really is even when two RCCL instances are mapped (torch's bundled one and /opt/rocm's); the
    /proc/self/maps scan by basename is the fallback."""
    import os
    try:
        class _DlInfo(ctypes.Structure):
            _fields_ = [("dli_fname", ctypes.c_char_p), ("dli_fbase", ctypes.c_void_p), ("dli_sname", ctypes.c_char_p), ("dli_saddr", ctypes.c_void_p)]
        libdl = ctypes.CDLL(None)
        libdl.dladdr.argtypes = [ctypes.c_void_p, ctypes.POINTER(_DlInfo)]
        libdl.dladdr.restype = ctypes.c_int
        for sym in ("ncclCommInitRank", "ncclGetUniqueId"):
            fn = getattr(cdll, sym, None)
            if fn is None:
                continue
            info = _DlInfo()
            if libdl.dladdr(ctypes.cast(fn, ctypes.c_void_p), ctypes.byref(info)) and info.dli_fname:
                return os.path.realpath(info.dli_fname.decode())
    except Exception:
        pass
    name = os.path.basename(getattr(cdll, "_name", "") or "")
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.split(None, 5)[-1].strip() if line.count(" ") >= 5 else ""
                if path.startswith("/") and os.path.basename(path).startswith(name.split(".so")[0]):
                    return os.path.realpath(path)
    except OSError:
        pass
    return None


class AbiCodesGather:
    """The exchange step through the C ABI: `escx_allgather_codes` (csrc/collective.cpp) on an RCCL communicator of its own.

    torch.distributed does not hand out its ncclComm_t, so the communicator is created here on the SAME RCCL instance libescx resolves
    (`escx_set_rccl_library` is pointed at the library this class loaded): rank 0 draws the unique id, the id travels over the existing
    torch.distributed group (bootstrap only), every rank calls ncclCommInitRank.  Equal shards only - it is one ncclAllGather."""

    _NAMES = ("librccl.so.1", "librccl.so")

    def __init__(self, model, device, group: Optional[dist.ProcessGroup] = None):
        import os
        from . import _native
        self.model = model                       # the native handle is re-fetched per call: load_state_dict / .to() rebuild it
        self.device = torch.device(device)
        self.lib, _ = model._handle(self.device)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.rccl, tried = None, []
        for name in self._NAMES + (os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so.1"):
            try:
                self.rccl = ctypes.CDLL(name)
                rc = self.lib.escx_set_rccl_library(name.encode())
                if rc == _native.ESCX_ERR_STATE:
                    # libescx resolved its RCCL earlier (a previous collective).  A communicator from another RCCL instance must never reach
                    # it, so the instance this class loaded has to be that very library: compare what the dynamic loader mapped.
                    mine, theirs = _mapped_path(self.rccl), (self.lib.escx_last_error() or b"").decode("utf-8", "replace")
                    if mine is None or mine not in theirs:
                        raise RuntimeError(f"libescx already bound an RCCL library ({theirs!r}); refusing to create a communicator on {name} ({mine})")
                else:
                    _native.check(rc)
                break
            except OSError as e:
                tried.append(f"{name}: {e}")
        if self.rccl is None:
            raise ImportError("no RCCL library found: " + "; ".join(tried))

        class UniqueId(ctypes.Structure):
            _fields_ = [("internal", ctypes.c_char * 128)]
        uid = UniqueId()
        if self.rank == 0 and self.rccl.ncclGetUniqueId(ctypes.byref(uid)) != 0:
            raise RuntimeError("ncclGetUniqueId failed")
        box = [bytes(bytearray(uid)) if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ctypes.memmove(ctypes.byref(uid), box[0], 128)
        self.comm = ctypes.c_void_p()
        self.rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
        with torch.cuda.device(self.device):
            rc = self.rccl.ncclCommInitRank(ctypes.byref(self.comm), self.world, uid, self.rank)
        if rc != 0:
            raise RuntimeError(f"ncclCommInitRank failed with {rc}")

    def __call__(self, codes_local: torch.Tensor) -> torch.Tensor:
        from . import _native
        if codes_local.dtype != torch.int64:
            raise TypeError(f"codes must be int64 (escx_allgather_codes reads n int64 values), got {codes_local.dtype}")
        if not codes_local.is_cuda or codes_local.device != self.device:
            raise ValueError(f"codes must live on {self.device}, got {codes_local.device}")
        c = codes_local.contiguous()
        out = torch.empty((self.world * c.shape[0],) + tuple(c.shape[1:]), dtype=torch.int64, device=c.device)
        _, hd = self.model._handle(self.device)
        with torch.cuda.device(self.device):
            _native.check(self.lib.escx_allgather_codes(hd, ctypes.c_void_p(c.data_ptr()), c.numel(), ctypes.c_void_p(out.data_ptr()), self.world,
                                                        self.comm, ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        return out

    def start(self, codes_local: torch.Tensor) -> "PendingCodes":
        """The same exchange on a side stream: ordered behind what the caller's stream has produced so far, joined in `wait()`.
        `__call__` and `start()` share the handle's single staging buffer: do not interleave them on different streams without `wait()` in between."""
        cur = torch.cuda.current_stream(self.device)
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            out = self(codes_local)
            codes_local.record_stream(self._side)
        done = torch.cuda.Event()
        done.record(self._side)

        def finish(out=out, done=done):
            consumer = torch.cuda.current_stream(self.device)
            consumer.wait_event(done)
            out.record_stream(consumer)        # `out` was allocated under the side stream; tell the caching allocator who reads it now
            return out
        return PendingCodes(finish)

    def close(self):
        if self.comm:
            torch.cuda.synchronize(self.device)
            self.rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
            self.rccl.ncclCommDestroy(self.comm)
            self.comm = ctypes.c_void_p()


def encode_sharded(model, x_local: torch.Tensor, num_streams: int = 6, group: Optional[dist.ProcessGroup] = None,
                   counts: Optional[Sequence[int]] = None, async_op: bool = False):
    """Each rank encodes its own clips; returns (codes of the WHOLE batch on every rank, local codes, feat_shape).
    async_op=True: the first element is a `PendingCodes` - decode the local shard first, `wait()` when the whole batch is needed, and the
    exchange over xGMI hides under the decode."""
    codes_local, shape = model.encode(x_local, num_streams)
    return all_gather_codes(codes_local, group, counts=counts, async_op=async_op), codes_local, shape


# ---- data-parallel training: gradient exchange ----------------------------------------------------------------------------
def all_reduce_gradients(grad_flat: torch.Tensor, group: Optional[dist.ProcessGroup] = None, bucket_mb: float = 16.0, force: bool = False) -> torch.Tensor:
    """Mean of the flat gradient buffer over the ranks, in place (what DDP does for `accel.backward`, /root/reference/scripts/
    trainer_no_adv.py:115).  The training backward leaves ALL gradients in one flat fp32 buffer (35 MB for ESC-Base, 62 MB for Large),
    so the exchange is a handful of large ring all-reduces instead of one per parameter: buckets of `bucket_mb` (xGMI is point-to-point,
    ~153 GB/s per link: 16 MB keeps every ring step well above the latency floor) issued asynchronously and awaited together."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):      # force: run the collective even with one rank (tests)
        return grad_flat
    world = dist.get_world_size(group)
    n = grad_flat.numel()
    per = max(1, int(bucket_mb * (1 << 20) / 4))
    avg = str(dist.get_backend(group)).lower() == "nccl"          # RCCL averages in the collective; gloo sums
    works = []
    for lo in range(0, n, per):
        chunk = grad_flat[lo:min(n, lo + per)]
        works.append(dist.all_reduce(chunk, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=group, async_op=True))
    for w in works:
        w.wait()
    if not avg:
        grad_flat.div_(world)
    return grad_flat
