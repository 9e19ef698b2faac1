"""Multi-GPU plumbing for the hot path: independent tuples shard over ranks, no data-path
collective; the single collective is the metric gather at the end (the reference's only
explicit collective is the 1-element ``all_reduce`` of the validation loss, ``train.py:102-106``).

One process per GPU.  The collective itself is the LIBRARY's (``e2emv_metric_allgather`` / ``e2emv_metric_allreduce`` over
RCCL/xGMI, ``csrc/comm.hip``: a plain-C or numpy host does the 8-GPU run with the same calls, bootstrap through a file) whenever
the ranks sit on GPUs of their own - ``LibraryComm``; ``torch.distributed`` is the launcher / rendezvous and stays the
collective of the CPU tests ("gloo") and the fallback when the library communicator cannot be made (ranks sharing one GPU, no
librccl).  A few KB per evaluation: latency-bound, ring bandwidth irrelevant.
"""
import ctypes
import os
import sys
import threading

import numpy as np
import torch


class LibraryComm:
    """The library-owned RCCL communicator of this rank (include/e2emv.h: e2emv_comm_*).  Create it on EVERY rank at the same
    point (collective).  `id_bytes`: the 128-byte id from `LibraryComm.unique_id()` of rank 0, handed over by the caller - or
    `id_file`: a path every rank can see (rank 0 writes it)."""

    def __init__(self, rank, world, device=None, id_bytes=None, id_file=None, timeout_s=60.0):
        from . import _lib
        self.ctx = _lib.context(device)
        self.rank, self.world = int(rank), int(world)
        h = ctypes.c_void_p()
        if id_file is not None:
            self.ctx.call("e2emv_comm_init_file", os.fsencode(id_file), self.rank, self.world, float(timeout_s), ctypes.byref(h))
        else:
            buf = (ctypes.c_char * 128).from_buffer_copy(bytes(id_bytes))
            self.ctx.call("e2emv_comm_init", buf, self.rank, self.world, ctypes.byref(h))
        self.h = h

    @staticmethod
    def unique_id(device=None):
        from . import _lib
        buf = (ctypes.c_char * 128)()
        _lib.context(device).call("e2emv_comm_unique_id", buf)
        return bytes(buf)

    def allgather(self, local):
        """local: float32 device tensor [n] (same n on every rank) -> [world, n] on the same device."""
        from . import _lib
        local = local.contiguous().float()
        out = torch.empty(self.world, local.numel(), dtype=torch.float32, device=local.device)
        self.ctx.call("e2emv_metric_allgather", self.h, _lib.ptr(local), local.numel(), _lib.ptr(out), _lib.stream_ptr(local.device))
        return out

    def allreduce_(self, buf, op="sum"):
        """In place over the ranks; op: "sum" | "max" | "min" (train.py:102-106 is `sum`)."""
        from . import _lib
        assert buf.dtype == torch.float32 and buf.is_contiguous()
        self.ctx.call("e2emv_metric_allreduce", self.h, _lib.ptr(buf), buf.numel(), {"sum": 0, "max": 1, "min": 2}[op], _lib.stream_ptr(buf.device))
        return buf

    def close(self):
        if getattr(self, "h", None):
            self.ctx.call("e2emv_comm_destroy", self.h)
            self.h = None


_lib_comm = {}
last_collective = "none"  # what the last gather_pair_errors call ran on (bench.py reports it)


def library_comm(device, group=None, timeout_s=90.0):
    """The process's LibraryComm for `device`, made on first use from the torch.distributed job it runs in (rank 0's id travels by
    broadcast_object_list), or None when it cannot be: no GPU of its own per rank (RCCL refuses two ranks on one device), no
    librccl, E2EMV_COLLECTIVE=torch.  Every rank takes the same decision (a MIN all-reduce of the outcome), so nobody waits in a
    collective the others skipped.  ncclCommInitRank has no timeout of its own: it runs in a thread that is given `timeout_s`."""
    import torch.distributed as dist
    key = (str(device), id(group))
    if key in _lib_comm:
        return _lib_comm[key]
    comm, ok = None, 0
    want = (os.environ.get("E2EMV_COLLECTIVE", "lib") != "torch" and device is not None and torch.device(device).type == "cuda"
            and dist.get_backend(group) == "nccl")
    try:
        if want:
            rank, world = dist.get_rank(group), dist.get_world_size(group)
            # one GPU per rank?  (bench.py binds rank -> LOCAL_RANK % device_count: on a 1-GPU box two ranks share the device)
            idx = torch.tensor([torch.cuda.current_device() if torch.device(device).index is None else torch.device(device).index], device=device)
            idxs = [torch.zeros_like(idx) for _ in range(world)]
            dist.all_gather(idxs, idx, group=group)
            if len({int(i.item()) for i in idxs}) == world:
                box = [LibraryComm.unique_id(device) if rank == 0 else None]
                dist.broadcast_object_list(box, src=0, group=group)
                res = {}

                def work():
                    try:
                        res["c"] = LibraryComm(rank, world, device, id_bytes=box[0])
                    except Exception as e:  # noqa: BLE001
                        res["e"] = e
                t = threading.Thread(target=work, daemon=True)
                t.start()
                t.join(timeout_s)
                if "c" in res:
                    comm, ok = res["c"], 1
                else:
                    print(f"[e2emv] library collective unavailable on rank {rank}: {res.get('e', 'timeout')} - torch.distributed takes the metric gather", file=sys.stderr)
    except Exception as e:  # noqa: BLE001
        print(f"[e2emv] library collective unavailable: {e} - torch.distributed takes the metric gather", file=sys.stderr)
    flag = torch.tensor([ok], dtype=torch.int32, device=device if (device is not None and dist.get_backend(group) == "nccl") else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if int(flag.item()) == 0:
        comm = None
    _lib_comm[key] = comm
    return comm


def shard_range(n_items, rank, world):
    """Contiguous block of item indices owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def close_library_comms():
    """Destroys the library communicators this process made (before torch.distributed.destroy_process_group / interpreter exit: an RCCL
    communicator left alive at exit is torn down in an unspecified order)."""
    for key, comm in list(_lib_comm.items()):
        try:
            if comm is not None:
                torch.cuda.synchronize()
                comm.close()
        except Exception:  # noqa: BLE001
            pass
        _lib_comm.pop(key, None)


import atexit  # noqa: E402

atexit.register(close_library_comms)


def gather_pair_errors(local_errors, device=None, group=None):
    """All-gather variable-length per-pair pose errors (degrees); every rank gets the full array
    in rank order.  inf marks a pair whose pose could not be computed (``eval_pairs.py:259``)."""
    import torch.distributed as dist
    e = torch.as_tensor(np.asarray(local_errors, dtype=np.float32))
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return e.numpy().astype(np.float64)
    if device is not None:
        e = e.to(device)
    world = dist.get_world_size(group)
    comm = library_comm(device, group) if device is not None else None
    global last_collective
    last_collective = "torch.distributed (" + dist.get_backend(group) + ")"
    if comm is not None:  # the library's own RCCL collective: counts first (n = 1), then the padded values
        last_collective = "library: e2emv_metric_allgather over RCCL (csrc/comm.hip)"
        sizes = comm.allgather(torch.tensor([float(e.numel())], dtype=torch.float32, device=e.device)).view(-1).round().long().tolist()
        m = max(sizes)
        pad = torch.full((max(m, 1),), float("inf"), dtype=torch.float32, device=e.device)
        pad[: e.numel()] = e
        allv = comm.allgather(pad).cpu().numpy()
        return np.concatenate([allv[r, :sizes[r]] for r in range(world)]).astype(np.float64)
    n = torch.tensor([e.numel()], dtype=torch.int64, device=e.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    m = int(max(int(s.item()) for s in sizes))
    pad = torch.full((m,), float("inf"), dtype=e.dtype, device=e.device)
    pad[: e.numel()] = e
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return np.concatenate([b[: int(s.item())].cpu().numpy() for b, s in zip(bufs, sizes)]).astype(np.float64)


def reduce_max_seconds(seconds, device=None, group=None):
    """MAX over ranks of a wall-clock interval (bench.py's timing rule)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
