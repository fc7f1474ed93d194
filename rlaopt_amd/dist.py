"""Process-group helpers for the row-sharded operator (one process per GPU).

The reference moves every operand and result through ``torch.multiprocessing`` queues with
host staging (rlaopt/linops/base.py:157-176,202-267).  Here the shards live in separate
ranks of a ``torch.distributed`` group (backend "nccl" = RCCL over xGMI on the GPUs, "gloo"
in the CPU tests) and exchange device-resident buffers:

* ROW-mode product  (linops/distributed.py:40-45):  concat      -> all-gather of row blocks
* COLUMN-mode / transpose product (:46-50, :81-86): sum         -> all-reduce
* block inner products of PCG / Nystrom:             one fused all-reduce per use

The row partition is the reference's ``torch.chunk(arange(n), p)`` (kernels/base.py:297-302):
ceil(n/p)-sized blocks, possibly FEWER than p of them; ranks beyond the last block own no
rows.
"""

from __future__ import annotations

import torch
import torch.distributed as td


import os


def initialized() -> bool:
    return td.is_available() and td.is_initialized()


def spmd_active(group=None) -> bool:
    """True when operators should use the one-rank-per-shard layout: a process group with more
    than one rank, or any group when ``RK_FORCE_SPMD=1`` (lets a single-rank NCCL group drive
    the collective code path on a one-GPU box)."""
    if not initialized():
        return False
    return td.get_world_size(group) > 1 or os.environ.get("RK_FORCE_SPMD", "0") == "1"


def _host_staged(t: torch.Tensor, group) -> bool:
    # gloo (the CPU test backend) cannot move device buffers: stage through the host.
    # nccl (= RCCL) takes device pointers directly - the production path.
    return t.is_cuda and td.get_backend(group) == "gloo"


def world(group=None) -> tuple[int, int]:
    """(rank, world_size); (0, 1) when no process group exists."""
    if not initialized():
        return 0, 1
    return td.get_rank(group), td.get_world_size(group)


def row_chunks(n: int, parts: int) -> list[tuple[int, int]]:
    """[lo, hi) of every block of ``torch.chunk(arange(n), parts)``."""
    if n <= 0 or parts <= 0:
        raise ValueError("n and parts must be positive")
    step = -(-n // parts)
    return [(lo, min(lo + step, n)) for lo in range(0, n, step)]


def my_rows(n: int, group=None) -> tuple[int, int]:
    """This rank's [lo, hi) (empty ``(n, n)`` for a rank past the last block)."""
    rank, size = world(group)
    chunks = row_chunks(n, size)
    return chunks[rank] if rank < len(chunks) else (n, n)


# What went over the wire since the last ``reset_wire_stats`` (ring accounting: an all-reduce of b bytes
# sends 2 (p - 1) / p b per rank, a reduce-scatter or an all-gather of b bytes in total (p - 1) / p b).
wire_stats = {"collectives": 0, "bytes_sent_per_rank": 0.0, "by_kind": {}}


def reset_wire_stats() -> None:
    wire_stats.update(collectives=0, bytes_sent_per_rank=0.0, by_kind={})


def _account(kind: str, nbytes: int, factor: float, group) -> None:
    size = td.get_world_size(group)
    wire_stats["collectives"] += 1
    wire_stats["bytes_sent_per_rank"] += factor * (size - 1) / max(size, 1) * nbytes
    wire_stats["by_kind"][kind] = wire_stats["by_kind"].get(kind, 0) + 1


# ---- optional: the collectives through the C ABI's own RCCL transport (RK_COMM=abi) ---------------------------
# Default: torch.distributed moves the buffers (backend "nccl" = RCCL).  With RK_COMM=abi and device buffers the
# three collectives go through rk_comm_* (include/rk.h) instead - the transport a maintainer binding the C ABI into
# the reference would use; the communicator is created once per process (rank 0's unique id travels through the
# torch.distributed store).
_abi_comm = {"handle": None, "size": 0}


def _use_abi_comm(t: torch.Tensor, group) -> bool:
    return (os.environ.get("RK_COMM", "") == "abi" and t.is_cuda and t.dtype in (torch.float32, torch.float64)
            and group is None and td.get_backend(group) == "nccl")


def _abi(group=None):
    import ctypes

    from . import _lib

    if _abi_comm["handle"] is None:
        lib = _lib.load()
        rank, size = td.get_rank(group), td.get_world_size(group)
        box = [None]
        if rank == 0:
            buf = ctypes.create_string_buffer(128)
            _lib.check(lib.rk_comm_unique_id(buf), "rk_comm_unique_id")
            box = [bytes(buf.raw)]
        td.broadcast_object_list(box, src=0, group=group)
        h = ctypes.c_void_p()
        _lib.check(lib.rk_comm_create(ctypes.byref(h), size, rank, box[0]), "rk_comm_create")
        _abi_comm.update(handle=h, size=size)
    return _abi_comm["handle"]


def _abi_dt(t):
    return 0 if t.dtype == torch.float32 else 1


def _abi_stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def shutdown_abi_comm() -> None:
    """Destroy the RK_COMM=abi communicator (tests; process exit does not need it)."""
    from . import _lib

    if _abi_comm["handle"] is not None:
        _lib.check(_lib.load().rk_comm_destroy(_abi_comm["handle"]), "rk_comm_destroy")
        _abi_comm.update(handle=None, size=0)


def all_reduce_sum(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum over ranks (no-op without a group)."""
    if spmd_active(group):
        _account("all_reduce", t.numel() * t.element_size(), 2.0, group)
        if _use_abi_comm(t, group) and t.is_contiguous():
            from . import _lib

            _lib.check(_lib.load().rk_comm_allreduce_sum(_abi(group), t.data_ptr(), t.numel(), _abi_dt(t), _abi_stream(t)),
                       "rk_comm_allreduce_sum")
        elif _host_staged(t, group):
            h = t.cpu()
            td.all_reduce(h, op=td.ReduceOp.SUM, group=group)
            t.copy_(h)
        else:
            td.all_reduce(t, op=td.ReduceOp.SUM, group=group)
    return t


def chunk_sizes(n: int, group=None) -> list[int]:
    """Rows owned by each rank under ``row_chunks`` (0 for ranks past the last block)."""
    _, size = world(group)
    ch = row_chunks(n, size)
    return [hi - lo for lo, hi in ch] + [0] * (size - len(ch))


def all_gather_rows(local: torch.Tensor, sizes, group=None) -> torch.Tensor:
    """Concatenate per-rank row blocks (``sizes[r]`` rows on rank r; an int ``n`` means the
    ``row_chunks`` partition of n rows).  Blocks are padded to the tallest one so that a
    single ``all_gather_into_tensor`` moves everything."""
    rank, size = world(group)
    if not spmd_active(group):
        return local
    if isinstance(sizes, int):
        sizes = chunk_sizes(sizes, group)
    if len(sizes) != size:
        sizes = list(sizes) + [0] * (size - len(sizes))
    squeeze = local.ndim == 1
    loc = local[:, None] if squeeze else local
    if loc.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank} holds {loc.shape[0]} rows, partition says {sizes[rank]}")
    k = loc.shape[1]
    step = max(sizes)
    if loc.shape[0] != step:
        pad = torch.zeros((step, k), dtype=loc.dtype, device=loc.device)
        pad[: loc.shape[0]] = loc
        loc = pad
    _account("all_gather", size * step * k * loc.element_size(), 1.0, group)
    if _host_staged(loc, group):
        host = torch.empty((size * step, k), dtype=loc.dtype)
        td.all_gather_into_tensor(host, loc.cpu().contiguous(), group=group)
        out = host.to(loc.device)
    elif _use_abi_comm(loc, group):
        from . import _lib

        out = torch.empty((size * step, k), dtype=loc.dtype, device=loc.device)
        src = loc.contiguous()
        _lib.check(_lib.load().rk_comm_allgather(_abi(group), src.data_ptr(), out.data_ptr(), src.numel(), _abi_dt(src),
                                                 _abi_stream(src)), "rk_comm_allgather")
    else:
        out = torch.empty((size * step, k), dtype=loc.dtype, device=loc.device)
        td.all_gather_into_tensor(out, loc.contiguous(), group=group)
    total = sum(sizes)
    if _tail_only(sizes, step):  # padding only behind the last non-empty block: one slice
        out = out[:total]
    else:
        out = torch.cat([out[r * step : r * step + sizes[r]] for r in range(size)], dim=0)
    return out[:, 0] if squeeze else out


def reduce_scatter_rows(partial: torch.Tensor, n: int, tail: torch.Tensor | None = None, group=None):
    """Sum the ranks' full-length partials (n x k each) and leave every rank with ITS rows of the sum
    (``row_chunks`` partition), plus - when ``tail`` is given - the sum of the ranks' small ``tail``
    blocks on every rank, all in ONE reduce-scatter: the wire buffer has one slot per rank, slot r =
    [rows of rank r, padded to the tallest block | tail], in the dtype of ``partial`` (``rk_spmd_pack``:
    one launch).  Against an all-reduce of the whole partial this sends half the bytes per element and -
    for float32 partials that used to ride in a float64 buffer next to a float64 tail - a quarter.
    Returns (rows, tail_sum) with tail_sum in float64 (None without a tail); no-op without a group."""
    from .ops import get_backend

    rank, size = world(group)
    if not spmd_active(group):
        return partial, (None if tail is None else tail.double())
    sizes = chunk_sizes(n, group)
    step = max(sizes)
    k = partial.shape[1]
    t = 0 if tail is None else tail.numel()
    slot = step * k + t
    buf = get_backend().spmd_pack(partial, n, step, size, tail)  # (size, slot)
    _account("reduce_scatter", size * slot * buf.element_size(), 1.0, group)
    if _host_staged(buf, group):
        host = torch.empty(slot, dtype=buf.dtype)
        td.reduce_scatter_tensor(host, buf.cpu().reshape(-1), op=td.ReduceOp.SUM, group=group)
        out = host.to(buf.device)
    elif _use_abi_comm(buf, group):
        from . import _lib

        out = torch.empty(slot, dtype=buf.dtype, device=buf.device)
        _lib.check(_lib.load().rk_comm_reduce_scatter_sum(_abi(group), buf.data_ptr(), out.data_ptr(), slot, _abi_dt(buf),
                                                          _abi_stream(buf)), "rk_comm_reduce_scatter_sum")
    else:
        out = torch.empty(slot, dtype=buf.dtype, device=buf.device)
        td.reduce_scatter_tensor(out, buf.view(-1), op=td.ReduceOp.SUM, group=group)
    rows = out[: sizes[rank] * k].view(sizes[rank], k)
    return rows, (None if tail is None else out[step * k :].double().view(tail.shape))


def _first_short(sizes, step):
    for i, s in enumerate(sizes):
        if s != step:
            return i
    return len(sizes)


def _tail_only(sizes, step):
    """True when every block before the first short one is full and all later ones empty."""
    i = _first_short(sizes, step)
    return all(s == 0 for s in sizes[i + 1 :])


def broadcast_object(obj, src: int = 0, group=None):
    """Small Python object from ``src`` to every rank (e.g. an RNG seed)."""
    if not spmd_active(group):
        return obj
    box = [obj]
    td.broadcast_object_list(box, src=src, group=group)
    return box[0]


_seed_stream = {"gen": None}


def reset_seed_stream() -> None:
    """Forget the shared seed stream: the next ``shared_randint`` re-bases it on rank 0's default
    generator (called at the start of every ``solve``, so ``torch.manual_seed`` before a solve decides
    its draws)."""
    _seed_stream["gen"] = None


def shared_randint(count: int = 1, group=None) -> list[int]:
    """``count`` integers in [0, 2**62) that every rank agrees on (sketch seeds, SAP's block draw).

    Single process: drawn from torch's default CPU generator.  One rank per GPU: the first call after
    ``reset_seed_stream`` broadcasts ONE value drawn from rank 0's default generator; it seeds a private
    generator that all ranks then advance in lockstep (every rank makes the same draws in the same
    order), so a draw costs no collective - a ``broadcast_object_list`` is two broadcasts and two host
    round trips, and a SAP step made three of them."""
    if not spmd_active(group):
        return [int(v) for v in torch.randint(0, 2**62, (count,), dtype=torch.int64)]
    if _seed_stream["gen"] is None:
        base = int(broadcast_object(int(torch.randint(0, 2**62, (1,), dtype=torch.int64)), group=group))
        gen = torch.Generator()
        gen.manual_seed(base)
        _seed_stream["gen"] = gen
    return [int(v) for v in torch.randint(0, 2**62, (count,), dtype=torch.int64, generator=_seed_stream["gen"])]
