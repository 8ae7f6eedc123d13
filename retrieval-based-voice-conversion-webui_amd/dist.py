"""Multi-GPU plumbing for batched conversion: one process per GPU; utterances are
independent, so the only collective is ONE broadcast of the packed IVF index over RCCL/xGMI at
start-up; afterwards ranks never talk.  The reference has no inference-side collective at all
(its only NCCL use is training DDP, infer/modules/train/train.py:205-216)."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def torchrun_argv(script: str, script_args: List[str], nproc: int, master_port: Optional[int] = None) -> List[str]:
    """The command that runs ``script`` as ``nproc`` ranks of ONE node, one per GPU (what a batch driver or ``bench.py --gpus N``
    exec's when it was started without a launcher): ``python -m torch.distributed.run`` with the rendezvous pinned to 127.0.0.1
    (container hostnames need not resolve) on a free port."""
    import socket
    import sys

    if master_port is None:
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", 0))
            master_port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % int(nproc),
            "--master-addr", "127.0.0.1", "--master-port", str(int(master_port)), script] + list(script_args)


def ranks_agree(t: torch.Tensor, src: int = 0, group=None) -> bool:
    """True on every rank iff every rank's ``t`` is bit-identical to ``src``'s (one broadcast + one MIN all-reduce).
    Used after the index broadcast: the same seeded queries must return the same (distances, ids) everywhere."""
    ref = t.clone()
    dist.broadcast(ref, src=src, group=group)
    same = (ref.view(torch.uint8) == t.view(torch.uint8)).all() if t.numel() else torch.tensor(True, device=t.device)
    ok = same.to(torch.int32).reshape(1).clone()
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    return bool(ok.item())


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split of ``n_items`` utterances: the first ``n_items % world`` ranks take one extra."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_bytes(payload: Optional[torch.Tensor], src: int = 0, device=None, group=None) -> torch.Tensor:
    """Broadcast a uint8 tensor whose size only ``src`` knows.  Works on any backend (nccl=RCCL on
    the GPUs, gloo on CPU for the tests).  Returns the payload on every rank."""
    rank = dist.get_rank(group)
    if device is None:
        device = payload.device if payload is not None else torch.device("cpu")
    n = torch.zeros(1, dtype=torch.int64, device=device)
    if rank == src:
        if payload is None or payload.dtype != torch.uint8:
            raise ValueError("src rank must supply a uint8 tensor")
        n[0] = payload.numel()
    dist.broadcast(n, src=src, group=group)
    if rank == src:
        buf = payload.to(device).contiguous()
    else:
        buf = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    dist.broadcast(buf, src=src, group=group)
    return buf


def broadcast_index(index, src: int = 0, device=None, group=None):
    """Replicate an ``IVFFlatHIP`` held by ``src`` onto every rank's GPU with one broadcast of its blob."""
    from .ivf import IVFFlatHIP

    rank = dist.get_rank(group)
    blob = index.blob() if rank == src else None
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    buf = broadcast_bytes(blob, src=src, device=device, group=group)
    if rank == src:
        return index
    return IVFFlatHIP.from_blob(buf)


MAX_GATHER_ELEMENTS = 1 << 16  # per tensor / array inside a gathered result (a 10 s waveform has 480 000)


def _largest_array(obj) -> int:
    """Element count of the largest tensor / ndarray inside ``obj`` (tuples, lists and dicts are walked)."""
    import numpy as np

    if isinstance(obj, torch.Tensor):
        return int(obj.numel())
    if isinstance(obj, np.ndarray):
        return int(obj.size)
    if isinstance(obj, dict):
        return max([_largest_array(v) for v in obj.values()] + [0])
    if isinstance(obj, (tuple, list)):
        return max([_largest_array(v) for v in obj] + [0])
    return 0


def convert_batch(paths, convert_one=None, index=None, src: int = 0, group=None, gather: bool = True, device=None, convert_many=None):
    """``VC.vc_multi`` (infer/modules/vc/modules.py:201-266: one ``vc_single`` per file of a folder, sequentially) across the
    ranks of one node.  Rank r converts the contiguous shard ``shard_range(len(paths), r, world)`` of ``paths`` -- the same list
    on every rank -- by calling ``convert_one(path, index)`` (e.g. ``lambda p, ix: vc.vc_single(sid, p, ..., file_index=ix, ...)``
    or a closure over ``Pipeline.pipeline``); files are independent, so there is NO data-path collective.

    ``index``: an ``IVFFlatHIP`` on rank ``src`` (None elsewhere) is replicated with ONE broadcast of its blob over RCCL/xGMI
    (``broadcast_index``) before the first file; any other value (a path every rank reads itself, None) is passed through.

    Returns ``[(path, result)]`` for THIS rank's shard, in order; with ``gather=True`` rank ``src`` instead gets the results of ALL
    files in the order of ``paths`` (one ``gather_object`` of the per-rank host results at the very end -- the reference's
    ``infos`` list) and the other ranks their own shard.  An exception inside ``convert_one`` becomes that file's result (the
    reference appends the traceback text to ``infos`` and goes on), so one bad file never stalls the other ranks.
    Without an initialised process group it is the plain sequential loop.

    ``convert_many(paths_of_this_rank, index) -> list of results`` (instead of ``convert_one``): the rank's whole shard in ONE call, so
    that the files are batched on its GPU -- a closure over ``rvc_amd.pipeline.convert_files`` (``Pipeline.convert_files`` after
    ``install()``), which sends the segments of all files through the generator as ragged batches and through ONE retrieval call.  If it
    raises, the shard is converted again file by file (``convert_many([p], index)``) so that the failure stays with its file.

    Results travel as Python objects (``gather_object``): return what ``vc_multi`` keeps -- an info string, or the path of the file the
    rank wrote -- not the waveforms of a 512-clip folder.  This is ENFORCED when results are gathered: a result that carries a tensor /
    array of more than ``MAX_GATHER_ELEMENTS`` elements (anywhere inside tuples / lists / dicts) raises ``ValueError`` on its rank
    before the collective, naming the file (``gather=False`` keeps big results on their rank)."""
    import sys
    import traceback

    paths = list(paths)
    on = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if on else 0
    world = dist.get_world_size(group) if on else 1
    if on and world > 1:
        from .ivf import IVFFlatHIP

        flag = torch.zeros(1, dtype=torch.int64, device=device if device is not None else (
            torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")))
        if rank == src and isinstance(index, IVFFlatHIP):
            flag[0] = 1
        dist.broadcast(flag, src=src, group=group)
        if int(flag.item()):
            index = broadcast_index(index, src=src, device=device, group=group)
    lo, hi = shard_range(len(paths), rank, world)
    if (convert_one is None) == (convert_many is None):
        raise ValueError("give exactly one of convert_one / convert_many")
    mine = []
    shard = paths[lo:hi]
    if convert_many is not None and shard:
        try:
            res = list(convert_many(shard, index))
            if len(res) != len(shard):
                raise ValueError("convert_many returned %d results for %d files" % (len(res), len(shard)))
            mine = list(zip(shard, res))
        except Exception:  # noqa  (find the file that failed: one call per file, failures become that file's result)
            # the batch's own traceback is logged ONCE: a systematic failure (out of memory, a bad argument) would otherwise only show
            # up as N per-file tracebacks after the whole shard was converted a second time
            print("[rvc_amd.dist] rank %d: convert_many failed on its shard of %d files; converting file by file.  The batch's error:\n%s"
                  % (rank, len(shard), traceback.format_exc()), file=sys.stderr)
            convert_one = lambda p, ix: convert_many([p], ix)[0]  # noqa: E731
    if not mine:
        for p in shard:
            try:
                mine.append((p, convert_one(p, index)))
            except Exception:  # noqa  (modules.py:196-199: vc_single returns the traceback text instead of raising)
                mine.append((p, traceback.format_exc()))
    if not (on and world > 1 and gather):
        return mine
    # the bound on what may be gathered: every rank checks its OWN shard, then all ranks agree (one MAX all-reduce) BEFORE the
    # collective, so a violation raises on every rank instead of leaving the others waiting inside gather_object
    over = [(p, _largest_array(r)) for p, r in mine if _largest_array(r) > MAX_GATHER_ELEMENTS]
    cdev = device if device is not None else (
        torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu"))
    bad = torch.tensor([1 if over else 0], dtype=torch.int64, device=cdev)
    dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=group)
    if int(bad.item()):
        where = ("file %r holds a tensor / array of %d elements" % over[0]) if over else "a result on another rank holds a large tensor / array"
        raise ValueError("convert_batch(gather=True): %s (bound MAX_GATHER_ELEMENTS = %d); return info strings or output paths, "
                         "or call with gather=False" % (where, MAX_GATHER_ELEMENTS))
    parts = [None] * world if rank == src else None
    dist.gather_object(mine, parts, dst=src, group=group)
    if rank != src:
        return mine
    return [pr for part in parts for pr in part]
