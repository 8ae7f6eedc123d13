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
