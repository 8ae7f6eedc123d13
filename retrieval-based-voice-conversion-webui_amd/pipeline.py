"""Retrieval glue of ``Pipeline.vc`` (infer/modules/vc/pipeline.py:113-138) and ``RVC.infer``
(infer/lib/rtrvc.py:167-187), device resident.

The reference moves HuBERT features to the host, calls faiss, blends with numpy and moves the result
back (2x D2H + 2x H2D per chunk).  ``retrieve_blend`` does the same arithmetic on the GPU in one call.
"""
from __future__ import annotations

import torch

from .ivf import IVFFlatHIP


def retrieve_blend(feats: torch.Tensor, index: IVFFlatHIP, index_rate: float, k: int = 8,
                   realtime_guard: bool = False) -> torch.Tensor:
    """feats [1 or B, T, d] (HuBERT output, any float dtype, CUDA) -> same shape/dtype with

        score, ix = index.search(npy, k=8); weight = square(1/score); weight /= weight.sum(1)
        npy = sum(big_npy[ix] * weight[..., None], 1); feats = npy*index_rate + (1-index_rate)*feats

    (pipeline.py:126-138).  ``index is None`` or ``index_rate == 0`` returns feats untouched, as the
    reference's guard at pipeline.py:113-117 does.  ``realtime_guard`` = rtrvc.py:173.
    """
    if index is None or index_rate == 0:
        return feats
    shape, dtype = feats.shape, feats.dtype
    flat = feats.reshape(-1, shape[-1]).to(torch.float32).contiguous()
    if flat.data_ptr() == feats.data_ptr():
        flat = flat.clone()  # the reference builds a new tensor; keep the caller's feats0 copy valid
    index.search_blend(flat, float(index_rate), k, skip_if_short=realtime_guard)
    return flat.reshape(shape).to(dtype)
