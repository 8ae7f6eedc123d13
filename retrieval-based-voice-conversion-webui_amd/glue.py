"""Device-resident versions of the host-side glue of ``Pipeline.vc`` / ``Pipeline.pipeline`` (SURVEY.md section 8f
row 2), so that nothing between the PyTorch-ROCm feature extractors and ``net_g.infer`` has to visit the host:

    retrieve_blend_expand(feats, index, index_rate, pitchf, protect, p_len)   pipeline.py:118-159
    rmvpe_f0(salience, p_len, f0_up_key, thred)                               rvc/f0/rmvpe.py:115-164, f0.py:31-78, gen.py:10-41
    f0_post(f0, f0_up_key)                                                    rvc/f0/gen.py:10-41
    change_rms(audio16k, 16000, audio_opt, tgt_sr, rms_mix_rate)              pipeline.py:26-46,351
    scale_int16_range(audio)                                                  pipeline.py:355-359

All take and return CUDA (ROCm) tensors and enqueue on the current stream; a CPU tensor raises (no fallback).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib


def _dev(t: torch.Tensor, what: str) -> torch.device:
    if t.device.type != "cuda":
        raise _lib.RvcmiError("%s must live on the GPU (got %s); the glue has no CPU fallback" % (what, t.device))
    return t.device


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _blend_expand_into(out: torch.Tensor, f: torch.Tensor, index, index_rate: float, pf: Optional[torch.Tensor], protect: float,
                       realtime_guard: bool) -> None:
    """out [p_len, d] (a contiguous row slice) <- x2(f [nq, d]), blended against ``index`` first when given, protect-mixed."""
    nq, d = int(f.shape[0]), int(f.shape[1])
    p_len = int(out.shape[0])
    dev = f.device
    L = _lib.lib()
    with torch.cuda.device(dev):
        if index is not None and index_rate != 0:
            if d != index.d:
                raise ValueError("index mistatch")  # the reference's message (pipeline.py:128)
            if _lib.device_index(index.device) != _lib.device_index(dev):
                raise _lib.RvcmiError("the index lives on %s, the features on %s: read the index on the pipeline's device "
                                      "(read_index(path, device=...))" % (index.device, dev))
            index.reserve(nq)
            _lib.check(L.rvcmi_ivf_search_blend_expand(index._h, nq, _ptr(f), float(index_rate), 8, 1 if realtime_guard else 0,
                                                       _ptr(pf), float(protect), p_len, _ptr(out), _stream(dev)))
        else:
            _lib.check(L.rvcmi_glue_expand_protect(_ptr(f), nq, d, 2, _ptr(pf), float(protect), p_len, _ptr(out), _stream(dev)))


def retrieve_blend_expand(feats: torch.Tensor, index, index_rate: float, pitchf: Optional[torch.Tensor] = None,
                          protect: float = 0.5, p_len: Optional[int] = None, realtime_guard: bool = False,
                          skip_rows: int = 0) -> torch.Tensor:
    """``feats`` [1, nq, d] HuBERT features -> [1, p_len, d]: retrieval blend (when ``index`` is given and
    ``index_rate != 0``), x2 nearest interpolation, truncation to ``p_len`` and the protect mix
    (``pitchf`` [1, >= p_len], applied when ``protect < 0.5`` as in pipeline.py:153).

    ``skip_rows`` > 0 is the realtime form (rtrvc.py:167-185, 221-233): only ``feats[0][skip_rows:]`` is searched and
    blended (the rolling window's old frames keep their HuBERT features); the protect mix runs over all rows."""
    dev = _dev(feats, "feats")
    if feats.dim() != 3 or feats.shape[0] != 1:
        raise ValueError("feats must be [1, nq, d]")
    nq, d = int(feats.shape[1]), int(feats.shape[2])
    p_len = 2 * nq if p_len is None else min(int(p_len), 2 * nq)
    f = feats[0].to(torch.float32).contiguous()
    pf = None
    if pitchf is not None and protect < 0.5:
        pf = pitchf.reshape(-1)[:p_len].to(dev, torch.float32).contiguous()
        if pf.numel() < p_len:
            raise ValueError("pitchf has %d frames, p_len is %d" % (pf.numel(), p_len))
    out = torch.empty(p_len, d, device=dev, dtype=torch.float32)
    h = max(0, min(int(skip_rows), nq))
    if h and index is not None and index_rate != 0:
        # rows [0, h): un-blended (feats0 == feats in the protect mix); rows [h, nq): searched and blended
        head = min(2 * h, p_len)
        _blend_expand_into(out[:head], f[:h], None, 0.0, None if pf is None else pf[:head], protect, False)
        if head < p_len:
            _blend_expand_into(out[head:], f[h:], index, index_rate, None if pf is None else pf[head:], protect, realtime_guard)
    else:
        _blend_expand_into(out, f, index, index_rate, pf, protect, realtime_guard)
    return out.unsqueeze(0).to(feats.dtype)


def rmvpe_f0(salience: torch.Tensor, p_len: int, f0_up_key: int = 0, thred: float = 0.03) -> Tuple[torch.Tensor, torch.Tensor]:
    """RMVPE salience [n, 360] -> (pitch int64 [1, p_len], pitchf float32 [1, p_len]) as ``Generator.calculate`` +
    pipeline.py:270-277 produce them."""
    dev = _dev(salience, "salience")
    if salience.dim() != 2:
        raise ValueError("salience must be [n, bins]")
    s = salience.to(torch.float32).contiguous()
    n, nb = int(s.shape[0]), int(s.shape[1])
    scratch = torch.empty(n, device=dev, dtype=torch.float64)
    pitch = torch.empty(int(p_len), device=dev, dtype=torch.int64)
    pitchf = torch.empty(int(p_len), device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().rvcmi_glue_rmvpe_f0(_ptr(s), n, nb, float(thred), int(p_len), int(f0_up_key), _ptr(scratch), _ptr(pitch),
                                                  _ptr(pitchf), _stream(dev)))
    return pitch.unsqueeze(0), pitchf.unsqueeze(0)


def f0_post(f0: torch.Tensor, f0_up_key: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """f0 in Hz [n] (any estimator) -> (pitch int64 [1, n], pitchf float32 [1, n]): rvc/f0/gen.py post_process."""
    dev = _dev(f0, "f0")
    x = f0.reshape(-1).to(torch.float64).contiguous()
    n = int(x.numel())
    pitch = torch.empty(n, device=dev, dtype=torch.int64)
    pitchf = torch.empty(n, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().rvcmi_glue_f0_post(_ptr(x), n, int(f0_up_key), _ptr(pitch), _ptr(pitchf), _stream(dev)))
    return pitch.unsqueeze(0), pitchf.unsqueeze(0)


def change_rms(data1: torch.Tensor, sr1: int, data2: torch.Tensor, sr2: int, rate: float) -> torch.Tensor:
    """``change_rms(audio, 16000, audio_opt, tgt_sr, rms_mix_rate)`` of pipeline.py:26-46,351 on the device, IN PLACE on
    ``data2`` (returned): ``data2 *= rms1**(1-rate) * max(rms2, 1e-6)**(rate-1)`` with half-second frame RMS envelopes."""
    dev = _dev(data2, "data2")
    for t, nm in ((data1, "data1"), (data2, "data2")):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.dim() != 1 or t.device != dev:
            raise ValueError("%s must be a contiguous 1-D float32 tensor on %s" % (nm, dev))
    n1, n2 = int(data1.numel()), int(data2.numel())
    scratch = torch.empty(2 + n1 // (int(sr1) // 2) + n2 // (int(sr2) // 2), device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().rvcmi_glue_change_rms(_ptr(data1), n1, int(sr1), _ptr(data2), n2, int(sr2), float(rate), _ptr(scratch),
                                                    _stream(dev)))
    return data2


def scale_int16_range(audio: torch.Tensor) -> torch.Tensor:
    """In place: ``audio *= 32768 / max(1, |audio|.max() / 0.99)`` (pipeline.py:355-359); returns ``audio``."""
    dev = _dev(audio, "audio")
    if audio.dtype != torch.float32 or not audio.is_contiguous():
        raise ValueError("audio must be a contiguous float32 tensor")
    scratch = torch.empty(256, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().rvcmi_glue_scale_int16_range(_ptr(audio), audio.numel(), _ptr(scratch), _stream(dev)))
    return audio


def sola(infer_wav: torch.Tensor, sola_buffer: torch.Tensor, fade_in: torch.Tensor, fade_out: torch.Tensor, block_frame: int,
         search_frame: int, return_offset: bool = False):
    """The SOLA stitch of gui.py:1057-1090 in one launch: returns the ``block_frame`` output samples and updates
    ``sola_buffer`` in place (``return_offset=True`` also returns the chosen offset as a 1-element int32 tensor)."""
    dev = _dev(infer_wav, "infer_wav")
    for t, nm in ((infer_wav, "infer_wav"), (sola_buffer, "sola_buffer"), (fade_in, "fade_in"), (fade_out, "fade_out")):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev:
            raise ValueError("%s must be a contiguous float32 tensor on %s" % (nm, dev))
    Lb = int(sola_buffer.numel())
    out = torch.empty(int(block_frame), device=dev, dtype=torch.float32)
    off = torch.empty(1, device=dev, dtype=torch.int32)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().rvcmi_glue_sola(_ptr(infer_wav), infer_wav.numel(), _ptr(sola_buffer), Lb, int(search_frame), _ptr(fade_in),
                                              _ptr(fade_out), int(block_frame), _ptr(out), _ptr(off), _stream(dev)))
    return (out, off) if return_offset else out
