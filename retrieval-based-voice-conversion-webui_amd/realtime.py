"""Host-side mirror of the realtime caller ``infer/lib/rtrvc.py`` ``RVC.infer`` for everything after HuBERT and the f0
extractor (rtrvc.py:163-251): retrieval on the new frames only, pitch caches, x2 + protect, ``net_g.infer`` with
``skip_head / return_length / return_length2``.  All tensors live on the GPU; the compute runs in the HIP library
(``IVFFlatHIP``, ``glue``, the accelerated ``net_g``), this class only keeps the rolling state the reference keeps.

    rt = RealtimeVC(net_g, index=rvc_amd.read_index(path), index_rate=0.75, device="cuda:0")
    wav = rt.infer(hubert_feats, n_input_samples, block_frame_16k, skip_head, return_length, pitch=p, pitchf=pf)

HuBERT and the f0 estimators stay PyTorch, as in the reference; the optional formant resample
(``torchaudio.transforms.Resample``, rtrvc.py:249-259) is ``SincResample`` below: the same windowed-sinc polyphase filter on a
HIP kernel (torchaudio is not installable offline: its published formula is restated, parity unpinned).
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import _lib, glue


def f0_extractor_frame(block_frame_16k: int, method: str = "fcpe", window: int = 160) -> int:
    """Samples of the rolling input the f0 estimator sees per block (rtrvc.py:203-207)."""
    n = int(block_frame_16k) + 800
    if method == "rmvpe":
        n = 5120 * ((n - 1) // 5120 + 1) - window
    return n


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """The filter bank of ``torchaudio.transforms.Resample(orig_freq, new_freq, dtype=torch.float32)`` (its defaults:
    ``sinc_interp_hann``, ``lowpass_filter_width`` 6, ``rolloff`` 0.99), restated from torchaudio's published
    ``functional._get_sinc_resample_kernel``: frequencies reduced by their gcd, ``base = min(orig, new) * rolloff``,
    ``width = ceil(lpw * orig / base)``, taps ``t = (-k/new + i/orig) * base`` for ``i`` in ``[-width, width + orig)`` clamped to
    ``+-lpw``, hann window ``cos(t * pi / lpw / 2)^2``, ``sinc(t * pi) * window * base / orig``; evaluated in float32 like
    torchaudio does for ``dtype=float32``.  -> (kernel [new, 2*width + orig] float32 CPU tensor, width, orig, new)."""
    g = math.gcd(int(orig_freq), int(new_freq))
    of, nf = int(orig_freq) // g, int(new_freq) // g
    base = min(of, nf) * rolloff
    width = int(math.ceil(lowpass_filter_width * of / base))
    idx = torch.arange(-width, width + of, dtype=torch.float32)[None, :] / of
    t = torch.arange(0, -nf, -1, dtype=torch.float32)[:, None] / nf + idx
    t = t * base
    t = t.clamp(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kern = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base / of)
    return kern.contiguous(), width, of, nf


class SincResample:
    """``torchaudio.transforms.Resample(orig_freq, new_freq, dtype=torch.float32).to(device)`` as the realtime path uses it
    (rtrvc.py:251-259): call with a [1, n] or [n] float32 CUDA tensor, get ``ceil(new * n / orig)`` samples back."""

    def __init__(self, orig_freq: int, new_freq: int, device):
        k, self.width, self.orig, self.new = sinc_resample_kernel(orig_freq, new_freq)
        self.device = torch.device(device)
        self.kernel = k.to(self.device)

    def __call__(self, wav: torch.Tensor) -> torch.Tensor:
        if wav.device.type != "cuda":
            raise _lib.RvcmiError("SincResample input must live on the GPU (no CPU fallback)")
        shape = wav.shape
        x = wav.reshape(-1, shape[-1]).to(torch.float32).contiguous()
        if self.orig == self.new:
            return wav
        n = int(x.shape[1])
        n_out = -(-self.new * n // self.orig)
        out = torch.empty(x.shape[0], n_out, device=x.device, dtype=torch.float32)
        import ctypes as C

        with torch.cuda.device(x.device):
            st = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
            for r in range(x.shape[0]):
                _lib.check(_lib.lib().rvcmi_glue_resample_poly(C.c_void_p(x[r].data_ptr()), n, C.c_void_p(self.kernel.data_ptr()), self.orig,
                                                               self.new, int(self.kernel.shape[1]), self.width, C.c_void_p(out[r].data_ptr()),
                                                               n_out, st))
        return out.reshape(shape[:-1] + (n_out,))


class PitchCache:
    """``cache_pitch`` / ``cache_pitchf`` of rtrvc.py:63-66 and their per-block update (rtrvc.py:213-217)."""

    def __init__(self, device, size: int = 1024):
        self.pitch = torch.zeros(size, device=device, dtype=torch.long)
        self.pitchf = torch.zeros(size, device=device, dtype=torch.float32)

    def update(self, pitch: torch.Tensor, pitchf: torch.Tensor, block_frame_16k: int, window: int = 160) -> None:
        """Shift both caches left by one block and write the estimator's new frames (its first three and last one are
        dropped) at the end."""
        shift = int(block_frame_16k) // int(window)
        n = int(pitch.shape[0])
        if n < 5 or n - 4 > self.pitch.numel():
            raise ValueError("pitch block of %d frames does not fit the cache" % n)
        if shift > 0:
            self.pitch[:-shift] = self.pitch[shift:].clone()
            self.pitchf[:-shift] = self.pitchf[shift:].clone()
        self.pitch[4 - n:] = pitch[3:-1].to(self.pitch.device, torch.long)
        self.pitchf[4 - n:] = pitchf[3:-1].to(self.pitchf.device, torch.float32)

    def window(self, p_len: int, return_length: int, return_length2: int):
        """(pitch, pitchf) [1, p_len] handed to ``net_g.infer`` (rtrvc.py:216-219; pitchf follows the formant factor)."""
        return self.pitch[None, -p_len:], self.pitchf[None, -p_len:] * return_length2 / return_length


class RealtimeVC:
    def __init__(self, net_g, index=None, index_rate: float = 0.0, device="cuda:0", if_f0: int = 1, tgt_sr: int = 48000,
                 f0_up_key: float = 0.0, formant_shift: float = 0.0, window: int = 160):
        self.net_g, self.index, self.index_rate = net_g, index, float(index_rate)
        self.device = torch.device(device)
        self.if_f0, self.tgt_sr, self.window = int(if_f0), int(tgt_sr), int(window)
        self.f0_up_key, self.formant_shift = f0_up_key, formant_shift
        self.cache = PitchCache(self.device)
        self._resample = {}
        self._consts = {}

    def _const(self, v: int) -> torch.Tensor:
        t = self._consts.get(v)
        if t is None:
            t = self._consts[v] = torch.tensor([v], dtype=torch.long, device=self.device)
        return t

    # rtrvc.py:122-132
    def set_key(self, new_key):
        self.f0_up_key = new_key

    def set_formant(self, new_formant):
        self.formant_shift = new_formant

    def set_index_rate(self, new_index_rate):
        self.index_rate = float(new_index_rate)

    def infer(self, feats: torch.Tensor, n_input_samples: int, block_frame_16k: int, skip_head: int, return_length: int,
              pitch: Optional[torch.Tensor] = None, pitchf: Optional[torch.Tensor] = None, protect: float = 1.0,
              sid: int = 0) -> torch.Tensor:
        """``feats`` [1, n, d]: the HuBERT output of the whole rolling window (before its last frame is repeated,
        rtrvc.py:163).  ``pitch`` / ``pitchf`` [m]: what the f0 estimator returned for the last
        ``f0_extractor_frame`` samples (``None`` for a model without f0).  Returns the block's waveform [L]."""
        if feats.dim() != 3 or feats.shape[0] != 1:
            raise ValueError("feats must be [1, n, d]")
        feats = torch.cat((feats, feats[:, -1:, :]), 1)                                   # rtrvc.py:163
        p_len = int(n_input_samples) // self.window                                       # :189
        factor = pow(2, self.formant_shift / 12)                                          # :190
        return_length2 = int(math.ceil(return_length * factor))                           # :191
        pf = pitchf if (protect < 0.5 and pitch is not None and pitchf is not None) else None   # :224
        if pf is not None and pf.numel() != p_len:
            # rtrvc.py:221-231 multiplies feats [1, p_len, d] by the RAW estimator output pitchff [m, 1]: torch broadcasting makes
            # that an error unless m == p_len.  Truncating instead would mix frames that are not aligned with the feature rows.
            raise RuntimeError("protect < 0.5 needs pitchf of exactly p_len = %d frames (got %d), as the reference's broadcast at "
                               "infer/lib/rtrvc.py:229 does" % (p_len, pf.numel()))
        cache_pitch = cache_pitchf = None
        if self.if_f0 == 1:
            if pitch is None or pitchf is None:
                raise ValueError("an f0 model needs the block's pitch / pitchf")
            self.cache.update(pitch, pitchf, block_frame_16k, self.window)                # :213-217
            cache_pitch, cache_pitchf = self.cache.window(p_len, return_length, return_length2)
        use_index = self.index is not None and self.index_rate > 0                        # :167
        phone = glue.retrieve_blend_expand(feats, self.index if use_index else None, self.index_rate if use_index else 0.0,
                                           pf, protect if pf is not None else 1.0, p_len, realtime_guard=True,
                                           skip_rows=int(skip_head) // 2)                  # :167-185, 221-233
        lengths, sid_t = self._const(p_len), self._const(int(sid))   # cached: no host-to-device copy per block
        with torch.no_grad():
            audio = self.net_g.infer(phone, lengths, sid_t, pitch=cache_pitch, pitchf=cache_pitchf, skip_head=skip_head,
                                     return_length=return_length, return_length2=return_length2)   # :236-247
        audio = audio.squeeze(1).float()
        upp_res = int(math.floor(factor * self.tgt_sr // 100))                            # :248
        if upp_res != self.tgt_sr // 100:                                                 # :249-259 (formant shift only)
            if upp_res not in self._resample:
                self._resample[upp_res] = SincResample(upp_res, self.tgt_sr // 100, self.device)
            audio = self._resample[upp_res](audio[:, : return_length * upp_res].contiguous())
        return audio.squeeze()


RT_GRAPH_AFTER = 3  # eager calls of one (window length, key) before its f0 chain is captured (MIOpen / rocFFT pick their algorithms and plans first)


def _rmvpe_f0_graphed(self, wav: torch.Tensor, p_len: int, key: int):
    """``pipeline._rmvpe_on_device`` for the realtime loop, whose f0 window has the SAME length block after block (rtrvc.py:203-207): the chain
    waveform -> mel -> RMVPE -> salience decode is ~400 launches for a 32-frame input -- 7 ms of host time eager, 90 M parameters of U-Net at a
    tiny size -- so after ``RT_GRAPH_AFTER`` eager calls it is captured ONCE per (window length, key) into a hipGraph with a static input buffer
    and replayed (the launch-bound inner loop the design captures everywhere else too).  Same kernels, same results.  ``RVCMI_RT_GRAPH=0``, a
    capture that fails (an op that cannot be captured in this build), or a CPU / foreign estimator keep the eager call.  -> (pitch, pitchf)
    [1, p_len] like ``_rmvpe_on_device`` (tensors the next replay overwrites: the caller copies them into its pitch cache at once), or None."""
    import os

    from .pipeline import _rmvpe_on_device

    if os.environ.get("RVCMI_RT_GRAPH", "1") == "0" or not (torch.is_tensor(wav) and wav.is_cuda and wav.dtype == torch.float32 and wav.dim() == 1):
        return _rmvpe_on_device(self, wav, p_len, key)
    cache = self.__dict__.setdefault("_rvcmi_f0_graphs", {})
    k = (int(wav.shape[0]), int(p_len), int(key), str(wav.device))
    e = cache.setdefault(k, {"calls": 0})
    if "graph" in e:
        e["in"].copy_(wav)
        e["graph"].replay()
        return e["out"]
    out = _rmvpe_on_device(self, wav, p_len, key)
    e["calls"] += 1
    if out is not None and e["calls"] == RT_GRAPH_AFTER:
        try:
            static_in = wav.clone()
            torch.cuda.synchronize(wav.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                o = _rmvpe_on_device(self, static_in, p_len, key)
            if o is None:
                raise RuntimeError("no device f0 inside the capture")
            e.update(graph=g, out=o)
            e["in"] = static_in
        except Exception as ex:  # noqa  (keeps the eager path; never tried again for this key)
            import warnings

            warnings.warn("rvc_amd: the realtime f0 chain could not be captured into a hipGraph (%s: %s); running it eager" % (type(ex).__name__, ex))
            torch.cuda.synchronize(wav.device)
    return out


def rvc_infer_hip(self, input_wav: torch.Tensor, block_frame_16k, skip_head, return_length, f0method, protect: float = 1.0):
    """Drop-in ``RVC.infer`` (infer/lib/rtrvc.py:134-260; bound by ``rvc_amd.install()``): HuBERT and the f0 estimator are the
    object's own PyTorch modules, everything after them runs on the device through a ``RealtimeVC`` that lives on the object.
    A precomputed ``(pitch, pitchf)`` tuple or an index object that is not ours goes to the reference's own method."""
    from .ivf import IVFFlatHIP

    index = getattr(self, "index", None)
    if isinstance(f0method, tuple) or (index is not None and not isinstance(index, IVFFlatHIP)):
        orig = getattr(rvc_infer_hip, "_rvcmi_original", None)
        if orig is None:
            raise TypeError("RVC.infer: unsupported arguments for the HIP path and no reference method is bound")
        return orig(self, input_wav, block_frame_16k, skip_head, return_length, f0method, protect)
    rt = getattr(self, "_rvcmi_rt", None)
    if rt is None or rt.net_g is not self.net_g:
        rt = self._rvcmi_rt = RealtimeVC(self.net_g, index=index, index_rate=self.index_rate, device=self.device, if_f0=self.if_f0,
                                         tgt_sr=self.tgt_sr, f0_up_key=self.f0_up_key, formant_shift=self.formant_shift,
                                         window=self.window)
    rt.index, rt.index_rate = index, float(self.index_rate)      # set_index_rate / set_key / set_formant act on the RVC object
    rt.f0_up_key, rt.formant_shift = self.f0_up_key, self.formant_shift
    with torch.no_grad():                                           # rtrvc.py:142-162
        feats = input_wav.half() if self.is_half else input_wav.float()
        feats = feats.to(self.device)
        if feats.dim() == 2:
            feats = feats.mean(-1)
        feats = feats.view(1, -1)
        mask = torch.zeros(feats.shape, dtype=torch.bool, device=feats.device)
        logits = self.hubert.extract_features(source=feats, padding_mask=mask, output_layer=9 if self.version == "v1" else 12)
        feats = self.hubert.final_proj(logits[0]) if self.version == "v1" else logits[0]
    pitch = pitchf = None
    if self.if_f0 == 1:                                             # rtrvc.py:203-212
        n = f0_extractor_frame(block_frame_16k, f0method, self.window)
        key = self.f0_up_key - self.formant_shift
        got = None
        if f0method == "rmvpe" and float(key).is_integer() and getattr(self, "f0_gen", None) is not None:
            # RMVPE stays on the device from the waveform to (pitch, pitchf): the reference's Generator.calculate (rvc/f0/gen.py:58-59, 103-113) copies
            # the window to the host, the salience back, and decodes it in numpy with a python loop over frames; pipeline._rmvpe_on_device is the same
            # chain (mel, network, rvcmi_glue_rmvpe_f0 = _decode + _resize_f0 + _interpolate_f0 + post_process, golden-tested against the reference's)
            # without a host hop, and swaps the network's GRU for the HIP one (3.7 / 7.2 ms -> 0.06 / 0.10 ms for a 32- / 64-frame window,
            # tools/gru_time.py), replayed from a hipGraph after the first blocks (_rmvpe_f0_graphed).  The integer key is the C ABI's; a fractional key (formant
            # slider) takes the reference's own method below.
            w_f0 = input_wav[-n:]
            got = _rmvpe_f0_graphed(self, w_f0.contiguous() if torch.is_tensor(w_f0) else w_f0, int(w_f0.shape[0]) // self.window, int(key))
        if got is not None:
            pitch, pitchf = got[0][0], got[1][0]
        else:
            if f0method == "rmvpe" and getattr(getattr(self, "f0_gen", None), "rmvpe", None) is not None:
                from .gru import accelerate_f0_rmvpe

                accelerate_f0_rmvpe(self.f0_gen.rmvpe)   # (beyond SURVEY 8) at least the network's GRU, once the generator has loaded it
            pitch, pitchf = self._get_f0(input_wav[-n:], key, method=f0method)
    return rt.infer(feats, int(input_wav.shape[0]), block_frame_16k, skip_head, return_length, pitch=pitch, pitchf=pitchf, protect=protect)
