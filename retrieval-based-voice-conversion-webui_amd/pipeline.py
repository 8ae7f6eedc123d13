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


# ------------------------------------------------------------------------------------------------------------------------
# Device-resident ``Pipeline.vc`` / ``Pipeline.pipeline`` (infer/modules/vc/pipeline.py:76-184, 186-360)
# ------------------------------------------------------------------------------------------------------------------------
# ``install()`` binds ``vc_hip`` / ``pipeline_hip`` as the ``vc`` / ``pipeline`` methods of the reference's own ``Pipeline``
# class (whose ``__init__`` state -- x_pad, window, t_pad*, is_half, device, f0_gen -- they read), so an unmodified
# ``VC.vc_single`` runs them.  HuBERT stays the caller's ``model`` on PyTorch-ROCm; from its output on nothing visits the host:
# search + blend + x2 + protect mix in one launch (rvcmi_ivf_search_blend_expand), ``net_g.infer`` on the HIP front + generator,
# segments concatenated, ``change_rms`` and the int16-range scaling on the device, ONE copy to the host at the very end
# (the API returns a numpy array).  Host work that remains is the reference's own input preparation on the 16 kHz waveform
# (high-pass ``filtfilt``, the quiet-point search that chooses the cut positions) and estimators other than RMVPE.

def _ref_module(self):
    import sys

    return sys.modules[type(self).__module__]


def _is_hip_index(index) -> bool:
    return isinstance(index, IVFFlatHIP)


def hubert_device(self, model, audio0, pitch, pitchf, version):
    """The HuBERT half of ``Pipeline.vc`` (pipeline.py:90-112, 146-151): -> (feats [1, nq, d] on the GPU, BEFORE retrieval,
    pitch [1, p_len] or None, pitchf or None, p_len)."""
    dev = torch.device(self.device)
    feats = torch.as_tensor(audio0)
    feats = feats.half() if self.is_half else feats.float()
    if feats.dim() == 2:  # double channels (pipeline.py:98-99)
        feats = feats.mean(-1)
    assert feats.dim() == 1, feats.dim()
    feats = feats.view(1, -1)
    padding_mask = torch.zeros(feats.shape, dtype=torch.bool, device=dev)
    with torch.no_grad():
        logits = model.extract_features(source=feats.to(dev), padding_mask=padding_mask, output_layer=9 if version == "v1" else 12)
        feats = model.final_proj(logits[0]) if version == "v1" else logits[0]
    use_f0 = pitch is not None and pitchf is not None
    nq = int(feats.shape[1])
    p_len = min(int(audio0.shape[0]) // self.window, 2 * nq)  # pipeline.py:146-151
    if use_f0:
        pitch, pitchf = pitch[:, :p_len], pitchf[:, :p_len]
    return feats, (pitch if use_f0 else None), (pitchf if use_f0 else None), p_len


def features_device(self, model, audio0, pitch, pitchf, times, index, index_rate, version, protect):
    """``Pipeline.vc`` from HuBERT to the tensor handed to ``net_g.infer`` (pipeline.py:90-159): features on the GPU, retrieval
    blend + x2 + protect mix in one launch.  Returns (feats [1, p_len, d], pitch [1, p_len] or None, pitchf or None, p_len)."""
    from time import time

    from . import glue

    t0 = time()
    feats, pitch, pitchf, p_len = hubert_device(self, model, audio0, pitch, pitchf, version)
    use_f0 = pitch is not None
    use_index = index is not None and index_rate != 0
    feats = glue.retrieve_blend_expand(feats, index if use_index else None, float(index_rate), pitchf if use_f0 else None,
                                       float(protect) if use_f0 else 0.5, p_len)
    times[0] += time() - t0
    return feats, pitch, pitchf, p_len


MAX_FILES_PER_GROUP = 64   # inputs of convert_files whose HuBERT frames / waveforms are resident at once (BASELINE configs[2]: 64 clips per GPU)
MAX_BATCH_QUERIES = 65536  # HuBERT frames per retrieval call of blend_segments (22 minutes of audio; the score scratch stays < 1 GiB)


def blend_segments(raw, index, index_rate, protect):
    """The retrieval half of ``Pipeline.vc`` (pipeline.py:113-159) for MANY segments at once.  ``raw``: ``hubert_device`` results
    (of one file or of several).  All their HuBERT frames go through ONE ``rvcmi_ivf_search_blend_expand`` call per
    ``MAX_BATCH_QUERIES`` frames -- one coarse pass, one list-major scan: the rows of a probed list are read once for every segment
    that probes it -- and every segment gets back exactly the rows its own call would have produced (a row's search, blend, x2 and
    protect mix depend on nothing but that row).  -> items for ``infer_segments``."""
    from . import glue

    use_index = index is not None and index_rate != 0
    items = [None] * len(raw)
    start = 0
    while start < len(raw):
        end, n = start, 0
        while end < len(raw) and (end == start or n + int(raw[end][0].shape[1]) <= MAX_BATCH_QUERIES):
            n += int(raw[end][0].shape[1])
            end += 1
        grp = raw[start:end]
        dev, dtype = grp[0][0].device, grp[0][0].dtype
        use_f0 = grp[0][1] is not None
        F = torch.cat([g[0][0] for g in grp]).to(torch.float32).contiguous()  # [n, d]
        pf = None
        if use_f0 and float(protect) < 0.5:  # the frame-rate pitchf of every segment at ITS output rows (2 x its first query row)
            pf = torch.ones(2 * n, device=dev, dtype=torch.float32)
            o = 0
            for f, _, pff, p_len in grp:
                if pff.numel() < p_len:  # the message of glue.retrieve_blend_expand (the per-segment path) instead of a slice-size mismatch
                    raise ValueError("pitchf has %d frames, p_len is %d" % (pff.numel(), p_len))
                pf[2 * o: 2 * o + p_len] = pff.reshape(-1)[:p_len].to(dev, torch.float32)
                o += int(f.shape[1])
        out = torch.empty(2 * n, F.shape[1], device=dev, dtype=torch.float32)
        glue._blend_expand_into(out, F, index if use_index else None, float(index_rate), pf, float(protect) if use_f0 else 0.5, False)
        o = 0
        for i, (f, pt, pff, p_len) in enumerate(grp):
            items[start + i] = (out[2 * o: 2 * o + p_len].unsqueeze(0).to(dtype), pt, pff, p_len)
            o += int(f.shape[1])
        start = end
    return items


def vc_device(self, model, net_g, sid, audio0, pitch, pitchf, times, index, index_rate, version, protect) -> torch.Tensor:
    """``Pipeline.vc`` up to (not including) its final ``.data.cpu().float().numpy()``: returns the converted segment as a
    1-D device tensor.  ``audio0``: numpy or tensor (16 kHz segment, padded); ``index``: an ``IVFFlatHIP`` or None."""
    from time import time

    dev = torch.device(self.device)
    feats, pitch, pitchf, p_len = features_device(self, model, audio0, pitch, pitchf, times, index, index_rate, version, protect)
    t1 = time()
    plen_t = torch.tensor([p_len], device=dev).long()
    with torch.no_grad():
        audio1 = net_g.infer(feats, plen_t, sid, pitch=pitch, pitchf=pitchf)[0, 0] if pitch is not None else net_g.infer(feats, plen_t, sid)[0, 0]
    times[2] += time() - t1
    return audio1.data


# ---- all segments of a file in ONE ``net_g.infer`` call ------------------------------------------------------------------------------
# The reference converts the segments of a long input one ``vc`` call after the other (pipeline.py:301-343); they are independent
# once f0 is known.  With the ragged-batch entry of the generator (include/rvcmi.h rvcmi_nsf_forward ``lengths``: every item computed
# exactly as a separate call of its own length) they go through the front and the generator as one batch.  The noise of item b is
# drawn as the b-th sequential call would have drawn it (randn(1, 192, T_b); rand(1, 1, 1); randn(1, T_b * upp, 1)), so a seeded run
# consumes the generator exactly like the sequential pipeline.  Padded frames per call (items x longest item) are bounded (``MAX_BATCH_FRAMES``) so that
# the workspace of an hour-long file stays a few GB; segments beyond the bound go into further calls.
MAX_BATCH_FRAMES = 32768


def _ragged_capable(net_g) -> bool:
    from .front import infer_hip

    from .nsf import GeneratorHIP, NSFGeneratorHIP

    inf = getattr(net_g, "infer", None)
    routed = bool(getattr(inf, "_rvcmi_ragged", False)) or getattr(inf, "func", None) is infer_hip
    # ... and the generator behind it must be the HIP one: infer_hip cannot tell a foreign dec the lengths (it raises), and
    # infer_segments reads dec.upp / dec.cfg
    return routed and isinstance(getattr(net_g, "dec", None), (NSFGeneratorHIP, GeneratorHIP))


def infer_segments(net_g, sid, items, times=None):
    """``items``: list of (feats [1, T_b, d], pitch [1, T_b] or None, pitchf [1, T_b] or None, T_b) as ``features_device`` returns them.
    -> list of 1-D device tensors (the waveform of every item, ``T_b * upp`` samples), from as few ``net_g.infer`` calls as
    ``MAX_BATCH_FRAMES`` allows.  Bit-identical to one call per item with the kernel family pinned; the launcher's own per-shape choices
    may differ between a batch and a single clip by operand rounding."""
    from time import time

    t0 = time()
    outs = [None] * len(items)
    dev = items[0][0].device
    use_f0 = items[0][1] is not None
    upp = int(getattr(net_g.dec, "upp"))
    IC = int(net_g.dec.cfg["inter_channels"])
    start = 0
    while start < len(items):
        # the workspace of a call is B x T_max frames (padded), so THAT is what the bound applies to, not the sum of the lengths
        end, tmax = start, 0
        while end < len(items) and (end == start or (end - start + 1) * max(tmax, items[end][3]) <= MAX_BATCH_FRAMES):
            tmax = max(tmax, items[end][3])
            end += 1
        grp = items[start:end]
        B, Tm = len(grp), max(it[3] for it in grp)
        d = int(grp[0][0].shape[2])
        phone = torch.zeros(B, Tm, d, device=dev, dtype=grp[0][0].dtype)
        pitch = torch.ones(B, Tm, device=dev, dtype=torch.long) if use_f0 else None
        pitchf = torch.zeros(B, Tm, device=dev, dtype=grp[0][2].dtype) if use_f0 else None
        nzp = torch.zeros(B, IC, Tm, device=dev, dtype=grp[0][0].dtype)
        ndec = torch.zeros(B, Tm * upp, device=dev, dtype=grp[0][2].dtype) if use_f0 else None
        for b, (f, pt, pf, Tb) in enumerate(grp):
            phone[b, :Tb] = f[0]
            nzp[b, :, :Tb] = torch.randn(1, IC, Tb, device=dev, dtype=f.dtype)[0]       # randn_like(m_p)        synthesizers.py:182
            if use_f0:
                pitch[b, :Tb], pitchf[b, :Tb] = pt[0], pf[0]
                torch.rand(1, 1, 1, device=dev)                                          # rand_ini               generators.py:164
                ndec[b, :Tb * upp] = torch.randn(1, Tb * upp, 1, device=dev, dtype=pf.dtype)[0, :, 0]  # generators.py:192
        lens = torch.tensor([it[3] for it in grp], device=dev).long()
        sidb = sid.reshape(-1)[:1].expand(B).contiguous()
        with torch.no_grad():
            if use_f0:
                o = net_g.infer(phone, lens, sidb, pitch=pitch, pitchf=pitchf, noise_zp=nzp, noise_dec=ndec, ragged=True)
            else:
                o = net_g.infer(phone, lens, sidb, noise_zp=nzp, ragged=True)
        for b, it in enumerate(grp):
            outs[start + b] = o[b, 0, : it[3] * upp].data
        start = end
    if times is not None:
        times[2] += time() - t0
    return outs


def vc_hip(self, model, net_g, sid, audio0, pitch, pitchf, times, index, big_npy, index_rate, version, protect):
    """Drop-in ``Pipeline.vc`` (same arguments, same numpy return).  An index object that is not ours (a real faiss index
    the HIP reader could not serve) is handed to the reference's own ``vc``."""
    if index is not None and not _is_hip_index(index):
        orig = getattr(vc_hip, "_rvcmi_original", None)
        if orig is None:
            raise TypeError("Pipeline.vc: index is a %s, not an IVFFlatHIP, and no reference vc is bound" % type(index).__name__)
        return orig(self, model, net_g, sid, audio0, pitch, pitchf, times, index, big_npy, index_rate, version, protect)
    use_index = index is not None and big_npy is not None and index_rate != 0  # the reference's guard, pipeline.py:113-117
    out = vc_device(self, model, net_g, sid, audio0, pitch, pitchf, times, index if use_index else None, index_rate, version, protect)
    return out.cpu().float().numpy()


def _cut_points(self, audio, audio_pad_w):
    """pipeline.py:219-232: where a long input is cut -- the quietest sample (|x| summed over one window) within
    +-t_query of every multiple of t_center.  numpy float64 like the reference (same sums, same first-minimum rule)."""
    import numpy as np

    opt_ts = []
    if audio_pad_w.shape[0] > self.t_max:
        audio_sum = np.zeros_like(audio)
        for i in range(self.window):
            audio_sum += np.abs(audio_pad_w[i: i - self.window])
        for t in range(self.t_center, audio.shape[0], self.t_center):
            seg = audio_sum[t - self.t_query: t + self.t_query]
            opt_ts.append(t - self.t_query + int(np.where(seg == seg.min())[0][0]))
    return opt_ts


RMVPE_THRED = 0.03  # rvc/f0/gen.py:113: Generator.calculate hard-codes compute_f0(..., filter_radius=0.03) for rmvpe


def _rmvpe_on_device(self, audio_pad, p_len, f0_up_key):
    """RMVPE salience on PyTorch-ROCm (the reference's own mel extractor + network), decoded by ``rvcmi_glue_rmvpe_f0``
    (rvc/f0/rmvpe.py:119-164, f0.py:31-78, gen.py:10-41) without visiting the host.  None when this f0_gen has no torch RMVPE.

    The caller's ``filter_radius`` is NOT the voicing threshold: it is harvest's median radius (the UI slider 0..7, web.py:794)
    that ``vc_single`` forwards to every estimator; ``Generator.calculate`` ignores it for rmvpe and passes the constant 0.03
    (rvc/f0/gen.py:113), and so does this function."""
    from . import glue

    gen = self.f0_gen
    if not hasattr(gen, "rmvpe"):
        try:
            from rvc.f0.rmvpe import RMVPE

            gen.rmvpe = RMVPE("%s/rmvpe.pt" % gen.rmvpe_root, is_half=gen.is_half, device=gen.device)
        except Exception:  # noqa  (no checkpoint / onnx-only build: the reference path decides what to do)
            return None
    r = gen.rmvpe
    if not (hasattr(r, "mel_extractor") and hasattr(r, "_mel2hidden")) or "privateuseone" in str(getattr(r, "device", "")):
        return None
    # (beyond SURVEY.md section 8) the network's bidirectional GRU -- 75-90 % of a conversion as MIOpen runs it, bench.py --e2e -- on the
    # persistent HIP kernel, once per RMVPE object; everything else of RMVPE stays on PyTorch-ROCm.  RVCMI_RMVPE_GRU=0 keeps torch's GRU.
    from .gru import accelerate_f0_rmvpe

    accelerate_f0_rmvpe(r)
    wav = torch.as_tensor(audio_pad)
    with torch.no_grad():
        mel = r.mel_extractor(wav.float().to(r.device).unsqueeze(0), center=True)
        hidden = r._mel2hidden(mel)
    return glue.rmvpe_f0(hidden.squeeze(0).float(), p_len, int(f0_up_key), RMVPE_THRED)


INDEX_CACHE_ENTRIES = 2  # index files kept resident (a WebUI session alternates between very few voices)
_INDEX_CACHE = {}
_INDEX_CACHE_LOCK = __import__("threading").Lock()  # (the WebUI serves requests from worker threads)


def _open_index(self, file_index, index_rate):
    """pipeline.py:205-218.  -> (IVFFlatHIP or None, needs_reference): ``needs_reference`` = an index kind only real faiss reads
    (not IVF-Flat / L2 ...) while the reference's own pipeline and the real faiss module are bound: the caller hands the whole
    file to that pipeline (retrieval included; its vc calls come back through vc_hip, which passes a non-HIP index on)."""
    import os
    import traceback

    from . import _lib, ivf

    if not (file_index != "" and os.path.exists(file_index) and index_rate != 0):
        return None, False
    try:
        # on THIS pipeline's device (config.device), not the process's current GPU; big_npy is never materialised.
        # The reference re-reads the file on every call (pipeline.py:205-218: faiss.read_index + reconstruct_n); here the parsed index stays
        # resident on the GPU between calls, keyed by (path, size, mtime, device) -- a 10000 x 768 index is 31 MB of file read + upload per
        # call otherwise (round 6, bench.py --e2e: the largest single item around the hot path at one file per call).  RVCMI_INDEX_CACHE=0 turns it off.
        dev = torch.device(self.device)
        st = os.stat(file_index)
        key = (os.path.abspath(file_index), st.st_size, st.st_mtime_ns, str(dev))
        cached = os.environ.get("RVCMI_INDEX_CACHE", "1") != "0"
        if cached:
            with _INDEX_CACHE_LOCK:
                hit = _INDEX_CACHE.get(key)
            if hit is not None:
                return hit, False
        index = ivf.read_index(file_index, device=dev)
        if cached:
            with _INDEX_CACHE_LOCK:
                while len(_INDEX_CACHE) >= INDEX_CACHE_ENTRIES:
                    _INDEX_CACHE.pop(next(iter(_INDEX_CACHE)))  # oldest entry out (dicts keep insertion order)
                _INDEX_CACHE[key] = index
        return index, False
    except _lib.RvcmiError as e:
        orig = getattr(pipeline_hip, "_rvcmi_original", None)
        real = getattr(getattr(_ref_module(self), "faiss", None), "_rvcmi_real", None)
        if e.code == _lib.ERR_IO and orig is not None and real is not None:
            return None, True
        traceback.print_exc()  # the reference prints and converts without an index (pipeline.py:216-218)
        return None, False


def _prepare_file(self, model, sid, audio, times, f0_up_key, f0_method, if_f0, filter_radius, version, f0_file, collect):
    """``Pipeline.pipeline`` from its input to the point where a segment would enter ``vc`` (pipeline.py:219-300): high-pass,
    cut points, reflection pad, f0 (RMVPE decoded on the device), then ``collect(audio_segment, pitch_slice, pitchf_slice)`` for
    every segment in order.  -> the filtered 16 kHz input (``change_rms`` needs it)."""
    import traceback
    from time import time

    import numpy as np

    ref = _ref_module(self)
    dev = torch.device(self.device)
    audio = ref.signal.filtfilt(ref.bh, ref.ah, audio)
    opt_ts = _cut_points(self, audio, np.pad(audio, (self.window // 2, self.window // 2), mode="reflect"))
    t1 = time()
    audio_pad = np.pad(audio, (self.t_pad, self.t_pad), mode="reflect")
    p_len = audio_pad.shape[0] // self.window
    inp_f0 = None
    if hasattr(f0_file, "name"):
        try:
            with open(f0_file.name, "r") as f:
                lines = f.read().strip("\n").split("\n")
            if lines and lines[0]:
                inp_f0 = np.array([[float(v) for v in ln.split(",")] for ln in lines], dtype="float32")
        except Exception:  # noqa
            traceback.print_exc()
    pitch = pitchf = None
    if if_f0:
        got = None
        if if_f0 == 1 and f0_method == "rmvpe" and inp_f0 is None:
            got = _rmvpe_on_device(self, audio_pad, p_len, f0_up_key)
        if got is not None:
            pitch, pitchf = got[0][:, :p_len].long(), got[1][:, :p_len].float()
        else:
            if if_f0 == 1:
                pitch, pitchf = self.f0_gen.calculate(audio_pad, p_len, f0_up_key, f0_method, filter_radius, inp_f0)
            else:
                pitch, pitchf = f0_method  # pipeline.py:268-269: a precomputed (coarse, Hz) pair
            pitch = torch.as_tensor(np.asarray(pitch)[:p_len], device=dev).unsqueeze(0).long()
            pitchf = torch.as_tensor(np.asarray(pitchf)[:p_len].astype(np.float32), device=dev).unsqueeze(0).float()
    times[1] += time() - t1
    w = self.window
    s, t = 0, None
    for t in opt_ts:
        t = t // w * w
        collect(audio_pad[s: t + self.t_pad2 + w], pitch[:, s // w: (t + self.t_pad2) // w] if if_f0 else None,
                pitchf[:, s // w: (t + self.t_pad2) // w] if if_f0 else None)
        s = t
    lo = (t // w) if t is not None else 0
    collect(audio_pad[t:], pitch[:, lo:] if if_f0 else None, pitchf[:, lo:] if if_f0 else None)
    return audio, len(opt_ts)


def _finish_file(self, segs, audio, tgt_sr, resample_sr, rms_mix_rate):
    """pipeline.py:344-360 on the device: concatenate the trimmed segments, ``change_rms``, (``resample_sr``: host, like the
    reference), int16-range scaling, ONE copy to the host."""
    import numpy as np

    from . import glue

    dev = torch.device(self.device)
    audio_opt = torch.cat([o[self.t_pad_tgt: o.shape[0] - self.t_pad_tgt].float() for o in segs]).contiguous()
    if rms_mix_rate != 1:
        a16 = torch.as_tensor(np.ascontiguousarray(audio, dtype=np.float32), device=dev)
        audio_opt = glue.change_rms(a16, 16000, audio_opt, int(tgt_sr), float(rms_mix_rate))
    if tgt_sr != resample_sr >= 16000:
        # librosa's soxr resampler has no device twin (and no offline oracle): this one option goes through the host like the
        # reference (pipeline.py:351-354) and comes back for the scaling
        host = _ref_module(self).librosa.resample(audio_opt.cpu().numpy(), orig_sr=tgt_sr, target_sr=resample_sr)
        audio_opt = torch.as_tensor(host, device=dev).float().contiguous()
    glue.scale_int16_range(audio_opt)
    return audio_opt.cpu().numpy()


def pipeline_hip(self, model, net_g, sid, audio, times, f0_up_key, f0_method, file_index, index_rate, if_f0, filter_radius, tgt_sr,
                 resample_sr, rms_mix_rate, version, protect, f0_file=None):
    """Drop-in ``Pipeline.pipeline``: same arguments, same numpy return (float array scaled to the int16 range)."""
    import os

    index, needs_ref = _open_index(self, file_index, index_rate)
    if needs_ref:
        return pipeline_hip._rvcmi_original(self, model, net_g, sid, audio, times, f0_up_key, f0_method, file_index, index_rate, if_f0,
                                            filter_radius, tgt_sr, resample_sr, rms_mix_rate, version, protect, f0_file)
    dev = torch.device(self.device)
    sid = torch.tensor(sid, device=dev).unsqueeze(0).long()
    # segments of a long input: one net_g.infer call for all of them when the synthesizer is the HIP one (infer_segments above);
    # RVCMI_PIPELINE_BATCH=0 (or a foreign net_g) keeps the reference's call-per-segment order
    can_batch = _ragged_capable(net_g) and os.environ.get("RVCMI_PIPELINE_BATCH", "1") != "0"
    pending, segs = [], []

    def collect(a0, pt, pf):
        pending.append((a0, pt, pf))

    audio, ncuts = _prepare_file(self, model, sid, audio, times, f0_up_key, f0_method, if_f0, filter_radius, version, f0_file, collect)
    if can_batch and ncuts > 0:
        from time import time

        t0 = time()
        items = blend_segments([hubert_device(self, model, a0, pt, pf, version) for a0, pt, pf in pending], index, index_rate, protect)
        times[0] += time() - t0
        segs = infer_segments(net_g, sid, items, times)
    else:
        segs = [vc_device(self, model, net_g, sid, a0, pt, pf, times, index, index_rate, version, protect) for a0, pt, pf in pending]
    return _finish_file(self, segs, audio, tgt_sr, resample_sr, rms_mix_rate)


def convert_files(self, model, net_g, sid, audios, times, f0_up_key, f0_method, file_index, index_rate, if_f0, filter_radius, tgt_sr,
                  resample_sr, rms_mix_rate, version, protect, f0_files=None):
    """``Pipeline.pipeline`` for SEVERAL inputs in one go -- the body of ``VC.vc_multi``'s loop (infer/modules/vc/modules.py:201-266:
    ``load_audio`` + ``vc_single`` -> ``pipeline`` per file of a folder) with the files batched on the GPU:

      1. per file, as ``pipeline`` does it: high-pass, cut points, f0 (RMVPE on the device), HuBERT per segment;
      2. ONE retrieval call for the HuBERT frames of every segment of every file (``blend_segments``: one coarse pass and one
         list-major scan per ``MAX_BATCH_QUERIES`` frames instead of one per segment);
      3. the segments of all files through ``net_g.infer`` as ragged batches (``infer_segments``, at most ``MAX_BATCH_FRAMES`` padded
         frames per call): a 10 s clip alone fills the chip only in part of the generator (stage 0: 355 tiles on 256 CUs), a batch of
         them does throughout (BASELINE configs[2]);
      4. per file: trim + concatenate, ``change_rms``, scaling, one copy to the host.

    ``audios``: list of 16 kHz float waveforms (``load_audio`` output); the other arguments are ``Pipeline.pipeline``'s, shared by all
    files (``vc_multi`` passes the same sid / f0 method / index / rates for the whole folder); ``f0_files``: None or one entry per
    input.  -> list of numpy arrays in input order.  Every item is computed exactly as its own call would compute it (same noise
    draws in the same order, ragged batch items = separate calls; bit-equal with the shape-dependent kernel choices pinned --
    generator ``RB_STREAM`` / ``NO_RB_SPLIT``, front ``FR_NJ`` / ``FR_FFN_SPLIT`` -- and equal to operand rounding otherwise), so the result does not depend on how the files are grouped.  A synthesizer that is not the HIP one, or an
    index only real faiss reads, or ``RVCMI_PIPELINE_BATCH=0``, takes the plain per-file loop over ``self.pipeline``."""
    audios = list(audios)
    f0_files = list(f0_files) if f0_files is not None else [None] * len(audios)
    if len(f0_files) != len(audios):
        raise ValueError("f0_files must hold one entry per input")
    if if_f0 == 2 and len(audios) > 1:
        raise ValueError("if_f0 == 2 hands ONE precomputed (pitch, pitchf) pair to the pipeline (pipeline.py:268-269); convert such inputs one by one")
    import os

    index, needs_ref = _open_index(self, file_index, index_rate)
    # RVCMI_PIPELINE_BATCH=0 (the switch pipeline_hip honours for a file's segments) turns cross-file batching off too
    if needs_ref or not _ragged_capable(net_g) or os.environ.get("RVCMI_PIPELINE_BATCH", "1") == "0":
        return [self.pipeline(model, net_g, sid, a, times, f0_up_key, f0_method, file_index, index_rate, if_f0, filter_radius, tgt_sr,
                              resample_sr, rms_mix_rate, version, protect, f) for a, f in zip(audios, f0_files)]
    if len(audios) > MAX_FILES_PER_GROUP:  # bound what is resident at once (HuBERT frames and waveforms of a group); same order, same draws
        res = []
        for lo in range(0, len(audios), MAX_FILES_PER_GROUP):
            res += convert_files(self, model, net_g, sid, audios[lo: lo + MAX_FILES_PER_GROUP], times, f0_up_key, f0_method, file_index, index_rate,
                                 if_f0, filter_radius, tgt_sr, resample_sr, rms_mix_rate, version, protect, f0_files[lo: lo + MAX_FILES_PER_GROUP])
        return res
    from time import time

    dev = torch.device(self.device)
    sid = torch.tensor(sid, device=dev).unsqueeze(0).long()
    raw, owner, filtered = [], [], []
    for i, (a, f0f) in enumerate(zip(audios, f0_files)):
        def collect(a0, pt, pf, i=i):
            t0 = time()
            raw.append(hubert_device(self, model, a0, pt, pf, version))
            owner.append(i)
            times[0] += time() - t0

        filtered.append(_prepare_file(self, model, sid, a, times, f0_up_key, f0_method, if_f0, filter_radius, version, f0f, collect)[0])
    t0 = time()
    items = blend_segments(raw, index, index_rate, protect)
    times[0] += time() - t0
    outs = infer_segments(net_g, sid, items, times)
    res = []
    for i in range(len(audios)):
        res.append(_finish_file(self, [o for o, w in zip(outs, owner) if w == i], filtered[i], tgt_sr, resample_sr, rms_mix_rate))
    return res
