"""TEST INFRASTRUCTURE ONLY -- numpy/torch-CPU restatement of the host-side glue of the reference pipeline
(SURVEY.md section 8f row 2), each function citing the reference lines it follows:

* ``expand_protect``      infer/modules/vc/pipeline.py:140-159   (x2 nearest interpolation, protect mix)
* ``rmvpe_decode``        rvc/f0/rmvpe.py:119-164                (salience -> local-average cents -> Hz)
* ``resize_f0``           rvc/f0/f0.py:68-78
* ``interpolate_f0``      rvc/f0/f0.py:31-66
* ``post_process``        rvc/f0/gen.py:10-41, constants :70-73,131-132
* ``scale_int16_range``   infer/modules/vc/pipeline.py:355-359
* ``sola``                gui.py:1057-1090 (the non-phase-vocoder branch)

``oracle/make_golden.py`` pins the f0 functions to the reference's own implementations (``F0Predictor``, ``RMVPE``
methods, ``post_process`` with numba stubbed out) on seeded inputs and stores the results in ``tests/golden/glue_f0.npz``.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def expand_protect(feats: torch.Tensor, feats0: torch.Tensor, pitchf, protect: float, p_len: int) -> torch.Tensor:
    """feats = blended features [1, nq, d], feats0 = features before the blend; pitchf [1, >= p_len] or None."""
    feats = F.interpolate(feats.permute(0, 2, 1), scale_factor=2).permute(0, 2, 1)  # :140
    use = pitchf is not None and protect < 0.5
    if use:
        feats0 = F.interpolate(feats0.permute(0, 2, 1), scale_factor=2).permute(0, 2, 1)  # :142-144
    if feats.shape[1] < p_len:  # :147-151
        p_len = feats.shape[1]
    if use:
        pitchf = pitchf[:, :p_len]
        pitchff = pitchf.clone()
        pitchff[pitchf > 0] = 1
        pitchff[pitchf < 1] = protect
        pitchff = pitchff.unsqueeze(-1)
        feats = feats[:, :p_len] * pitchff + feats0[:, :p_len] * (1 - pitchff)  # :153-158 (infer truncates to p_len through x_mask)
    return feats[:, :p_len]


def rmvpe_decode(salience: np.ndarray, thred: float = 0.03) -> np.ndarray:
    """rmvpe.py:119-164.  salience [n, 360] float32 -> f0 [n] float64 (0 = unvoiced)."""
    cents_mapping = np.pad(20 * np.arange(360) + 1997.3794084376191, (4, 4))  # :62-63
    center = np.argmax(salience, axis=1)
    sal = np.pad(salience, ((0, 0), (4, 4)))
    center = center + 4
    rows = np.arange(sal.shape[0])[:, None]
    win = center[:, None] + np.arange(-4, 5)[None, :]
    todo_salience = sal[rows, win]            # [n, 9] float32
    todo_cents = cents_mapping[win]           # [n, 9] float64
    product_sum = np.sum(todo_salience * todo_cents, 1)
    weight_sum = np.sum(todo_salience, 1)
    devided = product_sum / weight_sum
    maxx = np.max(sal, axis=1)
    devided[maxx <= thred] = 0
    f0 = 10 * (2 ** (devided / 1200))
    f0[f0 == 10] = 0
    return f0


def resize_f0(x: np.ndarray, target_len: int) -> np.ndarray:
    # f0.py:68-78
    source = np.array(x)
    source[source < 0.001] = np.nan
    target = np.interp(np.arange(0, len(source) * target_len, len(source)) / target_len, np.arange(0, len(source)), source)
    return np.nan_to_num(target)


def interpolate_f0(f0: np.ndarray) -> np.ndarray:
    """f0.py:31-66, including its aliasing: ``ip_data`` IS ``data``, so filled values are seen by later iterations."""
    data = np.array(f0, dtype=np.float64).reshape(-1)
    n = data.size
    last_value = 0.0
    for i in range(n):
        if data[i] <= 0.0:
            j = i + 1
            for j in range(i + 1, n):
                if data[j] > 0.0:
                    break
            if j < n - 1:
                if last_value > 0.0:
                    step = (data[j] - data[i - 1]) / float(j - i)
                    for k in range(i, j):
                        data[k] = data[i - 1] + step * (k - i + 1)
                else:
                    for k in range(i, j):
                        data[k] = data[j]
            else:
                for k in range(i, n):
                    data[k] = last_value
        else:
            last_value = data[i]
    return data


def post_process(f0: np.ndarray, f0_up_key: int):
    """gen.py:18,34-41 with f0_min/f0_max = 50/1100 (:70-71) -> (f0_coarse int, f0 float64)."""
    f0_mel_min = 1127 * math.log(1 + 50 / 700)
    f0_mel_max = 1127 * math.log(1 + 1100 / 700)
    f0 = np.multiply(f0, pow(2, f0_up_key / 12))
    f0_mel = 1127 * np.log(1 + f0 / 700)
    f0_mel[f0_mel > 0] = (f0_mel[f0_mel > 0] - f0_mel_min) * 254 / (f0_mel_max - f0_mel_min) + 1
    f0_mel[f0_mel <= 1] = 1
    f0_mel[f0_mel > 255] = 255
    return np.rint(f0_mel).astype(np.int32), f0


def rmvpe_f0(salience: np.ndarray, p_len: int, f0_up_key: int, thred: float = 0.03):
    """What Generator.calculate(x, p_len, key, "rmvpe", ...) returns from the salience map on (gen.py:103-113, 131-133,
    rmvpe.py:115-117) plus the casts of pipeline.py:270-277: (pitch int64 [p_len], pitchf float32 [p_len])."""
    f0 = interpolate_f0(resize_f0(rmvpe_decode(salience, thred), p_len))
    coarse, f0 = post_process(f0, f0_up_key)
    return coarse[:p_len].astype(np.int64), f0[:p_len].astype(np.float32)


def frame_rms(y: np.ndarray, frame_length: int, hop_length: int) -> np.ndarray:
    """``librosa.feature.rms(y=y, frame_length=..., hop_length=...)[0]`` as librosa >= 0.10.2 (requirements/main.txt:5)
    documents it: ``center=True, pad_mode="constant"`` (zero padding of frame_length//2 on both sides), frames
    ``y[i*hop : i*hop + frame_length]``, ``sqrt(mean(abs2(frame, dtype=float32)))``.

    PARITY UNPINNED: librosa is not installable here and the reference holds no fixture for this function.  The float32
    squares are summed here in float64 (librosa/numpy sum them in float32, in an order that depends on numpy's reduction
    loop for the strided frame view); the difference is below 1e-6 relative."""
    y = np.asarray(y, dtype=np.float32)
    pad = frame_length // 2
    yp = np.pad(y, (pad, pad), mode="constant")
    n_frames = 1 + (len(yp) - frame_length) // hop_length
    out = np.empty(n_frames, np.float32)
    for i in range(n_frames):
        fr = yp[i * hop_length: i * hop_length + frame_length]
        out[i] = np.sqrt(np.float32((fr * fr).astype(np.float64).sum() / frame_length))
    return out


def change_rms(data1: np.ndarray, sr1: int, data2: np.ndarray, sr2: int, rate: float) -> np.ndarray:
    """pipeline.py:26-46 with the same torch calls (F.interpolate linear, torch.pow with a float32 0-dim exponent)."""
    import torch.nn.functional as F

    rms1 = torch.from_numpy(frame_rms(data1, sr1 // 2 * 2, sr1 // 2)[None])
    rms2 = torch.from_numpy(frame_rms(data2, sr2 // 2 * 2, sr2 // 2)[None])
    rms1 = F.interpolate(rms1.unsqueeze(0), size=data2.shape[0], mode="linear").squeeze()
    rms2 = F.interpolate(rms2.unsqueeze(0), size=data2.shape[0], mode="linear").squeeze()
    rms2 = torch.max(rms2, torch.zeros_like(rms2) + 1e-6)
    out = np.array(data2, dtype=np.float32)
    out *= (torch.pow(rms1, torch.tensor(1 - rate)) * torch.pow(rms2, torch.tensor(rate - 1))).numpy()
    return out


def scale_int16_range(audio: np.ndarray) -> np.ndarray:
    # pipeline.py:355-359
    audio = np.array(audio, dtype=np.float32)
    audio_max = np.abs(audio).max() / 0.99
    max_int16 = 32768
    if audio_max > 1:
        max_int16 /= audio_max
    np.multiply(audio, max_int16, audio)
    return audio


def sola(infer_wav: torch.Tensor, sola_buffer: torch.Tensor, fade_in: torch.Tensor, fade_out: torch.Tensor, block_frame: int,
         search_frame: int):
    """gui.py:1057-1090 verbatim in torch-CPU (use_pv False).  Returns (out block, new sola_buffer, offset)."""
    Lb = sola_buffer.numel()
    infer_wav = infer_wav.clone()
    conv_input = infer_wav[None, None, : Lb + search_frame]
    cor_nom = F.conv1d(conv_input, sola_buffer[None, None, :])
    cor_den = torch.sqrt(F.conv1d(conv_input ** 2, torch.ones(1, 1, Lb)) + 1e-8)
    sola_offset = int(torch.argmax(cor_nom[0, 0] / cor_den[0, 0]))
    infer_wav = infer_wav[sola_offset:]
    infer_wav[:Lb] *= fade_in
    infer_wav[:Lb] += sola_buffer * fade_out
    new_buf = infer_wav[block_frame: block_frame + Lb].clone()
    return infer_wav[:block_frame].clone(), new_buf, sola_offset


def sinc_resample(x: np.ndarray, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99) -> np.ndarray:
    """PARITY UNPINNED (torchaudio is not installable offline).  ``torchaudio.transforms.Resample(orig_freq, new_freq,
    dtype=float32)`` (rtrvc.py:251-259) restated independently of the product from torchaudio's published
    ``_get_sinc_resample_kernel`` / ``_apply_sinc_resample_kernel``: hann-windowed sinc table in float32, zero padding
    ``(width, width + orig)``, strided correlation, ``ceil(new * n / orig)`` samples.  The sums run in float64."""
    import math

    g = math.gcd(int(orig_freq), int(new_freq))
    of, nf = int(orig_freq) // g, int(new_freq) // g
    base = np.float32(min(of, nf) * rolloff)
    width = int(math.ceil(lowpass_filter_width * of / (min(of, nf) * rolloff)))
    idx = (np.arange(-width, width + of, dtype=np.float32) / np.float32(of))[None, :]
    t = (np.arange(0, -nf, -1, dtype=np.float32) / np.float32(nf))[:, None] + idx
    t = np.clip((t * base).astype(np.float32), -lowpass_filter_width, lowpass_filter_width).astype(np.float32)
    window = (np.cos((t * np.float32(math.pi) / np.float32(lowpass_filter_width) / np.float32(2)).astype(np.float32)) ** 2).astype(np.float32)
    tp = (t * np.float32(math.pi)).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        sinc = np.where(tp == 0, np.float32(1), (np.sin(tp) / tp).astype(np.float32))
    kern = (sinc * window * np.float32(base / np.float32(of))).astype(np.float32)          # [nf, K]
    x = np.asarray(x, dtype=np.float32)
    n = x.shape[-1]
    xp = np.concatenate([np.zeros(width, np.float32), x, np.zeros(width + of, np.float32)])
    K = kern.shape[1]
    nblk = (xp.shape[0] - K) // of + 1
    frames = np.lib.stride_tricks.sliding_window_view(xp, K)[::of][:nblk]             # [nblk, K]
    out = (frames.astype(np.float64) @ kern.astype(np.float64).T).reshape(-1)            # [nblk * nf], phase fastest
    return out[: -(-nf * n // of)].astype(np.float32)
