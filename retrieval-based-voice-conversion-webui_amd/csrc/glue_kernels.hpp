// Device-resident pre-/post-glue of Pipeline.vc / Pipeline.pipeline: the small host-side
// numpy / python steps between the PyTorch-ROCm feature extractors and net_g.infer, so that nothing has to leave the GPU.
//
//   x2 nearest interpolation + protect mix   infer/modules/vc/pipeline.py:140-159
//   RMVPE salience -> cents -> Hz            rvc/f0/rmvpe.py:119-164
//   _resize_f0 / _interpolate_f0             rvc/f0/f0.py:31-78
//   post_process (key shift, mel binning)    rvc/f0/gen.py:10-41
//   int16 scaling                            infer/modules/vc/pipeline.py:355-359
//
// These are HBM/latency-bound byte shuffles and short fp64 recurrences -- no MFMA here.  fp64 wherever numpy computes in
// float64 (the whole f0 chain), fp32 with contraction off wherever torch computes in float32.
#pragma once
#include <hip/hip_runtime.h>

#include "exact_fp.hpp"
#include <stdint.h>

namespace rvcmi {

// feats = blend(search) per query row i (pipeline.py:129-138, see k_blend), then for output frame t < p_len:
//   x = row[t / 2]                               F.interpolate(scale_factor=2), nearest   (pipeline.py:140-144)
//   out[t] = x * pf + feats0[t/2] * (1 - pf),  pf = pitchf[t] >= 1 ? 1 : protect         (pipeline.py:153-158)
// One block per QUERY row (both of its output frames): the blended row is computed once, in registers.
// P == nullptr: no retrieval (index_rate == 0 / no index): the row is the input row.
static __global__ void __launch_bounds__(256) k_blend_expand(const float* __restrict__ feats, const float* __restrict__ D,
                                                             const int64_t* __restrict__ P, const float* __restrict__ vecs,
                                                             int d, int k, int64_t pos_last, float rate, float omr,
                                                             const int* __restrict__ any_short, int skip_if_short,
                                                             const float* __restrict__ pitchf, float protect, int64_t p_len,
                                                             int reps, float* __restrict__ out) {
#pragma clang fp contract(off)
    const int64_t qi = blockIdx.x;
    const bool blend = P != nullptr && !(skip_if_short && *any_short);
    float w[8];
    if (blend) {
        for (int s = 0; s < k; ++s) {
            const float inv = div_rn(1.0f, D[qi * k + s]);
            w[s] = mul_rn(inv, inv);
        }
        float sum;
        if (k == 8) {
            sum = add_rn(add_rn(add_rn(w[0], w[1]), add_rn(w[2], w[3])),
                            add_rn(add_rn(w[4], w[5]), add_rn(w[6], w[7])));
        } else {
            sum = w[0];
            for (int s = 1; s < k; ++s) sum = add_rn(sum, w[s]);
        }
        for (int s = 0; s < k; ++s) w[s] = div_rn(w[s], sum);
    }
    for (int e = threadIdx.x; e < d; e += 256) {
        const float f = feats[qi * d + e];
        float x = f;
        if (blend) {
            float acc = 0.f;
            for (int s = 0; s < k; ++s) {
                int64_t p = P[qi * k + s];
                if (p < 0) p = pos_last;
                const float prod = mul_rn(vecs[p * d + e], w[s]);
                acc = s == 0 ? prod : add_rn(acc, prod);
            }
            x = add_rn(mul_rn(acc, rate), mul_rn(omr, f));
        }
        for (int r = 0; r < reps; ++r) {
            const int64_t t = qi * reps + r;
            if (t >= p_len) break;
            float o = x;
            if (pitchf) {
                const float pf = pitchf[t] < 1.f ? protect : 1.f;  // pitchff[pitchf > 0] = 1; pitchff[pitchf < 1] = protect
                o = add_rn(mul_rn(x, pf), mul_rn(f, sub_rn(1.f, pf)));
            }
            out[t * d + e] = o;
        }
    }
}

// RMVPE._to_local_average_cents + _decode (rmvpe.py:119-164): one wave per frame.
//   center = argmax(salience[f]) (first maximum); 9-bin window, zero-padded salience AND zero-padded cents table;
//   cents = sum(sal*cm) / sum(sal)   product in float64, weight sum in float32 (numpy's pairwise order for n = 9);
//   0 where max <= thred;  f0 = 10 * 2^(cents/1200), 0 where that equals 10.
static __global__ void __launch_bounds__(256) k_rmvpe_decode(const float* __restrict__ sal, int n, int nbins, float thred,
                                                             double* __restrict__ f0) {
#pragma clang fp contract(off)
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (f >= n) return;
    const float* s = sal + (size_t)f * nbins;
    float best = -INFINITY;
    int bi = 0;
    for (int i = lane; i < nbins; i += 64) {
        const float v = s[i];
        if (v > best) {  // strictly greater keeps the first maximum inside a lane (indices ascend)
            best = v;
            bi = i;
        }
    }
    for (int off = 32; off >= 1; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (ob > best || (ob == best && oi < bi)) {
            best = ob;
            bi = oi;
        }
    }
    if (lane == 0) {
        double prod[9];
        float wv[9];
        for (int w = 0; w < 9; ++w) {
            const int b = bi - 4 + w;
            const bool in = b >= 0 && b < nbins;
            const float sv = in ? s[b] : 0.f;
            const double cm = in ? 20.0 * (double)b + 1997.3794084376191 : 0.0;
            wv[w] = sv;
            prod[w] = (double)sv * cm;
        }
        // numpy pairwise sum of 9 elements: 8 accumulators combined as a tree, then the 9th
        const double ps = (((prod[0] + prod[1]) + (prod[2] + prod[3])) + ((prod[4] + prod[5]) + (prod[6] + prod[7]))) + prod[8];
        const float ws = add_rn(add_rn(add_rn(add_rn(wv[0], wv[1]), add_rn(wv[2], wv[3])),
                                             add_rn(add_rn(wv[4], wv[5]), add_rn(wv[6], wv[7]))), wv[8]);
        double cents = ps / (double)ws;
        if (best <= thred) cents = 0.0;
        double hz = 10.0 * pow(2.0, cents / 1200.0);
        if (hz == 10.0) hz = 0.0;
        f0[f] = hz;
    }
}

// _resize_f0 (np.interp with NaN for unvoiced, f0.py:68-78) -> _interpolate_f0 (f0.py:31-66, sequential, with its
// aliasing quirks) -> post_process (gen.py:18,34-41).  Single block; src [n] fp64 -> pitch [p_len] int64, pitchf [p_len] fp32.
// Work arrays a [n] and r [p_len] (fp64): dynamic LDS when they fit (<= 20480 frames), otherwise the caller's global
// buffers ga / gr -- a whole file's f0 is computed in ONE call by the reference (pipeline.py:260-266; a 3-minute song is
// ~36k frames), so there must be no length limit.  ga may alias src (in-place NaN marking); gr may be the `pitch` output
// buffer itself (same size: the last loop reads r[i] and writes pitch[i] from the same thread).
static __global__ void __launch_bounds__(256) k_f0_post(const double* src, int n, int p_len, int do_resize,
                                                        int do_interp, double key_mul, double mel_min, double mel_max,
                                                        int64_t* pitch, float* __restrict__ pitchf, double* ga, double* gr) {
#pragma clang fp contract(off)
    extern __shared__ double sm[];
    double* a = ga ? ga : sm;           // [n]   source with NaN for unvoiced
    double* r = gr ? gr : sm + n;       // [p_len]
    const double NaN = __longlong_as_double(0x7ff8000000000000LL);
    if (do_resize) {
        for (int i = threadIdx.x; i < n; i += 256) {
            const double v = src[i];
            a[i] = v < 0.001 ? NaN : v;
        }
    }
    __syncthreads();
    if (do_resize) {
        // np.interp(np.arange(0, n*p_len, n) / p_len, np.arange(n), source); then nan_to_num
        for (int i = threadIdx.x; i < p_len; i += 256) {
            const double x = (double)((long long)i * n) / (double)p_len;
            double res;
            if (x > (double)(n - 1)) {
                res = a[n - 1];  // right = fp[-1]
            } else {
                const int j = (int)floor(x);
                if (j == n - 1 || (double)j == x) {
                    res = a[j];
                } else {
                    const double slope = (a[j + 1] - a[j]) / ((double)(j + 1) - (double)j);
                    res = slope * (x - (double)j) + a[j];
                    if (isnan(res)) {  // "if we get nan in one direction, try the other" (numpy arr_interp)
                        res = slope * (x - (double)(j + 1)) + a[j + 1];
                        if (isnan(res) && a[j] == a[j + 1]) res = a[j];
                    }
                }
            }
            if (isnan(res)) res = 0.0;  // np.nan_to_num (values are finite or NaN here)
            r[i] = res;
        }
    } else {
        for (int i = threadIdx.x; i < p_len; i += 256) r[i] = i < n ? src[i] : 0.0;
    }
    __syncthreads();
    if (do_interp && threadIdx.x == 0) {
        // sequential emulation of F0Predictor._interpolate_f0 (ip_data aliases data).  Filled gaps become positive, so the
        // scan jumps over them; a trailing fill ends the pass (re-running it cannot change anything).
        const int N = p_len;
        double last = 0.0;
        int i = 0;
        while (i < N) {
            if (r[i] <= 0.0) {
                int j = i + 1;
                for (int jj = i + 1; jj < N; ++jj) {
                    j = jj;
                    if (r[jj] > 0.0) break;
                }
                if (j < N - 1) {
                    if (last > 0.0) {
                        const double step = (r[j] - r[i - 1]) / (double)(j - i);
                        for (int k = i; k < j; ++k) r[k] = r[i - 1] + step * (double)(k - i + 1);
                    } else {
                        for (int k = i; k < j; ++k) r[k] = r[j];
                    }
                    last = r[j - 1] > 0.0 ? r[j - 1] : last;  // what the per-element pass would have left behind
                    i = j;
                } else {
                    for (int k = i; k < N; ++k) r[k] = last;
                    break;
                }
            } else {
                last = r[i];
                ++i;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < p_len; i += 256) {
        const double f = r[i] * key_mul;  // np.multiply(f0, pow(2, f0_up_key / 12))   (read BEFORE pitch[i] is written: gr may alias it)
        double mel = 1127.0 * log(1.0 + f / 700.0);
        if (mel > 0.0) mel = (mel - mel_min) * 254.0 / (mel_max - mel_min) + 1.0;
        if (mel <= 1.0) mel = 1.0;
        if (mel > 255.0) mel = 255.0;
        pitch[i] = (int64_t)rint(mel);
        pitchf[i] = (float)f;
    }
}

// audio_max = abs(audio).max() / 0.99; max_int16 = 32768; if audio_max > 1: max_int16 /= audio_max; audio *= max_int16
// (pipeline.py:355-359).  Two launches: per-block maxima, then scale (every block re-reduces the partials).
static __global__ void __launch_bounds__(256) k_absmax_partial(const float* __restrict__ x, int64_t n, float* __restrict__ part) {
    __shared__ float red[256];
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
static __global__ void __launch_bounds__(256) k_scale_int16_range(float* __restrict__ x, int64_t n, const float* __restrict__ part,
                                                                  int nparts) {
#pragma clang fp contract(off)
    __shared__ float red[256];
    float m = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) m = fmaxf(m, part[i]);
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    // numpy: float32 max / python float 0.99 -> float64; 32768 / that -> float64; multiply casts the scalar to float32
    const double amax = (double)red[0] / 0.99;
    double mi = 32768.0;
    if (amax > 1.0) mi /= amax;
    const float sc = (float)mi;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] = mul_rn(x[i], sc);
}

// SOLA (gui.py:1057-1090, "SOLA algorithm from DDSP-SVC"): find the offset in [0, Ls] at which the new chunk best continues
// the previous chunk's tail (normalised cross-correlation), cut there, cross-fade Lb samples, keep the next tail.
//   cor_nom[o] = sum_i x[o+i] * buf[i],  cor_den[o] = sqrt(sum_i x[o+i]^2 + 1e-8),  offset = argmax(cor_nom / cor_den)
//   y = x[offset:];  y[:Lb] = y[:Lb] * fade_in + buf * fade_out;  buf <- y[block : block + Lb];  out <- y[:block]
// The two sums are accumulated in fp64 and rounded once (the reference's F.conv1d leaves the fp32 summation order to the
// backend); everything after the argmax is the reference's fp32 expression, unfused.  One block.
static __global__ void __launch_bounds__(256) k_sola(const float* __restrict__ x, float* __restrict__ buf, int Lb, int Ls,
                                                     const float* __restrict__ fade_in, const float* __restrict__ fade_out,
                                                     int block, float* __restrict__ out, int* __restrict__ offset_out) {
#pragma clang fp contract(off)
    extern __shared__ float sl[];
    float* xs = sl;             // [Lb + Ls]
    float* bs = sl + Lb + Ls;   // [Lb]
    __shared__ float rv[256];
    __shared__ int ri[256];
    for (int i = threadIdx.x; i < Lb + Ls; i += 256) xs[i] = x[i];
    for (int i = threadIdx.x; i < Lb; i += 256) bs[i] = buf[i];
    __syncthreads();
    float best = -INFINITY;
    int bo = 0;
    for (int o = threadIdx.x; o <= Ls; o += 256) {
        double nom = 0.0, en = 0.0;
        for (int i = 0; i < Lb; ++i) {
            const double v = (double)xs[o + i];
            nom += v * (double)bs[i];
            en += v * v;
        }
        const float r = div_rn((float)nom, sqrtf(add_rn((float)en, 1e-8f)));
        if (r > best) {
            best = r;
            bo = o;
        }
    }
    rv[threadIdx.x] = best;
    ri[threadIdx.x] = bo;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) {
            const float ov = rv[threadIdx.x + s];
            const int oi = ri[threadIdx.x + s];
            if (ov > rv[threadIdx.x] || (ov == rv[threadIdx.x] && oi < ri[threadIdx.x])) {  // torch.argmax: first maximum
                rv[threadIdx.x] = ov;
                ri[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    const int off = ri[0];
    if (threadIdx.x == 0 && offset_out) *offset_out = off;
    auto y = [&](int p) {  // the shifted, cross-faded chunk at position p
        const float v = x[off + p];
        return p < Lb ? add_rn(mul_rn(v, fade_in[p]), mul_rn(bs[p], fade_out[p])) : v;
    };
    for (int i = threadIdx.x; i < block; i += 256) out[i] = y(i);
    for (int i = threadIdx.x; i < Lb; i += 256) buf[i] = y(block + i);  // old tail is in LDS: safe to overwrite
}

// ------------------------------------------------------------------------------------------------
// change_rms (infer/modules/vc/pipeline.py:26-46): the output's loudness envelope mixed with the input's.
//   rms_i = librosa.feature.rms(y, frame_length = sr//2*2, hop_length = sr//2)   (centred frames, zero padding -- the
//           default of the librosa >= 0.10.2 the reference requires), one point per half second;
//   both envelopes linearly interpolated to len(data2) (F.interpolate, align_corners=False); rms2 = max(rms2, 1e-6);
//   data2 *= rms1^(1-rate) * rms2^(rate-1).
// The frame energy is accumulated in fp64 (numpy sums the float32 squares in float32; its summation order for this strided
// view is not pinned by anything in the reference, see oracle/glue_oracle.py:change_rms) and rounded to fp32 before the sqrt.
// ------------------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(256) k_frame_rms(const float* __restrict__ y, int64_t n, int frame, int hop, int nframes,
                                                          float* __restrict__ rms) {
    __shared__ double red[256];
    const int f = blockIdx.x;
    if (f >= nframes) return;
    const int64_t lo = (int64_t)f * hop - frame / 2;
    double acc = 0.0;
    for (int i = threadIdx.x; i < frame; i += 256) {
        const int64_t t = lo + i;
        if (t >= 0 && t < n) {
            const float v = y[t];
            acc += (double)mul_rn(v, v);  // abs2 in float32, like librosa's util.abs2(dtype=float32)
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s2 = 128; s2 >= 1; s2 >>= 1) {
        if ((int)threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
        __syncthreads();
    }
    if (threadIdx.x == 0) rms[f] = sqrtf((float)(red[0] / (double)frame));
}

__device__ __forceinline__ float interp_linear_1d(const float* __restrict__ r, int nin, int64_t nout, int64_t o) {
    // F.interpolate(mode="linear", align_corners=False): src = scale*(o+0.5)-0.5 clamped at 0, scale = nin/nout (fp32)
    const float scale = (float)nin / (float)nout;
    float src = sub_rn(mul_rn(scale, add_rn((float)o, 0.5f)), 0.5f);
    if (src < 0.f) src = 0.f;
    int i0 = (int)src;
    if (i0 > nin - 1) i0 = nin - 1;
    const int i1 = i0 + (i0 < nin - 1 ? 1 : 0);
    const float l1 = sub_rn(src, (float)i0);
    const float l0 = sub_rn(1.f, l1);
    return add_rn(mul_rn(l0, r[i0]), mul_rn(l1, r[i1]));
}

static __global__ void __launch_bounds__(256) k_change_rms(float* __restrict__ data2, int64_t n2, const float* __restrict__ rms1, int nf1,
                                                           const float* __restrict__ rms2, int nf2, float e1, float e2) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n2) return;
    const float r1 = interp_linear_1d(rms1, nf1, n2, t);
    float r2 = interp_linear_1d(rms2, nf2, n2, t);
    r2 = fmaxf(r2, 1e-6f);
    data2[t] = mul_rn(data2[t], mul_rn(powf(r1, e1), powf(r2, e2)));
}

// Polyphase FIR resampling as torchaudio.transforms.Resample applies it (rtrvc.py:248-259: the formant-shifted block comes out
// of the generator at upp_res samples per 10 ms and is brought back to tgt_sr/100): out[j*nf + p] = sum_k w[p][k] * xpad[j*of + k],
// xpad = x padded with `width` zeros in front (and zeros behind), k < K = 2*width + of.  One thread per output sample, fp32,
// taps added in ascending k (torchaudio's F.conv1d leaves the order to the backend).  HBM/latency-bound: n_out * K MACs,
// K ~ 430 for the 423 -> 400 ratio of a +1 semitone shift at 40 kHz; the 400 x 437 coefficient table stays in L2.
static __global__ void __launch_bounds__(256) k_resample_poly(const float* __restrict__ x, int64_t n, const float* __restrict__ w, int of, int nf,
                                                       int K, int width, float* __restrict__ out, int64_t n_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_out) return;
    const int64_t j = i / nf;
    const int p = (int)(i - j * nf);
    const float* wp = w + (size_t)p * K;
    const int64_t base = j * of - width;  // index of tap 0 in the unpadded signal
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
        const int64_t t = base + k;
        const float xv = (t >= 0 && t < n) ? x[t] : 0.f;
        acc = fmaf(wp[k], xv, acc);
    }
    out[i] = acc;
}

}  // namespace rvcmi
