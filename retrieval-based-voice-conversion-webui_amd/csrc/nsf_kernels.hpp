// Device code of the NSF-HiFi-GAN generator for gfx950 (CDNA4).
//
// Internal activation layout is CHANNELS-LAST  [B][L][C]  (fp32 residual stream; the activated
// MFMA operand copies are OpT = bf16 / fp16).  With channels last, one MFMA B-fragment of a k-tap
// dilated convolution is "8 consecutive input channels of ONE time row": a single 16-byte LDS
// read, and moving to the next tap is a constant row offset -- no im2col is ever materialised.
//
// Reference semantics reproduced here (file:line in /root/reference):
//   sine source            rvc/layers/generators.py:148-194
//   tanh(Linear(1,1))      rvc/layers/nsf.py:57-61
//   conv_pre + cond        rvc/layers/nsf.py:164-166
//   ups / noise_convs      rvc/layers/nsf.py:169-174
//   ResBlock1              rvc/layers/residuals.py:68-85
//   stage mean, post, tanh rvc/layers/nsf.py:186-189
#pragma once
#include <hip/hip_runtime.h>

#include "exact_fp.hpp"
#include <stdint.h>

namespace rvcmi {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

// Ragged batches (rvcmi_nsf_forward `lengths`): item b of a batch holds lens[b] <= T valid frames and is computed EXACTLY as a
// separate call of that length would compute it -- every layer zero-pads its input beyond the item's own last row (the padded
// batch of the reference lets the rows behind a short item leak into its tail through every receptive field).  Each kernel
// replaces its row count by the item's: rows(lens, b, mul, Lmax) = lens[b] * mul (mul = the stage's upsampling factor so far);
// buffer strides stay those of the longest item.
// (clamped on BOTH sides: a device-resident lengths tensor is not validated on the host -- no sync -- and lens[b] <= 0 would make the
//  clamped staging loads `min(max(row, 0), rows - 1)` read row -1 of the stream, out of bounds for item 0)
__device__ __forceinline__ int item_rows(const int* __restrict__ lens, int b, int mul, int Lmax) {
    return lens ? max(min(Lmax, lens[b] * mul), min(Lmax, mul)) : Lmax;
}

// ------------------------------------------------------------------------------------------------
// Sine source (generators.py:148-194) + source module (nsf.py:57-61)
// ------------------------------------------------------------------------------------------------

// phase[b][t] = fmod(float(sum_{tau<t} w_tau), 1), w_tau = fmod(f0/sr*upp + 0.5, 1) - 0.5.
// torch's CPU cumsum accumulates fp32 inputs in DOUBLE and rounds each prefix to fp32
// (verified against torch 2.10); a wave-level scan in fp64 reproduces that to the last bit except
// on exact rounding boundaries.  One block of 256 threads per utterance (the frames' slow fmod / divide chain is spread
// over 4 waves: 15 -> ~5 us for a 10 s clip; the block-level scan goes through LDS).
static __global__ void __launch_bounds__(256) k_phase_scan(const float* __restrict__ f0, float* __restrict__ phase,
                                                   int T, float sr, float upp, const int* __restrict__ lens) {
#pragma clang fp contract(off)
    __shared__ double wsum[4];
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* f = f0 + (size_t)b * T;
    float* ph = phase + (size_t)b * T;
    // (ragged batch: the item's own frame count sets the thread partition, so the fp64 prefix sums are those of a separate call)
    const int n = item_rows(lens, b, 1, T) - 1;  // increments come from frames 0..T-2
    const int per = (n + 255) / 256;
    const int beg = min(n, tid * per);
    const int end = min(n, beg + per);
    double local = 0.0;
    for (int t = beg; t < end; ++t) {
        float rad = mul_rn(div_rn(f[t], sr), upp);
        float w = sub_rn(fmodf(add_rn(rad, 0.5f), 1.0f), 0.5f);
        local += (double)w;
    }
    // inclusive wave scan of the per-thread sums, then the waves' totals through LDS
    double incl = local;
    for (int off = 1; off < 64; off <<= 1) {
        double o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    double run = incl - local;  // exclusive prefix inside the wave
    for (int w = 0; w < wave; ++w) run += wsum[w];
    if (tid == 0) ph[0] = 0.f;
    for (int t = beg; t < end; ++t) {
        float rad = mul_rn(div_rn(f[t], sr), upp);
        float w = sub_rn(fmodf(add_rn(rad, 0.5f), 1.0f), 0.5f);
        run += (double)w;
        ph[t + 1] = fmodf((float)run, 1.0f);
    }
}

// har[b][t*upp + n-1] = tanh(lw * (0.1*sin(2*pi*(f0/sr*n + phase)) * uv + amp * noise) + lb)
static __global__ void __launch_bounds__(256) k_sine_source(const float* __restrict__ f0, const float* __restrict__ phase,
                                                     const float* __restrict__ noise, float* __restrict__ har,
                                                     int T, int upp, float sr, float lw, float lb, size_t total) {
#pragma clang fp contract(off)
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    size_t frame = i / (size_t)upp;  // b*T + t
    int n = (int)(i - frame * upp) + 1;
    float f = f0[frame];
    float rad = add_rn(mul_rn(div_rn(f, sr), (float)n), phase[frame]);
    float s = mul_rn(sinf(mul_rn(6.2831855f, rad)), 0.1f);
    float uv = f > 0.f ? 1.f : 0.f;
    float amp = f > 0.f ? 0.003f : div_rn(0.1f, 3.0f);
    float nz = noise ? noise[i] : 0.f;
    float v = add_rn(mul_rn(s, uv), mul_rn(amp, nz));
    har[i] = tanhf(add_rn(mul_rn(v, lw), lb));
}

// F.interpolate(mode="linear", align_corners=False) along the last axis of [rows][Lin] -> [rows][Lout]
// (nsf.py:155-162, generators.py:76-79).
static __global__ void __launch_bounds__(256) k_interp_linear(const float* __restrict__ in, float* __restrict__ out,
                                                       int Lin, int Lout, size_t total) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    size_t row = i / (size_t)Lout;
    int o = (int)(i - row * Lout);
    float scale = (float)Lin / (float)Lout;
    float src = scale * ((float)o + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    int i0 = (int)src;
    if (i0 > Lin - 1) i0 = Lin - 1;
    int i1 = i0 + (i0 < Lin - 1 ? 1 : 0);
    float l1 = src - (float)i0;
    float l0 = 1.f - l1;
    const float* r = in + row * (size_t)Lin;
    out[i] = l0 * r[i0] + l1 * r[i1];
}

// cond(g): a 1x1 conv over a length-1 sequence = one GEMV per utterance (nsf.py:165-166).
// One wave per output channel: the weight row is read coalesced and reduced with xor-shuffles.
static __global__ void __launch_bounds__(256) k_cond(const float* __restrict__ g, const float* __restrict__ Wc,
                                              const float* __restrict__ bc, float* __restrict__ out, int gin, int C0) {
    const int b = blockIdx.y;
    const int co = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (co >= C0) return;
    float acc = 0.f;
    for (int i = lane; i < gin; i += 64) acc = fmaf(Wc[(size_t)co * gin + i], g[(size_t)b * gin + i], acc);
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) out[(size_t)b * C0 + co] = acc + bc[co];
}

// ------------------------------------------------------------------------------------------------
// Exact-fp32 VALU kernels (RVCMI_OPERAND_F32): correctness-first reference-grade path on the GPU.
// ------------------------------------------------------------------------------------------------

enum InMode : int {
    IN_F32_ACT = 0,   // fp32 channels-last, apply v = lrelu(v / div, slope) on load
    IN_OP_RAW = 1,    // already-activated operand copy (OpT in MFMA mode, fp32 in F32 mode)
    IN_F32_CF = 2,    // fp32 channel-first [B][C][L] raw (conv_pre input, the reference layout)
    IN_HAR = 3        // the excitation viewed as frames: row t, channel c = har[t*hs - hpad + c] (c < hs), else 0
};
enum OutMode : int {
    OUT_ACT = 0,      // out = lrelu(acc + bias, slope_out) stored as operand type
    OUT_F32 = 1       // out (+)= acc + bias (+ res) (+ cb[b][co]) stored fp32
};

struct ConvArgs {
    // input
    const void* in;
    const float* in_b;  // optional 2nd / 3rd fp32 addends: v = ((in + in_b) + in_c) / div_in   (nsf.py:177-186)
    const float* in_c;
    long in_bstride;   // elements per batch item
    int Lin;           // valid input rows
    int cf_stride;     // IN_F32_CF: elements between channels (= the longest item's Lin)
    const int* lens;   // ragged batch (item_rows): Lin = lens[b] * lmul_in, Lq = lens[b] * lmul_q; nullptr = every item full length
    int lmul_in, lmul_q;
    int cin;
    int in_mode;
    int hs, hpad;      // IN_HAR: frame length (= noise-conv stride) and left padding; Lin counts SAMPLES of har
    float slope_in, div_in;
    // weights
    const void* w;     // F32: [taps][cin][cout] fp32.  MFMA: packed fragments (see pack_conv_weights)
    long w_ct_stride;  // MFMA: elements per 32-channel output tile
    const float* bias; // [cout] or nullptr
    int cout;
    // geometry: output position q, tap j reads input row  q + in_off + j*dstep
    int ntaps;         // real taps (F32) / padded taps (MFMA)
    int in_off, dstep;
    int roff;          // MFMA: LDS tile row of (q = q0, tap 0) ; tile starts at input row q0+in_off-roff
    int tile_rows;     // MFMA: rows staged in LDS
    int Lq;            // output positions
    // output
    int out_mode;
    void* out;
    long out_bstride;
    int out_C;         // channels per output row
    int out_mul, out_add;  // output row = q*out_mul + out_add  (polyphase transposed conv)
    float slope_out;
    const float* res;  // fp32 residual [B][Lq][out_C] or nullptr
    long res_bstride;
    int accumulate;    // out += ...
    const float* cb;   // per-(batch, channel) additive term [B][cout] or nullptr
    // polyphase (transposed conv): blockIdx.y selects the phase; per-phase geometry
    int nphase;
    int ph_in_off[16];
    int ph_ntaps[16];   // F32 only
    long ph_w_off[16];  // element offset into w
};

// One thread per (q, co); co fastest so that weight reads and stores coalesce.
static __global__ void __launch_bounds__(256) k_conv_f32(ConvArgs a) {
    const int b = blockIdx.z;
    const int Linb = item_rows(a.lens, b, a.lmul_in, a.Lin), Lqb = item_rows(a.lens, b, a.lmul_q, a.Lq);
    const int ph = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)Lqb * a.cout) return;
    const int q = (int)(i / a.cout);
    const int co = (int)(i - (size_t)q * a.cout);
    const int in_off = a.nphase > 1 ? a.ph_in_off[ph] : a.in_off;
    const int ntaps = a.nphase > 1 ? a.ph_ntaps[ph] : a.ntaps;
    const float* W = (const float*)a.w + (a.nphase > 1 ? a.ph_w_off[ph] : 0);
    const float* in = (const float*)a.in + (size_t)b * a.in_bstride;
    float acc = a.bias ? a.bias[co] : 0.f;
    for (int j = 0; j < ntaps; ++j) {
        const int r = q + in_off + j * a.dstep;
        if (r < 0 || r >= Linb) continue;
        const float* wj = W + (size_t)j * a.cin * a.cout + co;
        if (a.in_mode == IN_F32_CF) {
            for (int ci = 0; ci < a.cin; ++ci) acc = fmaf(in[(size_t)ci * a.cf_stride + r], wj[(size_t)ci * a.cout], acc);
        } else if (a.in_mode == IN_F32_ACT) {
            const float* row = in + (size_t)r * a.cin;
            const float* rowb = a.in_b ? a.in_b + (size_t)b * a.in_bstride + (size_t)r * a.cin : nullptr;
            const float* rowc = a.in_c ? a.in_c + (size_t)b * a.in_bstride + (size_t)r * a.cin : nullptr;
            for (int ci = 0; ci < a.cin; ++ci) {
                float v = row[ci];
                if (rowb) v = v + rowb[ci];
                if (rowc) v = v + rowc[ci];
                if (a.div_in != 1.f) v = v / a.div_in;
                acc = fmaf(lrelu(v, a.slope_in), wj[(size_t)ci * a.cout], acc);
            }
        } else {
            const float* row = in + (size_t)r * a.cin;
            for (int ci = 0; ci < a.cin; ++ci) acc = fmaf(row[ci], wj[(size_t)ci * a.cout], acc);
        }
    }
    const size_t orow = (size_t)q * a.out_mul + (a.nphase > 1 ? ph : a.out_add);
    if (a.out_mode == OUT_ACT) {
        ((float*)a.out)[(size_t)b * a.out_bstride + orow * a.out_C + co] = lrelu(acc, a.slope_out);
    } else {
        if (a.cb) acc = acc + a.cb[(size_t)b * a.cout + co];
        if (a.res) acc = acc + a.res[(size_t)b * a.res_bstride + orow * a.out_C + co];
        float* o = (float*)a.out + (size_t)b * a.out_bstride + orow * a.out_C + co;
        *o = a.accumulate ? (*o + acc) : acc;
    }
}

// x[b][t][co] += bn[co] + sum_j har[b][t*s - pad + j] * Wn[j][co]      (nsf.py:173-174)
static __global__ void __launch_bounds__(256) k_noise_add(float* __restrict__ x, const float* __restrict__ har,
                                                   const float* __restrict__ Wn /*[k][C]*/, const float* __restrict__ bn,
                                                   int L, int C, int Lh, int k, int s, int pad, const int* __restrict__ lens, int lmul,
                                                   int lhmul) {
    const int b = blockIdx.z;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)item_rows(lens, b, lmul, L) * C) return;
    const int Lhb = item_rows(lens, b, lhmul, Lh);
    const int t = (int)(i / C);
    const int co = (int)(i - (size_t)t * C);
    const float* h = har + (size_t)b * Lh;
    float acc = bn[co];
    const int base = t * s - pad;
    for (int j = 0; j < k; ++j) {
        int r = base + j;
        if (r >= 0 && r < Lhb) acc = fmaf(h[r], Wn[(size_t)j * C + co], acc);
    }
    x[((size_t)b * L + t) * C + co] += acc;
}

// Inter-stage streams Ya[j] as fp16 (option Y_F16): the three ResBlock outputs of a stage are only ever read by
// the next stage's staging loop (k_ups / k_post), which sums them, divides by 3, applies lrelu and -- k_ups -- rounds the result
// to an MFMA operand anyway.  Stored as fp16 (round to nearest even, saturating) they cost half the HBM bytes on the three
// HBM-bound consumers and half the store instructions of the producers; the fp32 residual stream INSIDE a ResBlock is untouched.
__device__ __forceinline__ _Float16 sat_h(float v) { return (_Float16)__builtin_amdgcn_fmed3f(v, -65504.f, 65504.f); }
__device__ __forceinline__ uint2 pack4_h(float a, float b, float c, float d) {
    using h4 = __attribute__((ext_vector_type(4))) _Float16;
    h4 o = {sat_h(a), sat_h(b), sat_h(c), sat_h(d)};
    return __builtin_bit_cast(uint2, o);
}
__device__ __forceinline__ void unpack8_h(const uint4 raw, float (&f)[8]) {
    using h8 = __attribute__((ext_vector_type(8))) _Float16;
    const h8 h = __builtin_bit_cast(h8, raw);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (float)h[e];
}

// out[b][t] = tanh(sum_{j,ci} lrelu(x[t-3+j][ci], 0.01) * Wp[j][ci]),  x = ((xa + xb) + xc) / div   (nsf.py:186-189)
// HBM-bound: the last stage is read exactly once with coalesced float4 loads into an LDS tile whose
// row stride C+4 floats keeps every row 16-byte aligned and the per-thread row walk (ds_read_b128, consecutive
// lanes = consecutive rows, 20- or 36-word stride) conflict-free: 56 LDS reads per output instead of 224 scalar ones.
constexpr int POST_TT = 256;  // = blockDim: one output sample per thread (a 128-row tile measured no faster)
// Loops over time tiles (any grid.x).  Measured at B = 1 (2246 tiles, 1536 resident blocks): one tile per block 40-41 us,
// grid = resident blocks 41 us, two tiles for every block 46 us -- the launch is not limited by its last half-empty round.
static __global__ void __launch_bounds__(256) k_post(const float* __restrict__ xa, const float* __restrict__ xb,
                                              const float* __restrict__ xc, const float* __restrict__ Wp /*[7][C]*/,
                                              float* __restrict__ out, int Lmax, int C, float div, int in_half,
                                              const int* __restrict__ lens, int lmul, int dbg = 0 /* timing ablation: 2 = staging only */) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* w = (float*)smem_raw;          // [7][C]
    float* tile = w + 7 * C;              // [POST_TT + 6][C + 4]
    const int b = blockIdx.y;
    const int S = C + 4;
    for (int i = threadIdx.x; i < 7 * C; i += 256) w[i] = Wp[i];
    const int C4 = C >> 2;
    const int L = item_rows(lens, b, lmul, Lmax);  // rows of this item (strides: Lmax); rows [L, Lmax) of the output are zeroed
    const size_t boff = (size_t)b * Lmax * C;
    const int ntiles = (Lmax + POST_TT - 1) / POST_TT;
    for (int ti = blockIdx.x; ti < ntiles; ti += gridDim.x) {
        const int t0 = ti * POST_TT;
        if (t0 >= L) {  // block-uniform: a tile behind the item's end
            if (t0 + (int)threadIdx.x < Lmax) out[(size_t)b * Lmax + t0 + threadIdx.x] = 0.f;
            continue;
        }
        // batched, unconditional (clamped) loads: the former per-chunk `if (in range) { load; if (xb) load; if (xc) load }` loop
        // was ~8 serial HBM round trips per thread
        constexpr int SB = 5;
        const int total = (POST_TT + 6) * C4;
        if (in_half) {  // fp16 streams (pack4_h): 8 channels per 16-byte load
            const int C8 = C >> 3;
            const int total8 = (POST_TT + 6) * C8;
            const _Float16 *ha = (const _Float16*)xa, *hb = (const _Float16*)xb, *hc = (const _Float16*)xc;
            for (int base = threadIdx.x; base < total8; base += SB * 256) {
                uint4 va[SB], vb[SB], vc[SB];
                size_t off[SB];
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    const int idx = min(base + u * 256, total8 - 1);
                    const int r = idx / C8, c8 = idx - r * C8;
                    const int tc = min(max(t0 - 3 + r, 0), L - 1);
                    off[u] = boff + (size_t)tc * C + c8 * 8;
                    va[u] = *(const uint4*)(ha + off[u]);
                }
                if (xb) {
#pragma unroll
                    for (int u = 0; u < SB; ++u) vb[u] = *(const uint4*)(hb + off[u]);
                }
                if (xc) {
#pragma unroll
                    for (int u = 0; u < SB; ++u) vc[u] = *(const uint4*)(hc + off[u]);
                }
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    const int idx = base + u * 256;
                    if (idx < total8) {
                        const int r = idx / C8, c8 = idx - r * C8;
                        const int t = t0 - 3 + r;
                        float v[8], w2[8];
                        unpack8_h(va[u], v);
                        if (xb) {
                            unpack8_h(vb[u], w2);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += w2[e];
                        }
                        if (xc) {
                            unpack8_h(vc[u], w2);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += w2[e];
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float x = v[e];
                            if (div == 3.f) x = div3_exact(x);
                            else if (div != 1.f) x = x / div;
                            x = lrelu(x, 0.01f);
                            v[e] = (t < 0 || t >= L) ? 0.f : x;
                        }
                        *(float4*)(tile + r * S + c8 * 8) = make_float4(v[0], v[1], v[2], v[3]);
                        *(float4*)(tile + r * S + c8 * 8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
                    }
                }
            }
        } else
        for (int base = threadIdx.x; base < total; base += SB * 256) {
            float4 va[SB], vb[SB], vc[SB];
            size_t off[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int idx = min(base + u * 256, total - 1);
                const int r = idx / C4, c4 = idx - r * C4;
                const int tc = min(max(t0 - 3 + r, 0), L - 1);
                off[u] = boff + (size_t)tc * C + c4 * 4;
                va[u] = *(const float4*)(xa + off[u]);
            }
            if (xb) {
#pragma unroll
                for (int u = 0; u < SB; ++u) vb[u] = *(const float4*)(xb + off[u]);
            }
            if (xc) {
#pragma unroll
                for (int u = 0; u < SB; ++u) vc[u] = *(const float4*)(xc + off[u]);
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int idx = base + u * 256;
                if (idx < total) {
                    const int r = idx / C4, c4 = idx - r * C4;
                    const int t = t0 - 3 + r;
                    float4 v = va[u];
                    if (xb) { v.x += vb[u].x; v.y += vb[u].y; v.z += vb[u].z; v.w += vb[u].w; }
                    if (xc) { v.x += vc[u].x; v.y += vc[u].y; v.z += vc[u].z; v.w += vc[u].w; }
                    if (div == 3.f) { v.x = div3_exact(v.x); v.y = div3_exact(v.y); v.z = div3_exact(v.z); v.w = div3_exact(v.w); }
                    else if (div != 1.f) { v.x = v.x / div; v.y = v.y / div; v.z = v.z / div; v.w = v.w / div; }
                    v.x = lrelu(v.x, 0.01f); v.y = lrelu(v.y, 0.01f); v.z = lrelu(v.z, 0.01f); v.w = lrelu(v.w, 0.01f);
                    if (t < 0 || t >= L) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    *(float4*)(tile + r * S + c4 * 4) = v;
                }
            }
        }
        __syncthreads();
        const int t = t0 + threadIdx.x;
        if (dbg & 2) {  // (timing ablation, wrong results: what the staging alone costs)
            if (threadIdx.x == 0) out[(size_t)b * Lmax + t0] = tile[0];
            __syncthreads();
            continue;
        }
        if (t < L) {
            float acc = 0.f;
            for (int j = 0; j < 7; ++j) {
                const float4* row = (const float4*)(tile + (threadIdx.x + j) * S);
                const float4* wj = (const float4*)(w + j * C);
                for (int c = 0; c < C4; ++c) {  // same accumulation order as the scalar loop
                    const float4 xv = row[c], wv = wj[c];
                    acc = fmaf(xv.x, wv.x, acc);
                    acc = fmaf(xv.y, wv.y, acc);
                    acc = fmaf(xv.z, wv.z, acc);
                    acc = fmaf(xv.w, wv.w, acc);
                }
            }
            out[(size_t)b * Lmax + t] = tanhf(acc);
        } else if (t < Lmax) {
            out[(size_t)b * Lmax + t] = 0.f;
        }
        __syncthreads();  // the tile is restaged by the next iteration
    }
}

// ---- k_post with register-free staging (round 6) ------------------------------------------------------------------------------
// Same arithmetic as k_post for the shipped case (three fp16 streams, div = 3); what changes is HOW the bytes arrive.  k_post's block is a
// chain load (15 x 16 B per thread held in registers) -> convert -> barrier -> conv -> barrier, and every value held across the load costs
// registers.  Here a PERSISTENT block walks its tiles and the raw fp16 rows of tile i + 1 travel global -> LDS by LDS-DMA
// (`global_load_lds_dwordx4`: no VGPR round trip, lane-linear 1 KB pieces; a tile's rows are contiguous in a [L][C] stream) while tile i
// is converted (one LDS -> LDS pass: (a + b) + c, div3_exact, lrelu 0.01, fp32) and convolved.  Ordering: the pieces a wave issued are
// retired by ITS `s_waitcnt vmcnt(0)`, then a barrier publishes all waves' pieces (the documented RAW rule for LDS-DMA); `raw` is free for
// the next tile after the barrier that ends the convert pass.
__device__ __forceinline__ void lds_barrier();  // (defined below)
// One 1 KB LDS-DMA piece: lane l's 16 bytes at `g` land at LDS byte address `lds_off` + 16 l (`lds_off` wave-uniform, goes through M0).
// Issued through INLINE ASM on purpose: with `__builtin_amdgcn_global_load_lds` hipcc treats every later LDS read as a possible reader of
// the piece and puts `s_waitcnt vmcnt(0)` in front of it (checked in the ISA of the first version of k_post_dma: the wait sat between the
// prefetch of tile i + 1 and the conv of tile i, so the "prefetch" never overlapped anything and the kernel's phases simply added up).
// The compiler does not see these operations: THE CALLER waits for them with explicit `s_waitcnt vmcnt(N)` (loads retire in order, so hidden
// older operations only make the compiler's own counted waits stricter) and must drain them before the kernel ends.
__device__ __forceinline__ void glds16(const void* g, unsigned lds_off) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g), "s"(lds_off)
                 : "memory");
}
constexpr int POSTD_TT = 128;  // output rows per tile = threads per block (two waves); three blocks per CU
// Ablations (round 6, B = 1, option POST_DBG): whole kernel 36 us; DMA alone 23 us (113 MB at 4.9 TB/s: three blocks per CU x one 26 KB
// tile in flight each against ~4 us of loaded latency); everything but the DMA ~15 us (LDS-bound: 112 ds_read_b128 per output).  A variant with TWO
// tiles in flight (counted vmcnt, two 75 KB blocks per CU) was built, bit-identical, and measured 46 us: four waves per CU hide even less of the LDS phase.  Removed.
template <int C>
static __global__ void __launch_bounds__(POSTD_TT) k_post_dma(const _Float16* __restrict__ xa, const _Float16* __restrict__ xb,
                                                             const _Float16* __restrict__ xc, const float* __restrict__ Wp /*[7][C]*/,
                                                             float* __restrict__ out, int Lmax, const int* __restrict__ lens, int lmul,
                                                             int dbg /* timing ablations only (wrong results): 1 no DMA, 2 no convert / conv */) {
    constexpr int TT = POSTD_TT, NW = TT / 64, S = C + 4, C8 = C / 8, ROWB = C * 2;
    constexpr int TILEB = (TT + 6) * ROWB;                 // bytes of one stream's rows of a tile
    constexpr int NP = (TILEB + 1023) / 1024, RAWB = NP * 1024;  // 1 KB DMA pieces per stream
    constexpr int PPW0 = (3 * NP + NW - 1) / NW;           // pieces per tile issued by a wave (at most)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* w = (float*)smem_raw;                           // [7][C]
    float* tile = w + 7 * C;                               // [TT + 6][C + 4] fp32
    char* raw = (char*)(tile + (TT + 6) * S);              // [3][RAWB] fp16 rows as they lie in HBM
    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 7 * C; i += TT) w[i] = Wp[i];
    const int L = item_rows(lens, b, lmul, Lmax);
    const size_t boff = (size_t)b * Lmax * C;
    const int nvalid = (L + TT - 1) / TT, nall = (Lmax + TT - 1) / TT;
    const char* src[3] = {(const char*)(xa + boff), (const char*)(xb + boff), (const char*)(xc + boff)};
    const unsigned raw_off = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) char*)raw);

    auto issue = [&](int ti) {  // the (TT + 6) rows [t0 - 3, t0 + TT + 3) of the three streams; rows outside [0, L) are clamped here, zeroed in convert
        if (dbg & 1) return;
        const int t0 = ti * TT;
#pragma unroll
        for (int p0 = 0; p0 < PPW0; ++p0) {
            const int p = p0 * NW + wave;  // wave-uniform
            if (p < 3 * NP) {
                const int st = p / NP, pp = p - st * NP;
                const int o = pp * 1024 + lane * 16;
                const int r = o / ROWB, col = o - r * ROWB;
                const int tc = min(max(t0 - 3 + r, 0), L - 1);
                glds16(src[st] + (size_t)tc * ROWB + col, raw_off + (unsigned)(st * RAWB + pp * 1024));
            }
        }
    };

    int ti = blockIdx.x;
    if (ti < nvalid) issue(ti);
    for (; ti < nvalid; ti += gridDim.x) {
        const int t0 = ti * TT;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the pieces THIS wave issued have landed ...
        lds_barrier();                                   // ... and so have everybody's; the previous tile's conv is done with `tile`
        if (dbg & 2) {
            lds_barrier();
            if (ti + (int)gridDim.x < nvalid) issue(ti + gridDim.x);
            if (tid == 0) out[(size_t)b * Lmax + t0] = *(const float*)raw;
            continue;
        }
        for (int idx = tid; idx < (TT + 6) * C8; idx += TT) {
            const int r = idx / C8, c8 = idx - r * C8;
            const int t = t0 - 3 + r;
            float v[8], u[8];
            unpack8_h(*(const uint4*)(raw + r * ROWB + c8 * 16), v);
            unpack8_h(*(const uint4*)(raw + RAWB + r * ROWB + c8 * 16), u);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += u[e];
            unpack8_h(*(const uint4*)(raw + 2 * RAWB + r * ROWB + c8 * 16), u);
            const bool in = t >= 0 && t < L;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = in ? lrelu(div3_exact(v[e] + u[e]), 0.01f) : 0.f;
            *(float4*)(tile + r * S + c8 * 8) = make_float4(v[0], v[1], v[2], v[3]);
            *(float4*)(tile + r * S + c8 * 8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
        lds_barrier();  // the tile is complete and `raw` is free
        if (ti + (int)gridDim.x < nvalid) issue(ti + gridDim.x);  // flies during the conv below
        const int t = t0 + tid;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const float4* row = (const float4*)(tile + (tid + j) * S);
            const float4* wj = (const float4*)(w + j * C);
#pragma unroll
            for (int c = 0; c < C / 4; ++c) {  // same accumulation order as k_post / the scalar loop
                const float4 xv = row[c], wv = wj[c];
                acc = fmaf(xv.x, wv.x, acc);
                acc = fmaf(xv.y, wv.y, acc);
                acc = fmaf(xv.z, wv.z, acc);
                acc = fmaf(xv.w, wv.w, acc);
            }
        }
        if (t < L) out[(size_t)b * Lmax + t] = tanhf(acc);
        else if (t < Lmax) out[(size_t)b * Lmax + t] = 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (nothing is in flight here by construction; the compiler cannot know)
    for (ti = blockIdx.x; ti < nall; ti += gridDim.x)  // (ragged batch) tiles behind the item's end
        if (ti >= nvalid && ti * TT + tid < Lmax) out[(size_t)b * Lmax + ti * TT + tid] = 0.f;
}
template <int C>
static constexpr size_t post_dma_smem() {
    return (size_t)(7 * C + (POSTD_TT + 6) * (C + 4)) * 4 + (size_t)3 * (((POSTD_TT + 6) * C * 2 + 1023) / 1024) * 1024;
}

// ... of fp16 streams (Y_F16)
static __global__ void __launch_bounds__(256) k_sum3h(const _Float16* __restrict__ a, const _Float16* __restrict__ b,
                                               const _Float16* __restrict__ c, float* __restrict__ y, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = ((float)a[i] + (b ? (float)b[i] : 0.f)) + (c ? (float)c[i] : 0.f);
}
static __global__ void __launch_bounds__(256) k_h2f(const _Float16* __restrict__ a, float* __restrict__ y, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = (float)a[i];
}
// y = (a + b) + c   (debug tap of the stage sum only)
static __global__ void __launch_bounds__(256) k_sum3(const float* __restrict__ a, const float* __restrict__ b,
                                              const float* __restrict__ c, float* __restrict__ y, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = (a[i] + (b ? b[i] : 0.f)) + (c ? c[i] : 0.f);
}

// channels-last fp32 [B][L][C] -> channel-first [B][C][L] (debug taps only)
static __global__ void __launch_bounds__(256) k_cl_to_cf(const float* __restrict__ in, float* __restrict__ out, int L, int C) {
    const int b = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)L * C) return;
    const int c = (int)(i / L);
    const int t = (int)(i - (size_t)c * L);
    out[(size_t)b * L * C + i] = in[((size_t)b * L + t) * C + c];
}

// ------------------------------------------------------------------------------------------------
// MFMA convolution (RVCMI_OPERAND_BF16 / _F16)
// ------------------------------------------------------------------------------------------------
//
// GEMM view of a k-tap conv: D[co][t] = sum_{tap,ci} W[co][tap,ci] * X[t + off(tap)][ci]
//   M = output channels, N = time, K = taps*C_in, instruction v_mfma_f32_32x32x16_{bf16,f16}.
//   A (weights)      lane l holds W[co = l&31][k = 8*(l>>5) .. +8]  -> pre-packed in that exact
//                    order on the host, streamed global->VGPR (each wave owns its own co slice,
//                    so LDS staging would buy nothing), double-buffered one k-group ahead.
//   B (activations)  lane l holds X[t = l&31][ci = 8*(l>>5) .. +8]  -> ONE ds_read_b128 from the
//                    channels-last LDS tile; rows padded by 16 B so the 16 lanes of a read group
//                    hit 16 distinct 16-byte bank slots.
//   D                lane l holds t = l&31, co = (r&3) + 8*(r>>2) + 4*(l>>5)  -> 4 consecutive
//                    channels per register quad = one float4 (or 8-byte OpT) store per quad.
// The activation tile is staged ONCE per block and reused by every tap and every output channel,
// so the K loop has no barrier at all.

template <typename OpT>
struct Op;
template <>
struct Op<__bf16> {
    using frag = bf16x8;
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <>
struct Op<_Float16> {
    using frag = f16x8;
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

constexpr int KGROUP = 4;  // k-steps (of 16) per software-pipeline group (also the weight-packing pad unit)

// float -> operand, round-to-nearest-even; fp16 saturates instead of overflowing to inf.
template <typename OpT>
__device__ __forceinline__ OpT to_op(float v) {
    return (OpT)v;
}
template <>
__device__ __forceinline__ _Float16 to_op<_Float16>(float v) {
    return (_Float16)__builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);  // one v_med3_f32 instead of max + min
}

template <int CIN>
struct Tile {
    static constexpr int CC = CIN / 16;              // k-steps per tap
    static constexpr int STRIDE = CIN * 2 + 16;      // bytes per LDS row (16-B pad: conflict-free b128 reads)
    static constexpr int TAPS_PER_GROUP = (CC >= KGROUP) ? 1 : KGROUP / CC;
    static_assert(CIN % 16 == 0, "C_in must be a multiple of 16");
    static_assert((CC >= KGROUP) ? (CC % KGROUP == 0) : (KGROUP % CC == 0), "k-group must tile a tap");
};

// Stage `rows` input rows starting at global row g0 into the LDS tile (zero outside [0, Lin)).
template <typename OpT, int CIN>
__device__ __forceinline__ void stage_tile(char* smem, const ConvArgs& a, int b, int g0, int rows) {
    using frag = typename Op<OpT>::frag;
    constexpr int STRIDE = Tile<CIN>::STRIDE;
    constexpr int C8 = CIN / 8;
    const int tid = threadIdx.x;
    const int Linb = item_rows(a.lens, b, a.lmul_in, a.Lin);  // ragged batch: this item's input rows
    if (a.in_mode == IN_F32_CF) {
        // channel-first input (the caller's [B][C][T], nsf.py:164): consecutive lanes walk TIME so that every load
        // instruction is one contiguous run per channel; 8 channels x SBC chunks in flight per thread (the row-major
        // mapping below made this 8 uncoalesced serial loads per chunk: 49 us for conv_pre at T = 1198)
        constexpr int SBC = 4;
        const float* ip = (const float*)a.in + (size_t)b * a.in_bstride;
        const int total = rows * C8;
        for (int base = tid; base < total; base += SBC * 256) {
            float f[SBC][8];
#pragma unroll
            for (int u = 0; u < SBC; ++u) {
                const int idx = min(base + u * 256, total - 1);
                const int c8 = idx / rows, r = idx - c8 * rows;
                const int grc = min(max(g0 + r, 0), Linb - 1);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[u][e] = ip[(size_t)(c8 * 8 + e) * a.cf_stride + grc];
            }
#pragma unroll
            for (int u = 0; u < SBC; ++u) {
                const int idx = base + u * 256;
                if (idx < total) {
                    const int c8 = idx / rows, r = idx - c8 * rows;
                    const int gr = g0 + r;
                    const bool ok = gr >= 0 && gr < Linb;
                    frag v;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = ok ? to_op<OpT>(f[u][e]) : (OpT)0.f;
                    *(frag*)(smem + (size_t)r * STRIDE + c8 * 16) = v;
                }
            }
        }
        return;
    }
    if (a.in_mode == IN_OP_RAW || (a.in_mode == IN_F32_ACT && !a.in_b && !a.in_c && a.div_in == 1.f)) {
        // plain row-major inputs (the conv-by-conv resblock path): SB chunks in flight per thread, clamped addresses so that every
        // load is unconditional (the one-chunk-at-a-time loop below is a load -> convert -> store round trip per chunk: 22 serial
        // L2 latencies for a 176-row tile at C = 256, the whole duration of a small launch)
        constexpr int SB = 8;
        const int total = rows * C8;
        const bool raw = a.in_mode == IN_OP_RAW;
        for (int base = tid; base < total; base += SB * 256) {
            frag vr[SB];
            float4 lo[SB], hi[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int idx = min(base + u * 256, total - 1);
                const int r = idx / C8, c8 = idx - r * C8;
                const int grc = min(max(g0 + r, 0), Linb - 1);
                if (raw) {
                    vr[u] = *(const frag*)((const OpT*)a.in + (size_t)b * a.in_bstride + (size_t)grc * CIN + c8 * 8);
                } else {
                    const float4* p = (const float4*)((const float*)a.in + (size_t)b * a.in_bstride + (size_t)grc * CIN + c8 * 8);
                    lo[u] = p[0];
                    hi[u] = p[1];
                }
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int idx = base + u * 256;
                if (idx < total) {
                    const int r = idx / C8, c8 = idx - r * C8;
                    const int gr = g0 + r;
                    const bool ok = gr >= 0 && gr < Linb;
                    frag v;
                    if (raw) {
                        v = vr[u];
                    } else {
                        const float f[8] = {lo[u].x, lo[u].y, lo[u].z, lo[u].w, hi[u].x, hi[u].y, hi[u].z, hi[u].w};
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = to_op<OpT>(lrelu(f[e], a.slope_in));
                    }
                    if (!ok) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (OpT)0.f;
                    }
                    *(frag*)(smem + (size_t)r * STRIDE + c8 * 16) = v;
                }
            }
        }
        return;
    }
    for (int idx = tid; idx < rows * C8; idx += 256) {
        const int r = idx / C8;
        const int c8 = idx - r * C8;
        const int gr = g0 + r;
        frag v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (OpT)0.f;
        if (a.in_mode == IN_HAR) {
            // noise_convs[i] (nsf.py:103-115) is Conv1d(1, C, k=2s, stride s, pad s/2): over frames of s samples it is
            // a plain 2-tap conv with C_in = s, so it runs on the MFMA path.  s % 8 == 0 keeps every 8-sample
            // chunk entirely inside or outside the signal and 16-byte aligned.
            // (hs % 8 == 0, hpad = hs/2 and Lin % 4 == 0 make every 4-sample half either fully inside or fully
            // outside the signal; an 8-sample chunk may straddle its start/end, so validity is per half.)
            const long base = (long)gr * a.hs - a.hpad + c8 * 8;
            if (c8 * 8 < a.hs) {
                const float* hp = (const float*)a.in + (size_t)b * a.in_bstride;
                const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 lo = (base >= 0 && base + 4 <= Linb) ? *(const float4*)(hp + base) : z4;
                const float4 hi = (base + 4 >= 0 && base + 8 <= Linb) ? *(const float4*)(hp + base + 4) : z4;
                const float f[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = to_op<OpT>(f[e]);
            }
        } else if (gr >= 0 && gr < Linb) {
            if (a.in_mode == IN_OP_RAW) {
                v = *(const frag*)((const OpT*)a.in + (size_t)b * a.in_bstride + (size_t)gr * CIN + c8 * 8);
            } else if (a.in_mode == IN_F32_ACT) {
                const float4* p = (const float4*)((const float*)a.in + (size_t)b * a.in_bstride + (size_t)gr * CIN + c8 * 8);
                float4 lo = p[0], hi = p[1];
                float f[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                if (a.in_b) {
                    const float4* pb = (const float4*)(a.in_b + (size_t)b * a.in_bstride + (size_t)gr * CIN + c8 * 8);
                    float4 l2 = pb[0], h2 = pb[1];
                    f[0] += l2.x; f[1] += l2.y; f[2] += l2.z; f[3] += l2.w; f[4] += h2.x; f[5] += h2.y; f[6] += h2.z; f[7] += h2.w;
                }
                if (a.in_c) {
                    const float4* pc = (const float4*)(a.in_c + (size_t)b * a.in_bstride + (size_t)gr * CIN + c8 * 8);
                    float4 l2 = pc[0], h2 = pc[1];
                    f[0] += l2.x; f[1] += l2.y; f[2] += l2.z; f[3] += l2.w; f[4] += h2.x; f[5] += h2.y; f[6] += h2.z; f[7] += h2.w;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float x = f[e];
                    if (a.div_in == 3.f) x = div3_exact(x);
                    else if (a.div_in != 1.f) x = x / a.div_in;
                    v[e] = to_op<OpT>(lrelu(x, a.slope_in));
                }
            } else {  // IN_F32_CF
                const float* p = (const float*)a.in + (size_t)b * a.in_bstride + (size_t)(c8 * 8) * a.cf_stride + gr;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = to_op<OpT>(p[(size_t)e * a.cf_stride]);
            }
        }
        *(frag*)(smem + (size_t)r * STRIDE + c8 * 16) = v;
    }
}

// The K loop.  acc[mi][jt] += sum over (padded) taps and channel chunks.
//   lds_lane : smem + (wave_t0 + (lane&31)) * STRIDE + (lane>>5)*16
//   wlane    : packed weights of this wave's first co tile + lane*8
// The K loop is split in two so that a kernel can put a barrier / epilogue BETWEEN "weights requested" and
// "weights used":
//   conv_prefetch  issues the global loads of the first NB-1 weight groups (KGROUP k-steps each) into registers;
//   conv_run       multiplies group g while group g+NB-1 is in flight.  (An explicit one-k-step-ahead double buffer
//                  of the B fragments was tried and measured SLOWER -- 11.2k -> 13.3k cycles per k=7 conv at C=64 --
//                  than leaving the ds_read placement to the compiler.)
template <typename OpT, int MI, int KGROUP, int NB>
__device__ __forceinline__ void conv_load_group(typename Op<OpT>::frag (&Ab)[KGROUP][MI], const OpT* wlane, long ct_stride, int grp) {
    using frag = typename Op<OpT>::frag;
#pragma unroll
    for (int g = 0; g < KGROUP; ++g)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
            Ab[g][mi] = *(const frag*)(wlane + (size_t)mi * ct_stride + (size_t)(grp * KGROUP + g) * 512);
}

template <typename OpT, int CIN, int MI, int KGROUP, int NB>
__device__ __forceinline__ void conv_prefetch(typename Op<OpT>::frag (&A)[NB][KGROUP][MI], const OpT* wlane, long ct_stride,
                                              int ntaps_p) {
    const int NG = ntaps_p * Tile<CIN>::CC / KGROUP;
#pragma unroll
    for (int i = 0; i < NB - 1; ++i)
        if (i < NG) conv_load_group<OpT, MI, KGROUP, NB>(A[i], wlane, ct_stride, i);
}

// One wave per SIMD issues in order, so every non-MFMA instruction must sit in the 32-cycle shadow of an MFMA:
//   * B addresses are a running per-group base + compile-time immediates (no per-k-step scalar math),
//   * every load is unconditional (clamped indices instead of branches, so a whole k-group is one scheduling region),
//   * the ds_reads of the NEXT k-step and the global weight loads of group g+NB-1 are spread one per MFMA and the
//     order is pinned with sched_group_barrier (MFMA, DS_READ, MFMA, DS_READ, ..., MFMA, VMEM, ...).
// Measured in tools/ubench/kloop.hip (C=64, k=7): 58.5 -> see DESIGN.md cycles per MFMA.
template <typename OpT, int CIN, int MI, int NJ, int KGROUP, int NB>
__device__ __forceinline__ void conv_run(f32x16 (&acc)[MI][NJ], typename Op<OpT>::frag (&A)[NB][KGROUP][MI],
                                         const char* lds_lane, const OpT* wlane, long ct_stride, int ntaps_p, int roff,
                                         int dstep, int dbg = 0) {
    using frag = typename Op<OpT>::frag;
    using TL = Tile<CIN>;
    constexpr int CC = TL::CC;
    constexpr int STRIDE = TL::STRIDE;
    static_assert((CC >= KGROUP) ? (CC % KGROUP == 0) : (KGROUP % CC == 0), "k-group must tile a tap");
    static_assert(NB >= 2 && NB <= 4, "2..4 weight buffers");
    static_assert(KGROUP % 2 == 0, "the B ping-pong needs an even k-group");
    constexpr int GPT = (CC >= KGROUP) ? CC / KGROUP : 1;  // k-groups per tap
    constexpr int TPG = (CC >= KGROUP) ? 1 : KGROUP / CC;  // taps per k-group
    const int NG = ntaps_p * CC / KGROUP;
    const int dS = dstep * STRIDE;
    (void)dbg;

    // byte offset of k-step k of a group relative to the group base
    auto koff = [&](int k) { return (CC >= KGROUP) ? k * 32 : (k / CC) * dS + (k % CC) * 32; };
    auto readB = [&](frag(&B)[NJ], const char* base, int k) {
        const char* bp = base + koff(k);
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) B[jt] = *(const frag*)(bp + jt * 32 * STRIDE);
    };

    const char* gb = lds_lane + roff * STRIDE;  // B base of the current group
    int gi = 0;                                 // group index inside the current tap (CC >= KGROUP only)
    const OpT* an = wlane + (size_t)min(NB - 1, NG - 1) * KGROUP * 512;  // weights of the group to request next
    int gn = min(NB - 1, NG - 1);

    frag Bf[2][NJ];
#ifndef RVCMI_KLOOP_V1
    // first k-step's B fragments, in ASCENDING tile order like every later k-step (the waitcnt pass merges the loop entry
    // with the back edge: a different order here makes it wait for lgkmcnt(0) at the head of every group)
#pragma unroll
    for (int jt = 0; jt < NJ; ++jt) {
        Bf[0][jt] = *(const frag*)(gb + koff(0) + jt * 32 * STRIDE);
        __builtin_amdgcn_sched_barrier(0);
    }
#else
    readB(Bf[0], gb, 0);
#endif
    for (int grp = 0; grp < NG; grp += NB) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int g = grp + u;
            if (g < NG) {
                const char* gbn;
                if constexpr (CC >= KGROUP) gbn = (gi + 1 == GPT) ? gb - (GPT - 1) * KGROUP * 32 + dS : gb + KGROUP * 32;
                else gbn = gb + TPG * dS;
#ifndef RVCMI_KLOOP_V1
                // Slot i of a k-step = MFMA i, then (i < NJ) the ds_read of the NEXT k-step's B fragment of time tile i, then
                // (last MI slots) one weight load of group g+NB-1.  The order is pinned with sched_barrier: left to the
                // scheduler, the reads that cross a group boundary came out in DESCENDING tile order, so the first MFMA of
                // every group waited for the LAST read issued (s_waitcnt lgkmcnt(0): a full LDS round trip, ~120 cycles per
                // 24 MFMAs = the 37 instead of 32 cycles per MFMA measured in round 2).
#pragma unroll
                for (int k = 0; k < KGROUP; ++k) {
                    const char* nbp = (k + 1 < KGROUP) ? gb + koff(k + 1) : gbn + koff(0);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int jt = 0; jt < NJ; ++jt) {
                            constexpr int NS = MI * NJ;
                            const int i = mi * NJ + jt;
                            acc[mi][jt] = Op<OpT>::mfma(A[u][k][mi], Bf[k & 1][jt], acc[mi][jt]);
#ifdef RVCMI_KLOOP_ABLATE  // tools/ubench/kloop.hip only: 1 = no weight loads, 2 = no B reads inside the loop
                            if (!(RVCMI_KLOOP_ABLATE & 2))
#endif
                            if (i < NJ) Bf[(k + 1) & 1][i] = *(const frag*)(nbp + i * 32 * STRIDE);
#ifdef RVCMI_KLOOP_ABLATE
                            if (!(RVCMI_KLOOP_ABLATE & 1))
#endif
                            if (i >= NS - MI)
                                A[(u + NB - 1) % NB][k][i - (NS - MI)] = *(const frag*)(an + (size_t)(i - (NS - MI)) * ct_stride + k * 512);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                }
#else
#pragma unroll
                for (int k = 0; k < KGROUP; ++k) {
                    // next k-step's B fragments (first k-step of the next group at the group boundary; past the very
                    // last group this reads a few rows beyond the tile -- in-bounds of LDS or zero, never used)
                    if (k + 1 < KGROUP) readB(Bf[(k + 1) & 1], gb, k + 1);
                    else readB(Bf[0], gbn, 0);
                    // MI of the KGROUP*MI weight fragments of group g+NB-1 (clamped: re-requests the last group at the tail)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        A[(u + NB - 1) % NB][k][mi] = *(const frag*)(an + (size_t)mi * ct_stride + k * 512);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int jt = 0; jt < NJ; ++jt) acc[mi][jt] = Op<OpT>::mfma(A[u][k][mi], Bf[k & 1][jt], acc[mi][jt]);
#pragma unroll
                    for (int i = 0; i < MI * NJ; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        // 1 MFMA
                        if (i < NJ) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);            // 1 DS read
                        else if (i < NJ + MI) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 VMEM read
                    }
                }
#endif
                gb = gbn;
                if constexpr (CC >= KGROUP) gi = (gi + 1 == GPT) ? 0 : gi + 1;
                if (gn < NG - 1) {
                    ++gn;
                    an += KGROUP * 512;
                }
            }
        }
    }
}

// ---- the lean K loop (round 3) ---------------------------------------------------------------------------------------
// conv_run's ring of NB groups of KGROUP k-steps pays a scalar/vector bookkeeping tail per group (one MFMA gap of ~9 and a tail
// of ~14 instructions per 24 MFMAs): 37.0 cycles per MFMA for a lone wave against 34.0 for the same traffic without it
// (tools/ubench/kloop2.hip).  kconv keeps that structure out of the loop:
//   * weights are RAW BUFFER loads: SGPR resource + SGPR running offset + one lane offset VGPR + immediate -- no 64-bit per-lane
//     address arithmetic at all;
//   * the weight ring is 8 k-steps deep and refilled IN PLACE: A[kk] is re-requested for the k-step 8 later right after its last
//     MFMA (twice the look-ahead of NB = 2 x KGROUP = 4 in the same registers), so the loop has no group structure;
//   * B fragments come from the channels-last LDS tile through one base VGPR per 8 k-steps and immediate offsets.
// For C_in a multiple of 128 (8 k-steps = a whole or half tap).  lds_row: LDS byte address of this lane's row at tap 0 (+ 16 for
// the upper half-wave), rsrc / loff: this wave's first output-channel tile and lane * 16, ctb: bytes per output-channel tile.
using lds_cptr_t = const __attribute__((address_space(3))) char*;
__device__ __forceinline__ unsigned lds_address(const void* p) { return (unsigned)(size_t)(lds_cptr_t)p; }
using u32x4_t = __attribute__((ext_vector_type(4))) unsigned;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t weight_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000);
}
template <typename OpT>
__device__ __forceinline__ typename Op<OpT>::frag weight_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(typename Op<OpT>::frag, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
template <typename OpT, int MI>
__device__ __forceinline__ void kconv_prefetch(typename Op<OpT>::frag (&A)[8][MI], __amdgpu_buffer_rsrc_t r, unsigned loff, unsigned ctb) {
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) A[kk][mi] = weight_load<OpT>(r, loff + kk * 1024, mi * ctb);
}
template <typename OpT, int CIN, int MI, int NJ, int STRIDE>
__device__ __forceinline__ void kconv(f32x16 (&acc)[MI][NJ], typename Op<OpT>::frag (&A)[8][MI], unsigned lds_row, __amdgpu_buffer_rsrc_t r,
                                      unsigned loff, unsigned ctb, int ntaps, int dstep) {
    using frag = typename Op<OpT>::frag;
    constexpr int CC = CIN / 16, GPT = CC / 8;  // k-steps per tap, groups of 8 k-steps per tap
    static_assert(CIN % 128 == 0, "kconv: C_in must be a multiple of 128");
    constexpr unsigned TS = 32u * STRIDE;
    auto ld = [](unsigned a) { return *(const __attribute__((address_space(3))) frag*)(size_t)a; };
    frag Bf[2][NJ];
#pragma unroll
    for (int jt = 0; jt < NJ; ++jt) {
        Bf[0][jt] = ld(lds_row + jt * TS);
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned dS = (unsigned)(dstep * STRIDE);
    const int ngroups = ntaps * GPT;
    const unsigned slast = (unsigned)(ngroups - 1) * 8192u;
    unsigned soff = ngroups > 1 ? 8192u : 0u;  // byte offset of the group being requested (clamped: the last one re-requests itself)
    int gi = 0;                                // group inside the tap
    for (int g = 0; g < ngroups; ++g) {
        const unsigned nrow = (gi + 1 == GPT) ? lds_row - (unsigned)((GPT - 1) * 256) + dS : lds_row + 256u;  // next group's first k-step
        // (opaque copies: loop strength reduction otherwise rebases every read on the POST-increment pointer with negative
        //  offsets, which do not fit the ds_read offset field -- one v_add per MFMA)
        unsigned b0 = lds_row, b1 = nrow;
        asm volatile("" : "+v"(b0), "+v"(b1));
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt) {
                    constexpr int NS = MI * NJ;
                    const int i = mi * NJ + jt;
                    acc[mi][jt] = Op<OpT>::mfma(A[kk][mi], Bf[kk & 1][jt], acc[mi][jt]);
                    if (i < NJ) Bf[(kk + 1) & 1][i] = ld((kk < 7 ? b0 + (unsigned)((kk + 1) * 32) : b1) + i * TS);
                    if (i >= NS - MI) A[kk][i - (NS - MI)] = weight_load<OpT>(r, loff + kk * 1024, soff + (unsigned)(i - (NS - MI)) * ctb);
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        lds_row = nrow;
        gi = (gi + 1 == GPT) ? 0 : gi + 1;
        soff = soff < slast ? soff + 8192u : soff;
    }
}

template <typename OpT, int CIN, int MI, int NJ, int KGROUP = ::rvcmi::KGROUP, int NB = 2>
__device__ __forceinline__ void conv_core(f32x16 (&acc)[MI][NJ], const char* lds_lane, const OpT* wlane,
                                          long ct_stride, int ntaps_p, int roff, int dstep) {
    typename Op<OpT>::frag A[NB][KGROUP][MI];
    conv_prefetch<OpT, CIN, MI, KGROUP, NB>(A, wlane, ct_stride, ntaps_p);
    conv_run<OpT, CIN, MI, NJ, KGROUP, NB>(acc, A, lds_lane, wlane, ct_stride, ntaps_p, roff, dstep);
}

// Generic single-conv kernel: stage tile -> conv_core -> epilogue.  Grid: x = time tile,
// y = (co block) * nphase + phase, z = batch.  Block = 4 waves laid out WCO (co) x WT (time).
template <typename OpT, int CIN, int MI, int NJ, int WCO, int NB = 2, bool PFW = false>
__device__ __forceinline__ void conv_mfma_body(const ConvArgs& a, int b, char* smem) {
    using TL = Tile<CIN>;
    constexpr int WT = 4 / WCO;
    constexpr int TT = WT * NJ * 32;
    const int ph = a.nphase > 1 ? (int)(blockIdx.y % a.nphase) : 0;
    const int cob = a.nphase > 1 ? (int)(blockIdx.y / a.nphase) : (int)blockIdx.y;
    const int q0 = blockIdx.x * TT;
    const int in_off = a.nphase > 1 ? a.ph_in_off[ph] : a.in_off;
    const OpT* wbase = (const OpT*)a.w + (a.nphase > 1 ? a.ph_w_off[ph] : 0);

    unsigned pf_acc = 0;
    if constexpr (PFW) {
        // Tiny launches (a realtime chunk) use every weight byte for a handful of row tiles only, so the weights are never
        // L2-resident: the K loop's 12 KB of loads in flight per wave then runs at the latency of the far memory side (measured:
        // 24 us per launch for a 6 us K loop).  Touch the block's whole weight slice up front -- every thread 16 bytes per 4 KB
        // round, all rounds in flight next to the tile staging -- so that the K loop finds it in the XCD's L2.
        const char* wb = (const char*)(wbase + (size_t)cob * WCO * MI * a.w_ct_stride);
        const size_t bytes = (size_t)WCO * MI * a.w_ct_stride * sizeof(OpT);
        const size_t rounds = bytes / 4096;
#pragma unroll 8
        for (size_t r = 0; r < rounds; ++r) {
            const uint4 v = *(const uint4*)(wb + r * 4096 + threadIdx.x * 16);
            pf_acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    stage_tile<OpT, CIN>(smem, a, b, q0 + in_off - a.roff, a.tile_rows);
    if constexpr (PFW) {
        if (pf_acc == 0x9e3779b9u && a.Lq < 0) *(unsigned*)smem = pf_acc;  // never true: keeps the prefetch loads alive
    }
    __syncthreads();

    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int wco = wave % WCO;
    const int wt = wave / WCO;
    const int ct0 = (cob * WCO + wco) * MI;  // first 32-channel output tile of this wave
    const int tw0 = wt * NJ * 32;            // first time row of this wave inside the block tile

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][jt][r] = 0.f;

    const char* lds_lane = smem + (size_t)(tw0 + (lane & 31)) * TL::STRIDE + (lane >> 5) * 16;
    const OpT* wlane = wbase + (size_t)ct0 * a.w_ct_stride + lane * 8;
    conv_core<OpT, CIN, MI, NJ, KGROUP, NB>(acc, lds_lane, wlane, a.w_ct_stride, a.ntaps, a.roff, a.dstep);

    // ---- epilogue -------------------------------------------------------------------------
    const int out_add = a.nphase > 1 ? ph : a.out_add;
    // bias (+ the per-utterance cond vector) of this lane's channels: fetched ONCE, as one batch, before the store loops
    // (a conditional load inside them becomes a branch with its own wait: 64 serialised L2 round trips = ~30 us on conv_pre)
    f32x4 bvec[MI][4];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int g = 0; g < 4; ++g) bvec[mi][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int g = 0; g < 4; ++g) bvec[mi][g] = *(const f32x4*)(a.bias + min((ct0 + mi) * 32 + 4 * (lane >> 5) + 8 * g, a.cout - 4));
    }
    if (a.cb) {  // (also with OUT_ACT: conv_pre writes lrelu(conv + bias + cond) as operands, round 6)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                bvec[mi][g] += *(const f32x4*)(a.cb + (size_t)b * a.cout + min((ct0 + mi) * 32 + 4 * (lane >> 5) + 8 * g, a.cout - 4));
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int cobase = (ct0 + mi) * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) {
            const int q = q0 + tw0 + jt * 32 + (lane & 31);
            if (q >= item_rows(a.lens, b, a.lmul_q, a.Lq)) continue;
            const size_t orow = (size_t)q * a.out_mul + out_add;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = cobase + 8 * g;
                if (co >= a.cout) continue;  // cout is a multiple of 4 whenever it is < the padded tile
                f32x4 v = {acc[mi][jt][4 * g + 0], acc[mi][jt][4 * g + 1], acc[mi][jt][4 * g + 2], acc[mi][jt][4 * g + 3]};
                v += bvec[mi][g];
                if (a.out_mode == OUT_ACT) {
                    using o4 = __attribute__((ext_vector_type(4))) OpT;
                    o4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = to_op<OpT>(lrelu(v[e], a.slope_out));
                    *(o4*)((OpT*)a.out + (size_t)b * a.out_bstride + orow * a.out_C + co) = o;
                } else {
                    if (a.res) v += *(const f32x4*)(a.res + (size_t)b * a.res_bstride + orow * a.out_C + co);
                    f32x4* o = (f32x4*)((float*)a.out + (size_t)b * a.out_bstride + orow * a.out_C + co);
                    if (a.accumulate) v += *o;
                    *o = v;
                }
            }
        }
    }
}

template <typename OpT, int CIN, int MI, int NJ, int WCO>
static __global__ void __launch_bounds__(256) k_conv_mfma(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    conv_mfma_body<OpT, CIN, MI, NJ, WCO>(a, (int)blockIdx.z, smem);
}

// The same conv for up to three independent jobs in ONE launch (grid.z = batch x jobs).  Used where a fused resblock launch would
// be a handful of blocks (the realtime chunk: 310 rows at C = 256 = 9 pair tiles, each streaming 2.9 MB of weights through one
// CU): conv1 / conv2 of all the stage's resblocks as separate, output-channel-split launches of ~70 blocks.
struct ConvJobs {
    ConvArgs job[3];
    int njobs;
};
// K-SPLIT form of the same conv (round 5).  In the conv-by-conv path of a realtime chunk (310 rows at C = 256, 3100 at C = 128) a wave's K
// loop is one MFMA per k-step behind a weight fragment from L2: 176 k-steps at k = 11 run at the LATENCY of that stream whatever the ring
// depth (19-22 us per launch for ~6 us of work), and three row tiles x eight channel tiles x three jobs are 72 blocks on 256 CUs.  Here the
// block's four waves split the TAPS of ONE 32-row x 32 MI-channel output tile (44 k-steps each at k = 11), the partial sums meet in LDS and
// wave 0 adds them in a fixed order ((w0 + w1) + w2) + w3 and runs the epilogue: a quarter of the dependent weight round trips per wave, no
// weight byte fetched twice, and row tiles x channel tiles x jobs = 240 blocks.  Grid: x = 32-row tile, y = channel tile, z = batch x jobs.
// NJ = row tiles per block.  Measured (realtime chunk, ABAB): C = 256, NJ = 1: six launches 134 -> 107 us; C = 128 (3100 rows): NJ = 1 114 -> 155 us
// (1164 blocks that each stage 82 rows for 32 outputs), NJ = 3 (396 blocks) 113 -> 107 us.
template <typename OpT, int CIN, int MI, int NJ>
__device__ __forceinline__ void conv_ks_body(const ConvArgs& a, int b, char* smem) {
    using TL = Tile<CIN>;
    constexpr int CC = TL::CC;
    constexpr int TT = 32 * NJ;  // rows per block
    static_assert(CC >= KGROUP, "K split by taps: a tap must be whole k-groups");
    const int cob = (int)blockIdx.y;
    const int q0 = blockIdx.x * TT;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ct0 = cob * MI;
    const OpT* wbase = (const OpT*)a.w;
    // this wave's taps [t0, t1) (k = 3: one tap each for three waves, the fourth adds zeros)
    const int per = (a.ntaps + 3) / 4;
    const int t0 = min(wave * per, a.ntaps), t1 = min(t0 + per, a.ntaps);
    unsigned pf_acc = 0;
    {   // touch this wave's weight slice up front (see conv_mfma_body PFW): the K loop then finds it in the XCD's L2
        const char* wb = (const char*)(wbase + (size_t)ct0 * a.w_ct_stride + (size_t)t0 * CC * 512);
        for (int r = 0; r < (t1 - t0) * CC; ++r)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const uint4 v = *(const uint4*)(wb + ((size_t)mi * a.w_ct_stride + (size_t)r * 512) * sizeof(OpT) + lane * 16);
                pf_acc ^= v.x ^ v.y ^ v.z ^ v.w;
            }
    }
    // epilogue operands of wave 0, requested before the staging (their round trip runs under it)
    const int cobase = ct0 * 32 + 4 * (lane >> 5);
    const int Lqb = item_rows(a.lens, b, a.lmul_q, a.Lq);
    f32x4 bvec[MI][4], rvec[MI][NJ][4];
    if (wave == 0) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = min(cobase + mi * 32 + 8 * g, a.cout - 4);
                bvec[mi][g] = a.bias ? *(const f32x4*)(a.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
                if (a.cb) bvec[mi][g] += *(const f32x4*)(a.cb + (size_t)b * a.cout + co);
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt) {
                    const size_t orow = (size_t)min(q0 + jt * 32 + (lane & 31), Lqb - 1) * a.out_mul + a.out_add;
                    rvec[mi][jt][g] = (a.res && a.out_mode != OUT_ACT) ? *(const f32x4*)(a.res + (size_t)b * a.res_bstride + orow * a.out_C + co)
                                                                      : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
    }
    stage_tile<OpT, CIN>(smem, a, b, q0 + a.in_off - a.roff, a.tile_rows);
    if (pf_acc == 0x9e3779b9u && a.Lq < 0) *(unsigned*)smem = pf_acc;  // never true: keeps the prefetch loads alive
    __syncthreads();

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][jt][r] = 0.f;
    const char* lds_lane = smem + (size_t)(lane & 31) * TL::STRIDE + (lane >> 5) * 16;
    if (t1 > t0)  // (wave-uniform)
        conv_core<OpT, CIN, MI, NJ, KGROUP, 4>(acc, lds_lane, wbase + (size_t)ct0 * a.w_ct_stride + (size_t)t0 * CC * 512 + lane * 8, a.w_ct_stride,
                                               t1 - t0, a.roff + t0 * a.dstep, a.dstep);
    // partial sums of waves 1..3 -> LDS behind the operand tile ([3][MI][NJ][16][64] floats)
    float* red = (float*)(smem + (size_t)a.tile_rows * TL::STRIDE);
    if (wave > 0) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((((wave - 1) * MI + mi) * NJ + jt) * 16 + r) * 64 + lane] = acc[mi][jt][r];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int jt = 0; jt < NJ; ++jt) {
        const int q = q0 + jt * 32 + (lane & 31);
        if (q >= Lqb) continue;
        const size_t orow = (size_t)q * a.out_mul + a.out_add;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = cobase + mi * 32 + 8 * g;
                if (co >= a.cout) continue;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    v[e] = ((acc[mi][jt][r] + red[(((0 * MI + mi) * NJ + jt) * 16 + r) * 64 + lane]) + red[(((1 * MI + mi) * NJ + jt) * 16 + r) * 64 + lane]) +
                           red[(((2 * MI + mi) * NJ + jt) * 16 + r) * 64 + lane];
                }
                v += bvec[mi][g];
                if (a.out_mode == OUT_ACT) {
                    using o4 = __attribute__((ext_vector_type(4))) OpT;
                    o4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = to_op<OpT>(lrelu(v[e], a.slope_out));
                    *(o4*)((OpT*)a.out + (size_t)b * a.out_bstride + orow * a.out_C + co) = o;
                } else {
                    if (a.res) v += rvec[mi][jt][g];
                    f32x4* o = (f32x4*)((float*)a.out + (size_t)b * a.out_bstride + orow * a.out_C + co);
                    if (a.accumulate) v += *o;
                    *o = v;
                }
            }
    }
}
template <typename OpT, int CIN, int MI, int NJ>
static __global__ void __launch_bounds__(256) k_conv_ks_jobs(ConvJobs js) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int j = (int)blockIdx.z % js.njobs, b = (int)blockIdx.z / js.njobs;
    conv_ks_body<OpT, CIN, MI, NJ>(js.job[j], b, smem);
}

template <typename OpT, int CIN, int MI, int NJ, int WCO>
static __global__ void __launch_bounds__(256) k_conv_mfma_jobs(ConvJobs js) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int j = (int)blockIdx.z % js.njobs, b = (int)blockIdx.z / js.njobs;
    // a weight ring of 4 groups (12 k-steps in flight): with one MFMA per k-step (NJ = 1) the default 2 groups cover 4 k-steps =
    // ~260 cycles, less than one L2 round trip, and the K loop ran at the latency of the weight loads (28 us per launch)
    conv_mfma_body<OpT, CIN, MI, NJ, WCO, 4, true>(js.job[j], b, smem);
}

// Branch-free helpers.  lrelu as max(x, slope*x) (0 < slope < 1); row masking by AND-ing the value bits so that
// the compiler cannot turn the select into divergent control flow (it did: ~45 exec-masked blocks per publish).
// (fmaxf costs a third VALU op per value -- hipcc canonicalises the operand it cannot prove quiet, `v_max_f32 t, v, v` -- but an
//  inline-asm v_max_f32 in its place pins registers: rb_stream phases 3.0k -> 4.1k cycles, 13-29 spills.  Measured, not kept.)
// (round 2 wrote max(v, slope*v) as the median of {v, slope*v, +inf}: LLVM folds a med3 with an infinite operand back into
//  v_max_f32 WITH the canonicalising v_max in front -- the ISA still had 5.5 VALU per masked value.)
__device__ __forceinline__ float lrelu_max(float v, float slope) { return __builtin_amdgcn_fmed3f(v, v * slope, __builtin_inff()); }
__device__ __forceinline__ float mask_bits(float v, unsigned m) { return __uint_as_float(__float_as_uint(v) & m); }

// lrelu(v, 0.1) of a value that is about to become an MFMA operand, as ONE v_mul + ONE v_med3: the median of {v, 0.1 v, TOP}
// is max(v, 0.1 v) whenever that is <= TOP, and TOP otherwise -- for fp16 operands TOP = 65504 is the saturation to_op<> applies
// (identical results for every |v| < 655040; beyond that the fp16 conversion overflows to inf as the reference's .half() does),
// for bf16 TOP = FLT_MAX (no fold to v_max: a finite constant).  3 VALU per published value with the row mask applied to the
// PACKED pair (one v_and per two values) instead of 5.5.
template <typename OpT>
__device__ __forceinline__ float lrelu_op(float v) {
    return __builtin_amdgcn_fmed3f(v, v * 0.1f, 3.4028235e38f);
}
template <>
__device__ __forceinline__ float lrelu_op<_Float16>(float v) {
    return __builtin_amdgcn_fmed3f(v, v * 0.1f, 65504.f);
}
// four activated values -> 4 operands (8 bytes), rows outside the utterance zeroed through the packed words
template <typename OpT, bool MASK>
__device__ __forceinline__ uint2 pack4_lrelu(float a, float b, float c, float d, unsigned m) {
    using o4 = __attribute__((ext_vector_type(4))) OpT;
    o4 o = {(OpT)lrelu_op<OpT>(a), (OpT)lrelu_op<OpT>(b), (OpT)lrelu_op<OpT>(c), (OpT)lrelu_op<OpT>(d)};
    uint2 u = __builtin_bit_cast(uint2, o);
    if constexpr (MASK) {
        u.x &= m;
        u.y &= m;
    }
    return u;
}

// acc (MFMA D layout) -> lrelu -> OpT -> LDS operand tile.  `base` already points at this lane's (row, 4*(lane>>5))
// element of the wave's first row; everything else is a compile-time offset.
template <typename OpT, int C, int MI, int NJ, int STRIDE, bool MASK>
__device__ __forceinline__ void publish_operand(char* base, const f32x16 (&acc)[MI][NJ], const unsigned (&rowmask)[NJ],
                                                int cbase = 0) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if ((C % 32 == 0) || cbase + mi * 32 + 8 * g < C) {  // compile-time true for C % 32 == 0
                    const f32x16& t = acc[mi][jt];
                    *(uint2*)(base + jt * 32 * STRIDE + (mi * 32 + 8 * g) * 2) =
                        pack4_lrelu<OpT, MASK>(t[4 * g + 0], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3], rowmask[jt]);
                }
            }
}

// Block barrier that orders LDS traffic ONLY: __syncthreads() carries a workgroup release fence over global memory too, i.e. an
// s_waitcnt vmcnt(0) in front of every s_barrier -- which would drain the weight fragments requested for the next slot (a full
// L2 round trip, exposed four times per pair-step).  The waves of a block exchange data through LDS only.
// (Measured for k_rb_pair / k_rb_full, whose two-plus waves per SIMD already hide that round trip: +1 % time -- they keep __syncthreads.)
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}


// ------------------------------------------------------------------------------------------------
// Fused ResBlock1 pair (residuals.py:73-82):   x <- conv2(lrelu(conv1_dil(lrelu(x)) + b1)) + b2 + x
// ------------------------------------------------------------------------------------------------
//
// One launch covers pair level m of ALL resblocks of a stage (grid.y = resblock j; they read the same
// or sibling fp32 streams and are independent).  Per block, for one utterance and one time tile:
//   1. stage lrelu(x) as OpT into an LDS tile X of 128 + (k-1)*dil rows           (HBM read #1)
//   2. conv1 (dilated) on MFMA -> 128 rows of h; barrier
//   3. h -> lrelu -> OpT written back INTO THE SAME LDS region (X is dead; rows outside [0,L) are
//      written as zeros because conv2 zero-pads ITS input)                        barrier
//   4. conv2 (dil 1) on MFMA over the h tile -> 128-(k-1) valid output rows
//   5. epilogue: + b2 + residual x (fp32, HBM read #2) -> fp32 store              (HBM write)
// The intermediate never leaves the CU, and the only barriers are around step 3.  A block is
// C/64 waves (each wave owns a 64-channel output slice and streams its own weights), so 2-8 blocks
// share a CU and one block's staging/epilogue overlaps another block's MFMA work.

struct RbJob {
    const float* src;   // x  [B][L][C] fp32
    float* dst;         // x' [B][L][C] fp32
    const void* w1;     // packed conv1 weights
    const void* w2;
    const float* b1;
    const float* b2;
    long ct1, ct2;      // packed elements per 32-channel output tile
    int k_p;            // padded tap count (same for conv1 and conv2)
    int k;              // real kernel size
    int dil;            // conv1 dilation
    int tt2;            // valid outputs per tile = 128 - (k-1)
    int ntiles;
};
struct RbPairArgs {
    RbJob job[4];
    int L;
    const int* lens;  // ragged batch (item_rows)
    int lmul;
    long bstride;
    int dbg;  // timing ablations only (RVCMI_DBG): 2 skip conv1, 4 skip conv2, 16 skip store, 32 phase stamps
    unsigned long long* ts;
};

constexpr int RB_ROWS = 128;  // default conv1 output rows per tile = 4 MFMA column tiles per wave (template NJ overrides)

// Waves are laid out NW (output-channel slices of 32*MI) x NWT (time slabs of 32*NJ_ rows); OCC = waves per SIMD the
// register budget is capped for.
// (Round 5, measured and removed: the lean K loop kconv + LDS-only barriers of k_rb_stream for the C = 256 instantiation -- 253 registers, no
//  spills, parity-green -- ran 0.295 -> 0.320 ms per clip, ABAB x 3 on one lease: with two co-resident blocks at pair level 0 the group
//  structure of conv_run was never the limit, and kconv's 8-deep ring holds twice the weight registers.  DESIGN.md 4f.)
template <typename OpT, int C, int MI, int NW, int KG, int NJ_ = RB_ROWS / 32, int NWT = 1, int OCC = 2>
static __global__ void __launch_bounds__(64 * NW * NWT, OCC) k_rb_pair(RbPairArgs a) {
    constexpr int ROWS = 32 * NJ_ * NWT;  // conv1 output rows per tile
    using TL = Tile<C>;
    using frag = typename Op<OpT>::frag;
    constexpr int STRIDE = TL::STRIDE;
    constexpr int C8 = C / 8;
    constexpr int NT = 64 * NW * NWT;
    constexpr int NJ = NJ_;
    constexpr int NB = 2;
    constexpr int CP = 32 * MI * NW;  // padded channel count
    extern __shared__ __attribute__((aligned(16))) char smem[];

#ifdef RVCMI_RB_INTERLEAVE
    // dev experiment: co-resident blocks take different jobs (kernel sizes) so their load / MFMA phases drift apart
    const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int bjob = lin % gridDim.y, btile = lin / gridDim.y;
#else
    const int bjob = blockIdx.y, btile = blockIdx.x;
#endif
    const RbJob& J = a.job[bjob];
    if (btile >= J.ntiles) return;
    const int b = blockIdx.z;
    const int Lb = item_rows(a.lens, b, a.lmul, a.L);  // (a local: writing to the by-value argument struct sends it to scratch)
    if (btile * J.tt2 >= Lb) return;  // (ragged batch) a tile behind the item's end: block-uniform, before any barrier
    const int p2 = (J.k - 1) / 2;
    const int p1 = J.dil * (J.k - 1) / 2;
    const int t0 = btile * J.tt2;          // first output time of this tile
    const int h0 = t0 - p2;                     // global time of h row 0
    const int x0 = h0 - p1;                     // global time of X row 0
    const int xrows = ROWS + (J.k_p - 1) * J.dil;
    const float* src = J.src + (size_t)b * a.bstride;
    float* bias_l = (float*)(smem + (size_t)xrows * STRIDE);  // [2][CP]: b1 then b2
    unsigned long long* tsl = (unsigned long long*)(bias_l + 2 * CP);  // dev-only phase stamps (dbg & 32)
    int tsn = 0;
    auto stamp = [&]() {
#ifdef RVCMI_DEV_STAMPS
        if ((a.dbg & 32) && (threadIdx.x & 63) == 0 && tsn < 8) tsl[(threadIdx.x >> 6) * 8 + tsn] = __builtin_readcyclecounter();
#endif
        ++tsn;
    };
    stamp();  // 0

    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int ct0 = (wave % NW) * MI;             // first 32-channel output tile of this wave
    const int slab = (wave / NW) * 32 * NJ_;      // first row of this wave's time slab
    const int half4 = 4 * (lane >> 5);

    // conv1's first weight groups are requested before anything else: their latency hides behind the staging
    frag A[NB][KG][MI];
    conv_prefetch<OpT, C, MI, KG, NB>(A, (const OpT*)J.w1 + (size_t)ct0 * J.ct1 + lane * 8, J.ct1, J.k_p);

    // ---- 1. stage lrelu(x) -> OpT tile.  SB independent 32-byte loads in flight per thread (the accumulators are
    //         not live yet); clamped addresses keep every load unconditional (no exec-masked block per load).
#ifndef RB_SB
#define RB_SB 12  // 188-row tiles are 23.5 chunks per thread: two batches of 12 (three of 8 cost one more HBM round trip)
#endif
    constexpr int SB = RB_SB;
    const int total = xrows * C8;
    for (int base = threadIdx.x; base < total; base += SB * NT) {
        float4 lo[SB], hi[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int idx = min(base + u * NT, total - 1);  // past the tile: re-request its last chunk (an L1 hit), not the neighbour's rows
            const int r = idx / C8;
            const int c8 = idx - r * C8;
            const int gr = x0 + r;
            const int grc = min(max(gr, 0), Lb - 1);
            const float4* p = (const float4*)(src + (size_t)grc * C + c8 * 8);
            lo[u] = p[0];
            hi[u] = p[1];
            if (gr < 0 || gr >= Lb) {
                lo[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                hi[u] = lo[u];
            }
        }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int idx = base + u * NT;
            if (idx >= total) continue;
            const int r = idx / C8;
            const int c8 = idx - r * C8;
            const float f[8] = {lo[u].x, lo[u].y, lo[u].z, lo[u].w, hi[u].x, hi[u].y, hi[u].z, hi[u].w};
            frag v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (OpT)lrelu_op<OpT>(f[e]);
            *(frag*)(smem + (size_t)r * STRIDE + c8 * 16) = v;
        }
    }
    for (int i = threadIdx.x; i < 2 * CP; i += NT) {
        const int c = i % CP;
        bias_l[i] = c < C ? (i < CP ? J.b1[c] : J.b2[c]) : 0.f;
    }
    stamp();  // 1: staged
    __syncthreads();
    stamp();  // 2

    const char* lds_lane = smem + (size_t)(slab + (lane & 31)) * STRIDE + (lane >> 5) * 16;
    f32x16 acc[MI][NJ];
    auto init_bias = [&](const float* bl) {  // accumulators start from the bias: no add in the epilogues
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = *(const f32x4*)(bl + (ct0 + mi) * 32 + 8 * g + half4);
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[mi][jt][4 * g + e] = bv[e];
            }
    };

    // ---- 2. conv1 -----------------------------------------------------------------------------------
    init_bias(bias_l);
    if (!(a.dbg & 2))
        conv_run<OpT, C, MI, NJ, KG, NB>(acc, A, lds_lane, (const OpT*)J.w1 + (size_t)ct0 * J.ct1 + lane * 8, J.ct1, J.k_p, 0, J.dil);
    conv_prefetch<OpT, C, MI, KG, NB>(A, (const OpT*)J.w2 + (size_t)ct0 * J.ct2 + lane * 8, J.ct2, J.k_p);  // across the barriers
    stamp();  // 3: conv1 done
    __syncthreads();  // every wave has finished reading X

    // ---- 3. h = lrelu(conv1 + b1) -> OpT, in place over X (zero outside the utterance: conv2 pads ITS input) ----
    {
        const bool interior = h0 >= 0 && h0 + ROWS <= Lb;  // block-uniform: no masking needed
        unsigned rowmask[NJ];
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) {
            const int th = h0 + slab + jt * 32 + (lane & 31);
            rowmask[jt] = (th >= 0 && th < Lb) ? 0xffffffffu : 0u;
        }
        char* hw = smem + (size_t)(slab + (lane & 31)) * STRIDE + (ct0 * 32 + half4) * 2;
        if (interior) publish_operand<OpT, C, MI, NJ, STRIDE, false>(hw, acc, rowmask, ct0 * 32);
        else publish_operand<OpT, C, MI, NJ, STRIDE, true>(hw, acc, rowmask, ct0 * 32);
    }
    stamp();  // 4: h published
    __syncthreads();
    stamp();  // 5

    // ---- 4. conv2 (accumulators start from b2) -----------------------------------------------------------
    init_bias(bias_l + CP);
    if (!(a.dbg & 4))
        conv_run<OpT, C, MI, NJ, KG, NB>(acc, A, lds_lane, (const OpT*)J.w2 + (size_t)ct0 * J.ct2 + lane * 8, J.ct2, J.k_p, 0, 1);

    stamp();  // 6: conv2 done
    // ---- 5. epilogue: x' = (conv2 + b2) + x --------------------------------------------------------------
    float* dst = J.dst + (size_t)b * a.bstride;
#pragma unroll
    for (int jt = 0; jt < NJ; ++jt) {
        const int o = slab + jt * 32 + (lane & 31);
        const int t = t0 + o;
        const bool valid = o < J.tt2 && t < Lb;
        const int tc = min(t, Lb - 1);
        f32x4 r[MI][4];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if ((ct0 + mi) * 0 + mi * 32 + 8 * g < CP)  // always true; keeps the loads unconditional
                    r[mi][g] = (C % 32 == 0 || (ct0 + mi) * 32 + 8 * g + half4 < C)
                                   ? *(const f32x4*)(src + (size_t)tc * C + min((ct0 + mi) * 32 + 8 * g + half4, C - 4))
                                   : f32x4{0.f, 0.f, 0.f, 0.f};
        if (valid && !(a.dbg & 16)) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = (ct0 + mi) * 32 + 8 * g + half4;
                    if ((C % 32 != 0) && co >= C) continue;
                    f32x4 v = {acc[mi][jt][4 * g + 0], acc[mi][jt][4 * g + 1], acc[mi][jt][4 * g + 2], acc[mi][jt][4 * g + 3]};
                    v += r[mi][g];
                    *(f32x4*)(dst + (size_t)t * C + co) = v;
                }
        }
    }
    stamp();  // 7: stores issued
#ifdef RVCMI_DEV_STAMPS
    if ((a.dbg & 32) && (threadIdx.x & 63) < 8) {
        const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        a.ts[(blk * NW * NWT + (threadIdx.x >> 6)) * 8 + (threadIdx.x & 63)] = tsl[(threadIdx.x >> 6) * 8 + (threadIdx.x & 63)];
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// Polyphase transposed convolution + noise conv (nsf.py:171-174):
//   X0[q*u + r] = b + sum_j lrelu(x)[q + off_r - j] * Wt[:, :, cm_r + j*u]  +  noise_conv(har)[q*u + r]
// ------------------------------------------------------------------------------------------------
//
// x = ((in_a + in_b) + in_c) / div is the mean of the previous stage's resblocks (nsf.py:186).  The
// activated input tile is staged ONCE per block and shared by every phase r and output-channel group
// ("virtual tile") the block's waves walk through; the epilogue adds the bias and the strided 1->C
// noise convolution (VALU, k = 2*s taps of the har excitation) so X0 is written exactly once.
struct UpsArgs {
    const float* in_a;
    const float* in_b;
    const float* in_c;
    int in_half;  // 1: in_a / in_b / in_c are fp16 streams (pack4_h) with the same [B][Lin][cin] element layout
    int in_raw;   // 1 (round 6, stage 0): in_a is ALREADY the activated operand tile source -- conv_pre wrote to_op(lrelu(conv + bias + cond, 0.1)),
                  //    exactly what this staging would compute from its fp32 output (div = 1, one input) -- rows are copied, 16 bytes per chunk
    float div;
    int Lin, cin;
    long in_bstride;
    const void* w;
    long ct_stride;
    long ph_w_off[16];
    int ph_in_off[16];
    int ntaps_p, u, cout;
    int lo;          // lowest input-row offset any (phase, tap) touches relative to q
    int tile_rows;
    const float* bias;
    float* out;
    int out_half;  // 1: X0 is written as fp16 (pack4_h; option X0_F16, stages whose ResBlocks run on k_rb_full), same element layout
    int out_tr;    // 1: the block owns WHOLE output rows (every phase, every channel: gridDim.y == 1) and stores them row-wise through an LDS
                   //    tile at byte offset out_tr_off of the dynamic LDS: each D-layout store touched 8 (fp16) / 16 (fp32) bytes of 32 rows --
                   //    16 partial writes per 128-byte line, 1.6 TB/s -- the row-wise stores write whole lines
    int out_tr_off;
    int out_tw_off;  // > 0 (round 6; NJ == 1, fp16 X0, the block does NOT own whole rows -- stage 1 of v2/48k): byte offset of four wave-private
                     //    transposition tiles [32][MI * 64 + 16 B]: a wave's virtual tile (32 rows x MI * 32 channels) is written there in D layout
                     //    and stored from there as 16 bytes per lane, eight lanes = one whole 128-byte line of a row
    long out_bstride;
    const float* addend;  // optional [B][Lin*u][cout] fp32 added in the epilogue (noise conv done by the MFMA conv)
    const float* har;  // [B][Lh] or nullptr (no-f0 generator / addend in use)
    int Lh;
    int Lh_stride;     // elements between the items of har (= the longest item's Lh)
    const int* lens;   // ragged batch (item_rows): Lin = lens[b] * lmul, Lh = lens[b] * lhmul
    int lmul, lhmul;
    const float* Wn;   // [nk][cout]
    const float* bn;
    int nk, ns, npad;
    const void* wnz;   // nz_k1: noise weights packed as ONE MFMA k-step [co_tile][lane][8] (taps >= nk are zero)
    int nz_k1;         // 1: the <=16-tap noise conv runs as one extra k-step fed from an fp16 copy of the har span in LDS
    int nvt, cog, vpw;
    int dbg;  // timing ablations only: 1 skip staging loads, 2 skip MFMA, 4 skip noise conv, 16 skip store
    int bias_off;      // > 0: byte offset of a [2][cout] fp32 LDS copy of bias / bn, staged with the tile (round 5: the epilogue's bias
                       // vectors were a dependent global round trip behind the K loop of every virtual tile)
};

template <typename OpT, int CIN, int MI, int WV, int NJ = 4>
#ifndef UPS_OCC
#define UPS_OCC 2
#endif
static __global__ void __launch_bounds__(256, UPS_OCC) k_ups(UpsArgs a) {  // 2 blocks per CU: one block stages / stores while the other multiplies
    using TL = Tile<CIN>;
    using frag = typename Op<OpT>::frag;
    constexpr int STRIDE = TL::STRIDE;
    constexpr int C8 = CIN / 8;
    constexpr int WT = 4 / WV;
    constexpr int TQ = 32 * NJ * WT;
    constexpr int SB = 5;  // chunk loads (x 3 inputs) in flight per thread: 1280 chunks per pass cover the 1056 / 1048 of the two HBM-bound stages in ONE pass (8 cost the C_in = 64 instantiation its fourth resident block: 134 registers)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.z;
    const int q0 = blockIdx.x * TQ;
    const int Linb = item_rows(a.lens, b, a.lmul, a.Lin), Lhb = item_rows(a.lens, b, a.lhmul, a.Lh);  // (locals: see k_rb_pair)
    if (q0 >= Linb) return;  // (ragged batch) a tile behind the item's end: block-uniform, before the barrier
    const int g0 = q0 + a.lo;
    const size_t boff = (size_t)b * a.in_bstride;

    // nz_k1: the excitation samples this block's output rows can touch -- requested FIRST (two per thread cover the usual spans),
    // together with this wave's bias vectors: with the tile's own loads they are ONE memory round trip instead of three dependent
    // ones (a block is ~4 round trips long; measured 20 us per 64-row tile at 3 blocks per CU)
    const int us = a.u * a.ns;
    const long hbase = (long)q0 * us - a.npad;
    const int hspan = TQ * us + 16;
    float hpre[2] = {0.f, 0.f};
#ifdef UPS_NO_PRELOAD  // dev A/B: the round-3 order (har and bias loads behind the tile staging / the K loop)
    constexpr bool UPS_PRE = false;
#else
    constexpr bool UPS_PRE = true;
#endif
    if (a.nz_k1 && UPS_PRE) {
        const float* hp = a.har + (size_t)b * a.Lh_stride;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long idx = hbase + (int)threadIdx.x + 256 * j;
            const long idc = min(max(idx, 0L), (long)Lhb - 1);
            const float v = hp[idc];
            hpre[j] = (idx >= 0 && idx < Lhb && (int)threadIdx.x + 256 * j < hspan) ? v : 0.f;
        }
    }
    float* bias_l = (float*)(smem + a.bias_off);
    if (a.bias_off) {
        for (int i = threadIdx.x; i < 2 * a.cout; i += 256) bias_l[i] = i < a.cout ? a.bias[i] : (a.bn ? a.bn[i - a.cout] : 0.f);
    }
    const int total = a.tile_rows * C8;
    // Round 6: EVERY load of a batch is issued before the first one is used.  The loop used to read `load a; if (in_b) load b; if (in_c) load c`
    // per chunk; hipcc compiled each conditional load into its own block with `s_waitcnt vmcnt(0)` behind it -- 24 DEPENDENT memory round trips
    // per thread and tile, one 4 KB wave-load in flight per wave (found in the ISA after the timing ablations showed the staging alone at
    // 46 of ups_c128's 60 us = 2.4 TB/s, against 23 us for the same bytes through k_post_dma).  Now the pointers of absent streams alias in_a
    // (unconditional loads) and only the register adds are conditional.
    auto convert_store = [&](int idx, float (&f)[8], bool ok) {
        const int r = idx / C8;
        const int c8 = idx - r * C8;
        if (!ok) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = 0.f;
        }
        frag v;
        if (a.div == 3.f) {  // the three ResBlocks of every shipped config: the short exact quotient, one v_mul + one v_med3 lrelu
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (OpT)lrelu_op<OpT>(div3_exact(f[e]));
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float x = f[e];
                if (a.div != 1.f) x = x / a.div;
                v[e] = to_op<OpT>(lrelu(x, 0.1f));
            }
        }
        *(frag*)(smem + (size_t)r * STRIDE + c8 * 16) = v;
    };
    if (a.in_raw) {
        const OpT* pa = (const OpT*)a.in_a;
        constexpr int SR = 8;
        for (int base = threadIdx.x; base < total; base += SR * 256) {
            uint4 rr[SR];
#pragma unroll
            for (int u = 0; u < SR; ++u) {
                const int idx = min(base + u * 256, total - 1);
                const int r = idx / C8;
                const int c8 = idx - r * C8;
                const int grc = min(max(g0 + r, 0), Linb - 1);
                rr[u] = *(const uint4*)(pa + boff + (size_t)grc * CIN + c8 * 8);
            }
#pragma unroll
            for (int u = 0; u < SR; ++u) {
                const int idx = base + u * 256;
                if (idx >= total) continue;
                const int r = idx / C8;
                const int c8 = idx - r * C8;
                const int gr = g0 + r;
                *(uint4*)(smem + (size_t)r * STRIDE + c8 * 16) = (gr >= 0 && gr < Linb) ? rr[u] : uint4{0u, 0u, 0u, 0u};
            }
        }
    } else if (a.in_half) {  // fp16 streams: one 16-byte load per input and chunk
        const _Float16* pa = (const _Float16*)a.in_a;
        const _Float16* pb = a.in_b ? (const _Float16*)a.in_b : pa;
        const _Float16* pc = a.in_c ? (const _Float16*)a.in_c : pa;
        for (int base = threadIdx.x; base < total; base += SB * 256) {
            uint4 ra[SB], rb[SB], rc[SB];
            size_t off[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int idx = min(base + u * 256, total - 1);  // past the tile: re-request its last chunk (an L1 hit), not the neighbour's rows
                const int r = idx / C8;
                const int c8 = idx - r * C8;
                const int grc = min(max(g0 + r, 0), Linb - 1);  // clamped address: unconditional loads
                off[u] = boff + (size_t)grc * CIN + c8 * 8;
                ra[u] = *(const uint4*)(pa + off[u]);
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) rb[u] = *(const uint4*)(pb + off[u]);
#pragma unroll
            for (int u = 0; u < SB; ++u) rc[u] = *(const uint4*)(pc + off[u]);
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int idx = base + u * 256;
                if (idx >= total) continue;
                float f[8], t[8];
                unpack8_h(ra[u], f);
                if (a.in_b) {
                    unpack8_h(rb[u], t);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += t[e];
                }
                if (a.in_c) {
                    unpack8_h(rc[u], t);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += t[e];
                }
                const int gr = g0 + idx / C8;
                convert_store(idx, f, gr >= 0 && gr < Linb);
            }
        }
    } else {  // fp32 streams (stage 0 reads conv_pre's output, stage 1 the fp32 streams of k_rb_pair<256>): two 16-byte loads per input and chunk
        constexpr int SF = 4;
        const float* pb = a.in_b ? a.in_b : a.in_a;
        const float* pc = a.in_c ? a.in_c : a.in_a;
        for (int base = threadIdx.x; base < total; base += SF * 256) {
            float4 la[SF], ha[SF], lb[SF], hb[SF], lc[SF], hc[SF];
            size_t off[SF];
#pragma unroll
            for (int u = 0; u < SF; ++u) {
                const int idx = min(base + u * 256, total - 1);
                const int r = idx / C8;
                const int c8 = idx - r * C8;
                const int grc = min(max(g0 + r, 0), Linb - 1);
                off[u] = boff + (size_t)grc * CIN + c8 * 8;
                la[u] = *(const float4*)(a.in_a + off[u]);
                ha[u] = *(const float4*)(a.in_a + off[u] + 4);
            }
#pragma unroll
            for (int u = 0; u < SF; ++u) {
                lb[u] = *(const float4*)(pb + off[u]);
                hb[u] = *(const float4*)(pb + off[u] + 4);
            }
#pragma unroll
            for (int u = 0; u < SF; ++u) {
                lc[u] = *(const float4*)(pc + off[u]);
                hc[u] = *(const float4*)(pc + off[u] + 4);
            }
#pragma unroll
            for (int u = 0; u < SF; ++u) {
                const int idx = base + u * 256;
                if (idx >= total) continue;
                float f[8] = {la[u].x, la[u].y, la[u].z, la[u].w, ha[u].x, ha[u].y, ha[u].z, ha[u].w};
                if (a.in_b) {
                    f[0] += lb[u].x; f[1] += lb[u].y; f[2] += lb[u].z; f[3] += lb[u].w; f[4] += hb[u].x; f[5] += hb[u].y; f[6] += hb[u].z; f[7] += hb[u].w;
                }
                if (a.in_c) {
                    f[0] += lc[u].x; f[1] += lc[u].y; f[2] += lc[u].z; f[3] += lc[u].w; f[4] += hc[u].x; f[5] += hc[u].y; f[6] += hc[u].z; f[7] += hc[u].w;
                }
                const int gr = g0 + idx / C8;
                convert_store(idx, f, gr >= 0 && gr < Linb);
            }
        }
    }
    // nz_k1: fp16 copy of the excitation samples every output row of this block can touch (zero outside the signal)
    OpT* har16 = (OpT*)(smem + (size_t)a.tile_rows * STRIDE);
    if (a.nz_k1) {
        const float* hp = a.har + (size_t)b * a.Lh_stride;
        if (UPS_PRE) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if ((int)threadIdx.x + 256 * j < hspan) har16[threadIdx.x + 256 * j] = to_op<OpT>(hpre[j]);
        }
        for (int i = threadIdx.x + (UPS_PRE ? 512 : 0); i < hspan; i += 256) {  // (spans beyond 512 samples: large noise strides)
            const long idx = hbase + i;
            har16[i] = (idx >= 0 && idx < Lhb) ? to_op<OpT>(hp[idx]) : (OpT)0.f;
        }
    }
    __syncthreads();

    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int wv = wave % WV, wt = wave / WV;
    const int tw0 = wt * NJ * 32;
    const char* lds_lane = smem + (size_t)(tw0 + (lane & 31)) * STRIDE + (lane >> 5) * 16;
    const float* har = (a.har && !a.nz_k1) ? a.har + (size_t)b * a.Lh_stride : nullptr;  // VALU noise path only
    float* out = a.out + (size_t)b * a.out_bstride;

    for (int i = 0; i < a.vpw; ++i) {
        const int vt = ((int)blockIdx.y * WV + wv) * a.vpw + i;
        if (vt >= a.nvt) break;
        const int r = vt / a.cog;
        const int ct0 = (vt - r * a.cog) * MI;
        f32x16 acc[MI][NJ];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mi][jt][e] = 0.f;
        const OpT* wlane = (const OpT*)a.w + a.ph_w_off[r] + (size_t)ct0 * a.ct_stride + lane * 8;
        // (requesting the epilogue's bias vectors here, before the K loop, costs 78 registers = one resident block per CU:
        //  ups_c128 65 -> 78 us on one lease, ABAB.  Occupancy, not the number of dependent round trips, is what this kernel lives on.)
// (a 4-group weight ring for the one-tile waves: 148 -> 200 VGPRs, 3 -> 2 blocks per CU; C_in 256 unchanged, 128 slower)
        // (round 5, measured: requesting the first virtual tile's weights BEFORE the staging -- conv_prefetch hoisted, conv_run here -- costs
        //  154 -> 214 registers at C_in = 128 / 256, i.e. 3 -> 2 resident blocks, like every other value held across the staging)
        if (!(a.dbg & 2)) conv_core<OpT, CIN, MI, NJ>(acc, lds_lane, wlane, a.ct_stride, a.ntaps_p, a.ph_in_off[r] - a.lo, -1);
        if (a.nz_k1) {  // + noise_convs[i](har): one k-step, B = the row's 16-sample window (nsf.py:173-174)
            frag An[MI], Bn[NJ];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) An[mi] = *(const frag*)((const OpT*)a.wnz + (size_t)(ct0 + mi) * 512 + lane * 8);
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) {
                const int ql = tw0 + jt * 32 + (lane & 31);
                const unsigned* wp = (const unsigned*)(har16 + ql * us + r * a.ns + 8 * (lane >> 5));  // 4-byte aligned (ns even)
                union { unsigned u[4]; frag f; } cv;
#pragma unroll
                for (int e = 0; e < 4; ++e) cv.u[e] = wp[e];
                Bn[jt] = cv.f;
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt) acc[mi][jt] = Op<OpT>::mfma(An[mi], Bn[jt], acc[mi][jt]);
        }

        const float* addend = a.addend ? a.addend + (size_t)b * a.out_bstride : nullptr;
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) {
            const int q = q0 + tw0 + jt * 32 + (lane & 31);
            const bool valid = q < Linb;
            const int t = min(q, Linb - 1) * a.u + r;  // clamped: loads below stay unconditional
            // noise conv for this output row: nv[mi][g] (4 channels each) += har[t*s - pad + j] * Wn[j][co]
            f32x4 nv[MI][4];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = min((ct0 + mi) * 32 + 8 * g + 4 * (lane >> 5), a.cout - 4);
                    nv[mi][g] = addend ? *(const f32x4*)(addend + (size_t)t * a.cout + co) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            if (har && !(a.dbg & 4)) {
                const int hb = t * a.ns - a.npad;
                for (int j = 0; j < a.nk; ++j) {
                    const int hi = hb + j;
                    const float hv = (hi >= 0 && hi < Lhb) ? har[hi] : 0.f;
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int co = min((ct0 + mi) * 32 + 8 * g + 4 * (lane >> 5), a.cout - 4);
                            const f32x4 wv4 = *(const f32x4*)(a.Wn + (size_t)j * a.cout + co);
                            nv[mi][g] += wv4 * hv;
                        }
                }
            }
            if (valid && !(a.dbg & 16)) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int co = (ct0 + mi) * 32 + 8 * g + 4 * (lane >> 5);
                        if (co >= a.cout) continue;
                        f32x4 v = {acc[mi][jt][4 * g + 0], acc[mi][jt][4 * g + 1], acc[mi][jt][4 * g + 2], acc[mi][jt][4 * g + 3]};
                        const float* bsrc = a.bias_off ? bias_l : a.bias;
                        const float* bnsrc = a.bias_off ? bias_l + a.cout : a.bn;
                        v += *(const f32x4*)(bsrc + co);
                        if (har) v += nv[mi][g] + *(const f32x4*)(bnsrc + co);  // x + (noise_conv + its bias), nsf.py:173-174
                        else if (addend) v += nv[mi][g];                       // the addend already carries the noise bias
                        else if (a.nz_k1) v += *(const f32x4*)(bnsrc + co);
                        if (a.out_tr) {  // row (q - q0) * u + r of the block's output tile; row pitch = cout elements + 16 bytes
                            const int orow = (tw0 + jt * 32 + (lane & 31)) * a.u + r;
                            char* ot = smem + a.out_tr_off;
                            if (a.out_half) *(uint2*)(ot + (size_t)orow * (a.cout * 2 + 16) + co * 2) = pack4_h(v[0], v[1], v[2], v[3]);
                            else *(f32x4*)(ot + (size_t)orow * (a.cout * 4 + 16) + co * 4) = v;
                        } else if (NJ == 1 && a.out_tw_off) {
                            *(uint2*)(smem + a.out_tw_off + (wave * 32 + (lane & 31)) * (MI * 64 + 16) + (mi * 32 + 8 * g + 4 * (lane >> 5)) * 2) =
                                pack4_h(v[0], v[1], v[2], v[3]);
                        } else if (a.out_half) *(uint2*)((_Float16*)a.out + (size_t)b * a.out_bstride + (size_t)t * a.cout + co) = pack4_h(v[0], v[1], v[2], v[3]);
                        else *(f32x4*)(out + (size_t)t * a.cout + co) = v;
                    }
            }
        }
        if (NJ == 1 && a.out_tw_off && !(a.dbg & 16)) {
            // (wave-private: a wave's LDS operations execute in order, no barrier; each D-layout global store touched 8 bytes of 32 different rows)
            constexpr int TWP = MI * 64 + 16, CPR = MI * 4;
            const char* sc = smem + a.out_tw_off + wave * 32 * TWP;
            for (int i2 = lane; i2 < 32 * CPR; i2 += 64) {
                const int row = i2 / CPR, c = i2 - row * CPR;
                const int q = q0 + tw0 + row;
                if (q < Linb)
                    *(uint4*)((char*)((_Float16*)a.out + (size_t)b * a.out_bstride + ((size_t)q * a.u + r) * a.cout + ct0 * 32) + c * 16) =
                        *(const uint4*)(sc + row * TWP + c * 16);
            }
        }
    }
    if (a.out_tr) {  // (block-uniform) whole rows out of the LDS tile: 16 bytes per lane, consecutive lanes = consecutive chunks of a row
        __syncthreads();
        const int esz = a.out_half ? 2 : 4;
        const int pitch = a.cout * esz + 16, cpr = a.cout * esz / 16;  // bytes per tile row, 16-byte chunks per row
        const int nrows = min(TQ, Linb - q0) * a.u;                    // rows of this tile inside the item
        const char* ot = smem + a.out_tr_off;
        char* og = (char*)a.out + ((size_t)b * a.out_bstride + (size_t)q0 * a.u * a.cout) * esz;
        for (int i = threadIdx.x; i < nrows * cpr; i += 256) {
            const int row = i / cpr, c = i - row * cpr;
            *(uint4*)(og + ((size_t)row * cpr + c) * 16) = *(const uint4*)(ot + (size_t)row * pitch + c * 16);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Fully fused ResBlock1 (residuals.py:68-85) for C <= 64: all (conv1, conv2) pairs in ONE kernel
// ------------------------------------------------------------------------------------------------
//
// The fp32 residual stream x never leaves the register file: each of the block's 4 waves owns a slab
// of R/4 time rows x ALL channels as MFMA accumulators (conv2 accumulates straight onto x, which IS the
// residual add).  LDS holds only the two activated OpT operand tiles (X = lrelu(x), H = lrelu(conv1)),
// through which the waves exchange their halo rows.  The tile is computed "overlap-save": the block
// loads R rows, every pair runs on all R rows, edge garbage creeps inwards by (p1+p2) rows per pair and
// only the central R - 2*HL rows are stored (HL = sum of the pair halos).  Per resblock: one HBM read of
// x, one HBM write -- instead of one of each per PAIR -- and 2 barriers per pair.
struct RbFullJob {
    const float* src;
    float* dst;
    const void* w1[3];
    const void* w2[3];
    const float* b1[3];
    const float* b2[3];
    long ct1, ct2;
    int k, k_p;
    int dil[3];
    int nd;
    int HL;       // total one-sided halo of the resblock
    int tvalid;   // R - 2*HL
    int ntiles;
};
struct RbFullArgs {
    RbFullJob job[3];
    int L;
    const int* lens;  // ragged batch (item_rows)
    int lmul;
    long bstride;
    int dbg;
    int yh;  // 1: dst streams are fp16 (pack4_h), same element layout
    int xh;  // 1: src (X0, the start of the fp32 residual stream) is fp16 (option X0_F16)
    unsigned long long* ts;  // dbg & 32: per-wave s_memtime stamps [block][wave][16]
};

constexpr int RBF_G = 32;   // zero guard rows around X (>= max dilated half-width + one padded tap)
constexpr int RBF_G2 = 8;   // zero guard rows around H

template <typename OpT, int C, int MI, int NJ, int KG, int NB, int OCC = 1, int NWV = 4>
static __global__ void __launch_bounds__(64 * NWV, OCC) k_rb_full(RbFullArgs a) {
    constexpr int NT = 64 * NWV;  // NWV waves, each owning a slab of 32*NJ rows
    using TL = Tile<C>;
    constexpr int STRIDE = TL::STRIDE;
    constexpr int SLAB = 32 * NJ;
    constexpr int R = NWV * SLAB;
    constexpr int XROWS = R + 2 * RBF_G;
    constexpr int HROWS = R + 2 * RBF_G2;
    constexpr int CP = 32 * MI;  // padded channel count (C == 16 runs as one 32-channel tile)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* X = smem;
    char* H = smem + (size_t)XROWS * STRIDE;
    float* bias_l = (float*)(smem + (size_t)(XROWS + HROWS) * STRIDE);          // [nd][2][CP]
    unsigned long long* tsl = (unsigned long long*)(bias_l + 3 * 2 * CP);         // dev-only phase stamps

    const RbFullJob& J = a.job[blockIdx.y];
    if ((int)blockIdx.x >= J.ntiles) return;
    const int b = blockIdx.z;
    const int Lb = item_rows(a.lens, b, a.lmul, a.L);  // (a local: see k_rb_pair)
    const int tg0 = blockIdx.x * J.tvalid - J.HL;  // global time of tile row 0
    if (tg0 + J.HL >= Lb) return;  // (ragged batch) no valid row of this tile lies inside the item: block-uniform, before any barrier
    const float* src = J.src + (size_t)b * a.bstride;
    float* dst = J.dst + (size_t)b * a.bstride;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slab = wave * SLAB;
    const int half4 = 4 * (lane >> 5);

    int tsn = 0;
    auto stamp = [&]() {  // dev-only (-DRVCMI_DEV_STAMPS): kept in LDS and flushed once at the end (a global store
#ifdef RVCMI_DEV_STAMPS   // per stamp would sit in front of the next barrier's vmcnt(0))
        if ((a.dbg & 32) && lane == 0 && tsn < 16) tsl[wave * 16 + tsn] = __builtin_readcyclecounter();
#endif
        ++tsn;
    };
    stamp();  // 0

    // Rows of this lane's NJ column tiles that lie outside the utterance must read as zero in every operand tile
    // (each conv zero-pads ITS input).  Interior tiles (the vast majority) skip the masking altogether.
    const bool interior = tg0 >= 0 && tg0 + R <= Lb;  // block-uniform
    unsigned rowmask[NJ];
#pragma unroll
    for (int jt = 0; jt < NJ; ++jt) {
        const int tg = tg0 + slab + jt * 32 + (lane & 31);
        rowmask[jt] = (tg >= 0 && tg < Lb) ? 0xffffffffu : 0u;
    }

    // ---- load x into the accumulator layout ---------------------------------------------------------------
    // A D-layout global access touches 32 rows x 32 bytes per instruction and costs ~54 cycles of the CU's address path
    // (measured); with one block per CU nothing overlaps it.  TIO: the tile is read as whole rows
    // (1 KB contiguous per instruction) into the not-yet-used operand regions of LDS and picked up from there in D layout
    // (row stride C*4 + 16 bytes: 16 consecutive rows hit 16 different 16-byte bank groups); the store goes the same way back.
#ifndef RBF_TIO
#define RBF_TIO 1
#endif
    constexpr bool TIO = (C % 32 == 0) && (RBF_TIO != 0);
    constexpr int TS = C * 4 + 16;
    constexpr int C4 = C / 4;
    static_assert(!TIO || (size_t)R * TS <= (size_t)(XROWS + HROWS) * STRIDE, "transposition tile must fit the X | H regions");
    static_assert(!TIO || (R * C4) % NT == 0, "whole batches of row chunks");
    f32x16 xacc[MI][NJ];
    if (a.xh) {  // fp16 X0: half the bytes through HBM and through the LDS transposition
        const _Float16* srch = (const _Float16*)J.src + (size_t)b * a.bstride;
        if constexpr (TIO) {
            constexpr int C8 = C / 8, TSH = C * 2 + 16;
            static_assert((R * C8) % NT == 0, "whole batches of fp16 row chunks");
            constexpr int PERH = R * C8 / NT;
            uint4 v[PERH];
#pragma unroll
            for (int u = 0; u < PERH; ++u) {
                const int idx = (int)threadIdx.x + u * NT;
                const int row = idx / C8, c8 = idx - row * C8;
                const int tg = tg0 + row;
                const int tgc = min(max(tg, 0), Lb - 1);
                v[u] = *(const uint4*)(srch + (size_t)tgc * C + c8 * 8);
                if (tg < 0 || tg >= Lb) v[u] = make_uint4(0u, 0u, 0u, 0u);  // rows outside the utterance read as zero
            }
#pragma unroll
            for (int u = 0; u < PERH; ++u) {
                const int idx = (int)threadIdx.x + u * NT;
                const int row = idx / C8, c8 = idx - row * C8;
                *(uint4*)(smem + (size_t)row * TSH + c8 * 16) = v[u];
            }
            __syncthreads();
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        using h4 = __attribute__((ext_vector_type(4))) _Float16;
                        const h4 q = *(const h4*)(smem + (size_t)(slab + jt * 32 + (lane & 31)) * TSH + (mi * 32 + 8 * g + half4) * 2);
#pragma unroll
                        for (int e = 0; e < 4; ++e) xacc[mi][jt][4 * g + e] = (float)q[e];
                    }
            __syncthreads();  // the guard zeroing and the first publish below overwrite the tile
        } else {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt) {
                    const int tg = tg0 + slab + jt * 32 + (lane & 31);
                    const int tgc = min(max(tg, 0), Lb - 1);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (mi * 32 + 8 * g < C) {
                            using h4 = __attribute__((ext_vector_type(4))) _Float16;
                            const h4 q = *(const h4*)(srch + (size_t)tgc * C + mi * 32 + 8 * g + half4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) xacc[mi][jt][4 * g + e] = mask_bits((float)q[e], rowmask[jt]);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) xacc[mi][jt][4 * g + e] = 0.f;
                        }
                    }
                }
        }
    } else if constexpr (TIO) {
        constexpr int PER = R * C4 / NT;
        float4 v[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int idx = (int)threadIdx.x + u * NT;
            const int row = idx / C4, c4 = idx - row * C4;
            const int tg = tg0 + row;
            const int tgc = min(max(tg, 0), Lb - 1);
            v[u] = *(const float4*)(src + (size_t)tgc * C + c4 * 4);
            if (tg < 0 || tg >= Lb) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);  // rows outside the utterance read as zero
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int idx = (int)threadIdx.x + u * NT;
            const int row = idx / C4, c4 = idx - row * C4;
            *(float4*)(smem + (size_t)row * TS + c4 * 16) = v[u];
        }
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 q = *(const f32x4*)(smem + (size_t)(slab + jt * 32 + (lane & 31)) * TS + (mi * 32 + 8 * g + half4) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) xacc[mi][jt][4 * g + e] = q[e];
                }
        __syncthreads();  // the guard zeroing and the first publish below overwrite the tile
    } else
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) {
            const int tg = tg0 + slab + jt * 32 + (lane & 31);
            const int tgc = min(max(tg, 0), Lb - 1);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (mi * 32 + 8 * g < C) {  // folds to a constant once the loops are unrolled
                    const f32x4 v = *(const f32x4*)(src + (size_t)tgc * C + mi * 32 + 8 * g + half4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) xacc[mi][jt][4 * g + e] = mask_bits(v[e], rowmask[jt]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) xacc[mi][jt][4 * g + e] = 0.f;
                }
            }
        }
    // (the x loads above are in flight while this runs) zero the guard rows once -- nobody writes them afterwards --
    // and stage every bias vector of the resblock in LDS
    {
        constexpr int W = STRIDE / 16;  // 16-byte words per row
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (int i = threadIdx.x; i < 2 * RBF_G * W; i += NT) {
            const int r = i / W, c = i - r * W;
            const int row = r < RBF_G ? r : RBF_G + R + (r - RBF_G);
            *(uint4*)(X + (size_t)row * STRIDE + c * 16) = z;
        }
        for (int i = threadIdx.x; i < 2 * RBF_G2 * W; i += NT) {
            const int r = i / W, c = i - r * W;
            const int row = r < RBF_G2 ? r : RBF_G2 + R + (r - RBF_G2);
            *(uint4*)(H + (size_t)row * STRIDE + c * 16) = z;
        }
        for (int i = threadIdx.x; i < J.nd * 2 * CP; i += NT) {
            const int m = i / (2 * CP), w = (i / CP) & 1, c = i % CP;
            const float* bp = w ? J.b2[m] : J.b1[m];
            bias_l[i] = c < C ? bp[c] : 0.f;
        }
    }

    stamp();  // 1: x loaded
    typename Op<OpT>::frag A[NB][KG][MI];  // weight register ring, requested one phase ahead of its use
    conv_prefetch<OpT, C, MI, KG, NB>(A, (const OpT*)J.w1[0] + lane * 8, J.ct1, J.k_p);
    char* xw = X + (size_t)(RBF_G + slab + (lane & 31)) * STRIDE + half4 * 2;   // this lane's publish base in X
    char* hw = H + (size_t)(RBF_G2 + slab + (lane & 31)) * STRIDE + half4 * 2;  // ... and in H
    publish_operand<OpT, C, MI, NJ, STRIDE, false>(xw, xacc, rowmask);          // masked already
    stamp();  // 2
    __syncthreads();
    stamp();  // 3

    const char* xl = X + (size_t)(slab + (lane & 31)) * STRIDE + (lane >> 5) * 16;
    const char* hl = H + (size_t)(slab + (lane & 31)) * STRIDE + (lane >> 5) * 16;
    const int p2 = (J.k - 1) / 2;

    for (int m = 0; m < J.nd; ++m) {
        const int p1 = J.dil[m] * (J.k - 1) / 2;
        // ---- conv1 (dilated) -> h; the accumulators start from b1 ----------------------------------------
        f32x16 hacc[MI][NJ];
        {
            const float* bl = bias_l + (m * 2 + 0) * CP + half4;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bv = *(const f32x4*)(bl + mi * 32 + 8 * g);
#pragma unroll
                    for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
                        for (int e = 0; e < 4; ++e) hacc[mi][jt][4 * g + e] = bv[e];
                }
        }
        conv_run<OpT, C, MI, NJ, KG, NB>(hacc, A, xl, (const OpT*)J.w1[m] + lane * 8, J.ct1, J.k_p, RBF_G - p1, J.dil[m], a.dbg);
        conv_prefetch<OpT, C, MI, KG, NB>(A, (const OpT*)J.w2[m] + lane * 8, J.ct2, J.k_p);  // in flight across the publish + barrier
        if (m == 0) stamp();  // 4: conv1 done
        if (interior) publish_operand<OpT, C, MI, NJ, STRIDE, false>(hw, hacc, rowmask);
        else publish_operand<OpT, C, MI, NJ, STRIDE, true>(hw, hacc, rowmask);
        if (m == 0) stamp();  // 5
        __syncthreads();  // h complete; every wave is also done reading X
        if (m == 0) stamp();  // 6
        // ---- conv2 accumulates onto x: x <- (x + b2) + conv2(h) ------------------------------------------
        {
            const float* bl = bias_l + (m * 2 + 1) * CP + half4;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bv = *(const f32x4*)(bl + mi * 32 + 8 * g);
#pragma unroll
                    for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
                        for (int e = 0; e < 4; ++e) xacc[mi][jt][4 * g + e] += bv[e];
                }
        }
        conv_run<OpT, C, MI, NJ, KG, NB>(xacc, A, hl, (const OpT*)J.w2[m] + lane * 8, J.ct2, J.k_p, RBF_G2 - p2, 1, a.dbg);
        if (m + 1 < J.nd) conv_prefetch<OpT, C, MI, KG, NB>(A, (const OpT*)J.w1[m + 1] + lane * 8, J.ct1, J.k_p);
        if (m == 0) stamp();  // 7: conv2 done
        if (m + 1 < J.nd) {  // publish lrelu(x') for the next pair
            if (interior) publish_operand<OpT, C, MI, NJ, STRIDE, false>(xw, xacc, rowmask);
            else publish_operand<OpT, C, MI, NJ, STRIDE, true>(xw, xacc, rowmask);
            if (m == 0) stamp();  // 8
            __syncthreads();
            if (m == 0) stamp();  // 9
        }
    }
    stamp();  // 10: all pairs done

    // ---- store the valid centre of the tile ---------------------------------------------------------------
    if (a.yh) {  // fp16 output stream: half the LDS round trip and half the store instructions
        _Float16* dsth = (_Float16*)J.dst + (size_t)b * a.bstride;
        if constexpr (TIO) {
            constexpr int TSH = C * 2 + 16;
            constexpr int C8 = C / 8;
            __syncthreads();  // every wave is done with the operand tiles
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *(uint2*)(smem + (size_t)(slab + jt * 32 + (lane & 31)) * TSH + (mi * 32 + 8 * g + half4) * 2) =
                            pack4_h(xacc[mi][jt][4 * g + 0], xacc[mi][jt][4 * g + 1], xacc[mi][jt][4 * g + 2], xacc[mi][jt][4 * g + 3]);
            __syncthreads();
            if (!(a.dbg & 16)) {
                const int nrow = min(R - 2 * J.HL, Lb - (tg0 + J.HL));
                const int tot = nrow * C8;
                for (int i = threadIdx.x; i < tot; i += NT) {
                    const int r = i / C8, c8 = i - r * C8;
                    const int row = J.HL + r;
                    *(uint4*)(dsth + (size_t)(tg0 + row) * C + c8 * 8) = *(const uint4*)(smem + (size_t)row * TSH + c8 * 16);
                }
            }
        } else if (!(a.dbg & 16)) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt) {
                    const int row = slab + jt * 32 + (lane & 31);
                    const int tg = tg0 + row;
                    if (row >= J.HL && row < R - J.HL && tg < Lb) {
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            if (mi * 32 + 8 * g < C)
                                *(uint2*)(dsth + (size_t)tg * C + mi * 32 + 8 * g + half4) =
                                    pack4_h(xacc[mi][jt][4 * g + 0], xacc[mi][jt][4 * g + 1], xacc[mi][jt][4 * g + 2], xacc[mi][jt][4 * g + 3]);
                    }
                }
        }
    } else if constexpr (TIO) {
        __syncthreads();  // every wave is done with the operand tiles
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 q = {xacc[mi][jt][4 * g + 0], xacc[mi][jt][4 * g + 1], xacc[mi][jt][4 * g + 2], xacc[mi][jt][4 * g + 3]};
                    *(f32x4*)(smem + (size_t)(slab + jt * 32 + (lane & 31)) * TS + (mi * 32 + 8 * g + half4) * 4) = q;
                }
        __syncthreads();
        if (!(a.dbg & 16)) {
            const int nrow = min(R - 2 * J.HL, Lb - (tg0 + J.HL));  // valid rows of this tile that lie inside the utterance
            const int tot = nrow * C4;
            for (int i = threadIdx.x; i < tot; i += NT) {
                const int r = i / C4, c4 = i - r * C4;
                const int row = J.HL + r;
                *(float4*)(dst + (size_t)(tg0 + row) * C + c4 * 4) = *(const float4*)(smem + (size_t)row * TS + c4 * 16);
            }
        }
    } else if (!(a.dbg & 16)) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) {
                const int row = slab + jt * 32 + (lane & 31);
                const int tg = tg0 + row;
                if (row >= J.HL && row < R - J.HL && tg < Lb) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (mi * 32 + 8 * g < C) {  // folds to a constant once the loops are unrolled
                            const f32x4 v = {xacc[mi][jt][4 * g + 0], xacc[mi][jt][4 * g + 1], xacc[mi][jt][4 * g + 2], xacc[mi][jt][4 * g + 3]};
                            *(f32x4*)(dst + (size_t)tg * C + mi * 32 + 8 * g + half4) = v;
                        }
                    }
                }
            }
    }
    stamp();  // 11: stores issued
#ifdef RVCMI_DEV_STAMPS
    if ((a.dbg & 32) && lane < 16) {
        const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        a.ts[(blk * NWV + wave) * 16 + lane] = lane < tsn ? tsl[wave * 16 + lane] : 0ull;
    }
#endif
}

}  // namespace rvcmi
