// Device code of the synthesizer "front": everything SynthesizerTrnMsNSFsid.infer runs before the generator
// (SURVEY.md section 8f row 1), for gfx950.
//
//   TextEncoder            rvc/layers/encoders.py:134-159   (emb_phone + emb_pitch, 6 x [rel-pos MHA, LN, FFN k3, LN], proj)
//   MultiHeadAttention     rvc/layers/attentions.py:74-146  (window 10 relative keys / values, heads share the embeddings)
//   FFN                    rvc/layers/attentions.py:262-272
//   LayerNorm              rvc/layers/norms.py:20-23
//   z_p sampling           rvc/layers/synthesizers.py:182-183
//   flow, reverse          rvc/layers/residuals.py:214-238,254,319-321
//   WN + gate              rvc/layers/norms.py:96-124, rvc/layers/utils.py:47-55
//
// Same conventions as the generator kernels: activations CHANNELS-LAST [B][T][C] (fp32 streams, OpT = fp16/bf16
// operand copies), weights pre-packed in MFMA fragment order, v_mfma_f32_32x32x16 with fp32 accumulation, every
// elementwise step (bias, gates, masks, residual, LayerNorm, the coupling update) fused into the producing kernel.
// The channel Flip between coupling layers is folded into the packed weights (no data movement).
#pragma once
#include "nsf_kernels.hpp"

namespace rvcmi {

enum FrEpi : int {
    FR_EMB = 0,       // lrelu((acc + b + emb_pitch[pitch]) * sqrt(H), 0.1) * mask            -> fp32   encoders.py:142-148
    FR_QKV = 1,       // q/sqrt(dk) | k -> OpT [T][2H];  v -> OpT transposed [H][Tp]                   attentions.py:70-72,98
    FR_RES_LN = 2,    // LayerNorm(res + (acc + b)[* mask])                                   -> fp32   encoders.py:76-80
    FR_RELU_OP = 3,   // relu(acc + b) * mask                                                 -> OpT    attentions.py:263-268
    FR_PROJ_ZP = 4,   // (m, logs) = (acc + b) * mask; (m + exp(logs) * noise * 0.66666) * mask -> fp32  synthesizers.py:182-183
    FR_F32_MASK = 5,  // (acc + b) * mask                                                     -> fp32   residuals.py:222
    FR_COUPLE = 6     // x1 <- (x1 - (acc + b) * mask) * mask, in place in the flow stream               residuals.py:224-236
};

struct FrConvArgs {
    const void* in;      // fp32 [B][T][CIN] (in_op = 0) or OpT [B][T][CIN] (in_op = 1)
    int in_op;
    long in_bstride;     // elements
    int T;               // rows of this call
    int t_off;           // frame index of row 0: mask(row) = row + t_off < len[b]
    const long long* len;  // [B] valid frame counts (phone_lengths) or nullptr
    int premask;         // zero masked input rows (x * x_mask ahead of the conv)
    const void* w;       // packed weights
    long ct_stride;
    int ntaps, pad;
    const float* bias;   // indexed by ORIGINAL output channel
    int cout;            // original output channels
    float* out;          // fp32 [B][T][out_C]
    long out_bstride;
    int out_C;
    void* out_op;        // OpT output (FR_RELU_OP: [B][T][cout]; FR_QKV: q|k [B][T][2H])
    long out_op_bstride;
    // epilogue extras
    const float* res;    // FR_RES_LN residual, layout of `out`
    const float* gamma;
    const float* beta;
    int postmask;        // FR_RES_LN: multiply the conv result by the mask before the residual add (FFN, attentions.py:272)
    const long long* pitch;   // FR_EMB [B][T]
    const float* emb_pitch;   // [256][H]
    float scale;              // sqrt(H)
    void* vt;            // FR_QKV: OpT [B][H][Tp]
    long vt_bstride;
    int Tp;
    float qdiv;          // sqrt(dk)
    int H;               // hidden channels (FR_QKV split point, FR_PROJ_ZP pairing)
    const float* noise;  // FR_PROJ_ZP: [B][C][T] channel-first (what randn_like(m_p) is)
    int phys_base;       // FR_COUPLE: first physical channel of the x1 half (flip folded)
};

__device__ __forceinline__ bool fr_valid(const FrConvArgs& a, int b, int row) {
    return a.len == nullptr || (long long)(row + a.t_off) < a.len[b];
}

// Stage rows [g0, g0+rows) of the input as OpT into the LDS tile; zero outside [0,T) and (premask) beyond the length.
template <typename OpT, int CIN, int NT>
__device__ __forceinline__ void fr_stage(char* smem, const void* in, int in_op, long boff, int T, int g0, int rows,
                                         int lenrow /* rows >= lenrow are zeroed */) {
    using frag = typename Op<OpT>::frag;
    constexpr int STRIDE = Tile<CIN>::STRIDE;
    constexpr int C8 = CIN / 8;
    const int hi = min(T, lenrow);
    for (int idx = threadIdx.x; idx < rows * C8; idx += NT) {
        const int r = idx / C8;
        const int c8 = idx - r * C8;
        const int gr = g0 + r;
        const bool ok = gr >= 0 && gr < hi;
        const int grc = min(max(gr, 0), T - 1);  // clamped address: unconditional loads
        frag v;
        if (in_op) {
            v = *(const frag*)((const OpT*)in + boff + (size_t)grc * CIN + c8 * 8);
        } else {
            const float4* p = (const float4*)((const float*)in + boff + (size_t)grc * CIN + c8 * 8);
            const float4 lo = p[0], hi4 = p[1];
            const float f[8] = {lo.x, lo.y, lo.z, lo.w, hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = to_op<OpT>(f[e]);
        }
        if (!ok) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (OpT)0.f;
        }
        *(frag*)(smem + (size_t)r * STRIDE + c8 * 16) = v;
    }
}

// Generic fused conv layer.  Block = NW waves, wave w owns MI consecutive packed 32-channel tiles, all waves share the
// NJ*32-row time tile.  Grid: x = time tile, y = block of NW*MI tiles, z = utterance.
template <typename OpT, int CIN, int MI, int NJ, int NW, int EPI>
static __global__ void __launch_bounds__(64 * NW) k_fr_conv(FrConvArgs a) {
    using TL = Tile<CIN>;
    constexpr int STRIDE = TL::STRIDE;
    constexpr int NT = 64 * NW;
    constexpr int TT = NJ * 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.z;
    const int q0 = blockIdx.x * TT;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rows = TT + a.ntaps - 1 + 2;  // + slack for the K loop's one-step-ahead reads
    const int lenrow = (a.premask && a.len) ? (int)min((long long)a.T, a.len[b] - a.t_off) : a.T;
    fr_stage<OpT, CIN, NT>(smem, a.in, a.in_op, (long)b * a.in_bstride, a.T, q0 - a.pad, rows, lenrow);
    __syncthreads();

    const int ct0 = ((int)blockIdx.y * NW + wave) * MI;
    f32x16 acc[MI][NJ];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][jt][e] = 0.f;
    const char* lds_lane = smem + (size_t)(lane & 31) * STRIDE + (lane >> 5) * 16;
    const OpT* wlane = (const OpT*)a.w + (size_t)ct0 * a.ct_stride + lane * 8;
    conv_core<OpT, CIN, MI, NJ>(acc, lds_lane, wlane, a.ct_stride, a.ntaps, 0, 1);

    const int hl = lane >> 5;
    if constexpr (EPI == FR_RES_LN) {
        // the block holds all cout (= NW*MI*32) channels of its rows: LayerNorm statistics go through LDS
        static_assert(MI == 1, "LN epilogue: one tile per wave");
        __syncthreads();  // staging tile is dead
        float* red = (float*)smem;  // [2][NW][TT]
        float v[NJ][16];
        const int cb = ct0 * 32 + 4 * hl;
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) {
            const int t = q0 + jt * 32 + (lane & 31);
            const int tc = min(t, a.T - 1);
            const float mk = (a.postmask && !fr_valid(a, b, tc)) ? 0.f : 1.f;
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = cb + 8 * g;
                const f32x4 bv = *(const f32x4*)(a.bias + co);
                const f32x4 rv = *(const f32x4*)(a.res + (size_t)b * a.out_bstride + (size_t)tc * a.out_C + co);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float y = (acc[0][jt][4 * g + e] + bv[e]) * mk;
                    v[jt][4 * g + e] = rv[e] + y;
                    s += v[jt][4 * g + e];
                }
            }
            s += __shfl_xor(s, 32, 64);
            if (hl == 0) red[(wave * NJ + jt) * 32 + (lane & 31)] = s;
        }
        __syncthreads();
        float mean[NJ];
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += red[(w * NJ + jt) * 32 + (lane & 31)];
            mean[jt] = tot / (float)a.cout;
            float s2 = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                v[jt][e] -= mean[jt];
                s2 += v[jt][e] * v[jt][e];
            }
            s2 += __shfl_xor(s2, 32, 64);
            if (hl == 0) red[NW * NJ * 32 + (wave * NJ + jt) * 32 + (lane & 31)] = s2;
        }
        __syncthreads();
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) {
            const int t = q0 + jt * 32 + (lane & 31);
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += red[NW * NJ * 32 + (w * NJ + jt) * 32 + (lane & 31)];
            const float rstd = 1.f / sqrtf(tot / (float)a.cout + 1e-5f);
            if (t < a.T) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = cb + 8 * g;
                    const f32x4 ga = *(const f32x4*)(a.gamma + co), be = *(const f32x4*)(a.beta + co);
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = v[jt][4 * g + e] * rstd * ga[e] + be[e];
                    *(f32x4*)(a.out + (size_t)b * a.out_bstride + (size_t)t * a.out_C + co) = o;
                }
            }
        }
    } else if constexpr (EPI == FR_PROJ_ZP) {
        static_assert(MI == 2, "paired (m, logs) tiles");
        const int pair = ct0 / 2;
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) {
            const int t = q0 + jt * 32 + (lane & 31);
            if (t >= a.T) continue;
            const float mk = fr_valid(a, b, t) ? 1.f : 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = pair * 32 + 8 * g + 4 * hl;
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float m = (acc[0][jt][4 * g + e] + a.bias[c + e]) * mk;
                    const float lg = (acc[1][jt][4 * g + e] + a.bias[a.H + c + e]) * mk;
                    const float nz = a.noise[((size_t)b * a.H + c + e) * a.T + t];
                    o[e] = (m + expf(lg) * nz * 0.66666f) * mk;
                }
                *(f32x4*)(a.out + (size_t)b * a.out_bstride + (size_t)t * a.out_C + c) = o;
            }
        }
    } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) {
                const int t = q0 + jt * 32 + (lane & 31);
                if (t >= a.T) continue;
                const float mk = fr_valid(a, b, t) ? 1.f : 0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = (ct0 + mi) * 32 + 8 * g + 4 * hl;
                    if (co >= a.cout) continue;
                    f32x4 v = {acc[mi][jt][4 * g + 0], acc[mi][jt][4 * g + 1], acc[mi][jt][4 * g + 2], acc[mi][jt][4 * g + 3]};
                    v += *(const f32x4*)(a.bias + co);
                    if constexpr (EPI == FR_EMB) {
                        const long long pi = a.pitch ? a.pitch[(size_t)b * a.T + t] : 0;
                        if (a.pitch) v += *(const f32x4*)(a.emb_pitch + (size_t)pi * a.cout + co);
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = lrelu(v[e] * a.scale, 0.1f) * mk;
                        *(f32x4*)(a.out + (size_t)b * a.out_bstride + (size_t)t * a.out_C + co) = o;
                    } else if constexpr (EPI == FR_QKV) {
                        using o4 = __attribute__((ext_vector_type(4))) OpT;
                        if (co < 2 * a.H) {
                            o4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = to_op<OpT>(co < a.H ? v[e] / a.qdiv : v[e]);
                            *(o4*)((OpT*)a.out_op + (size_t)b * a.out_op_bstride + (size_t)t * (2 * a.H) + co) = o;
                        } else {
                            OpT* vt = (OpT*)a.vt + (size_t)b * a.vt_bstride + (size_t)(co - 2 * a.H) * a.Tp + t;
#pragma unroll
                            for (int e = 0; e < 4; ++e) vt[(size_t)e * a.Tp] = to_op<OpT>(v[e]);
                        }
                    } else if constexpr (EPI == FR_RELU_OP) {
                        using o4 = __attribute__((ext_vector_type(4))) OpT;
                        o4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = to_op<OpT>(fmaxf(v[e], 0.f) * mk);
                        *(o4*)((OpT*)a.out_op + (size_t)b * a.out_op_bstride + (size_t)t * a.cout + co) = o;
                    } else if constexpr (EPI == FR_F32_MASK) {
                        *(f32x4*)(a.out + (size_t)b * a.out_bstride + (size_t)t * a.out_C + co) = v * mk;
                    } else if constexpr (EPI == FR_COUPLE) {
                        f32x4* xp = (f32x4*)(a.out + (size_t)b * a.out_bstride + (size_t)t * a.out_C + a.phys_base + co);
                        *xp = (*xp - v * mk) * mk;
                    }
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------
// One WN layer (norms.py:104-123) in one launch: in_layer (k taps) -> + cond -> tanh * sigmoid -> LDS -> res_skip 1x1
// -> x' = (x + res) * mask, skip += skip part.
// ------------------------------------------------------------------------------------------------
struct FrWnArgs {
    const float* x;     // [B][T][H] fp32 (already masked)
    float* x_out;       // other ping-pong buffer (not used by the last layer)
    float* skip;        // [B][T][H] fp32 running sum of the skip parts ("output" in the reference)
    int first;          // 1: skip is written, not accumulated
    long bstride;
    int T, t_off;
    const long long* len;
    const void* w_in;   // packed [tanh tile 0, sigmoid tile 0, tanh tile 1, ...]
    long ct_in;
    int ntaps, pad;
    const float* b_in;  // [2H] original order
    const float* gc;    // cond_layer(g) slice of this layer: [B][gc_bstride] -> 2H values, or nullptr
    long gc_bstride;
    const void* w_rs;   // packed [res tile 0, skip tile 0, ...] (last layer: [skip tile 0, skip tile 1, ...])
    long ct_rs;
    const float* b_rs;  // original order
};

template <typename OpT, int H, int NJ, bool LAST>
static __global__ void __launch_bounds__(64 * (H / 32)) k_fr_wn(FrWnArgs a) {
    using TL = Tile<H>;
    constexpr int STRIDE = TL::STRIDE;
    constexpr int NW = H / 32;
    constexpr int NT = 64 * NW;
    constexpr int TT = NJ * 32;
    using o4 = __attribute__((ext_vector_type(4))) OpT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.y;
    const int q0 = blockIdx.x * TT;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, hl = lane >> 5;
    const int xrows = TT + a.ntaps - 1 + 2;
    char* ACT = smem + (size_t)xrows * STRIDE;
    const long boff = (long)b * a.bstride;
    fr_stage<OpT, H, NT>(smem, a.x, 0, boff, a.T, q0 - a.pad, xrows, a.T);
    // slack rows of the ACT tile (read one k-step ahead by the second K loop) must hold finite values
    for (int i = threadIdx.x; i < 2 * STRIDE / 4; i += NT) ((unsigned*)(ACT + (size_t)TT * STRIDE))[i] = 0u;
    __syncthreads();

    const char* lds_lane = smem + (size_t)(lane & 31) * STRIDE + hl * 16;
    {
        f32x16 acc[2][NJ];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mi][jt][e] = 0.f;
        const OpT* wlane = (const OpT*)a.w_in + (size_t)(2 * wave) * a.ct_in + lane * 8;
        conv_core<OpT, H, 2, NJ>(acc, lds_lane, wlane, a.ct_in, a.ntaps, 0, 1);
        const float* gc = a.gc ? a.gc + (size_t)b * a.gc_bstride : nullptr;
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = wave * 32 + 8 * g + 4 * hl;
                o4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float ta = acc[0][jt][4 * g + e] + a.b_in[c + e];
                    float sa = acc[1][jt][4 * g + e] + a.b_in[H + c + e];
                    if (gc) {
                        ta += gc[c + e];
                        sa += gc[H + c + e];
                    }
                    o[e] = to_op<OpT>(tanhf(ta) * (1.f / (1.f + expf(-sa))));
                }
                *(o4*)(ACT + (size_t)(jt * 32 + (lane & 31)) * STRIDE + c * 2) = o;
            }
    }
    __syncthreads();

    constexpr int MI2 = LAST ? 1 : 2;
    f32x16 acc[MI2][NJ];
#pragma unroll
    for (int mi = 0; mi < MI2; ++mi)
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][jt][e] = 0.f;
    const char* act_lane = ACT + (size_t)(lane & 31) * STRIDE + hl * 16;
    const OpT* wlane = (const OpT*)a.w_rs + (size_t)(MI2 * wave) * a.ct_rs + lane * 8;
    conv_core<OpT, H, MI2, NJ>(acc, act_lane, wlane, a.ct_rs, 1, 0, 1);
#pragma unroll
    for (int jt = 0; jt < NJ; ++jt) {
        const int t = q0 + jt * 32 + (lane & 31);
        if (t >= a.T) continue;
        const float mk = (a.len == nullptr || (long long)(t + a.t_off) < a.len[b]) ? 1.f : 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = wave * 32 + 8 * g + 4 * hl;
            const size_t o = boff + (size_t)t * H + c;
            if constexpr (!LAST) {
                f32x4 r = {acc[0][jt][4 * g + 0], acc[0][jt][4 * g + 1], acc[0][jt][4 * g + 2], acc[0][jt][4 * g + 3]};
                r += *(const f32x4*)(a.b_rs + c);
                *(f32x4*)(a.x_out + o) = (*(const f32x4*)(a.x + o) + r) * mk;
            }
            f32x4 s = {acc[MI2 - 1][jt][4 * g + 0], acc[MI2 - 1][jt][4 * g + 1], acc[MI2 - 1][jt][4 * g + 2], acc[MI2 - 1][jt][4 * g + 3]};
            s += *(const f32x4*)(a.b_rs + (LAST ? 0 : H) + c);
            if (!a.first) s += *(const f32x4*)(a.skip + o);
            *(f32x4*)(a.skip + o) = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Relative-position multi-head self-attention (attentions.py:88-139), flash style.
// ------------------------------------------------------------------------------------------------
//
// Block = (32-query tile, head, utterance), 4 waves; wave w walks key tiles w, w+4, ... with its own online softmax
// state; the four partial (max, sum, O) are merged through LDS.
//   S^T tile   = K_tile (A: 32 keys x dk) . Q^T (B: dk x 32 queries)      6 MFMAs, lane holds 16 keys of ITS query
//   rel. keys  : S[q][j] += R[q][j-q+ws], R = Q . E_k^T computed once per block by MFMA (the 2ws+1 band)
//   P          = exp(S - m) packed to OpT straight from the accumulator registers: with the k-slot order
//                {4h..4h+3, 8+4h..8+4h+3} the D layout of S^T IS the B-fragment layout of P, so P never leaves VGPRs;
//   O^T       += V^T (A: dk x 32 keys, read with the same slot order from the transposed V) . P
//   rel. values: the 2ws+1 band of p is recomputed in fp32 at the end (21 dots per query) and applied to E_v.
struct FrAttnArgs {
    const void* qk;     // OpT [B][T][2H]: q (pre-scaled) | k
    const void* vt;     // OpT [B][H][Tp]
    void* out;          // OpT [B][T][H]
    const void* relk;   // packed E_k: [dk/16][64][8] OpT (A-fragment order, rows >= 2ws+1 zero)
    const float* relv;  // [2ws+1][dk] fp32
    const long long* len;
    int T, Tp, H, ws;
    long qk_bstride, vt_bstride, out_bstride;
};

template <typename OpT, int DK>
static __global__ void __launch_bounds__(256) k_fr_attn(FrAttnArgs a) {
    using frag = typename Op<OpT>::frag;
    constexpr int KS = DK / 16;   // k-steps of the score product
    constexpr int DT = DK / 32;   // 32-row tiles of O^T
    constexpr int OS = DK + 1;    // LDS row stride of the merge buffers
    static_assert(DK % 32 == 0, "head dim must be a multiple of 32");
    __shared__ float Rl[32 * 33];
    __shared__ float Ml[4 * 32], Ll[4 * 32];
    __shared__ float Ol[4 * 32 * OS];
    __shared__ float Pb[32 * 32];
    __shared__ float Mf[32], Lf[32];
    const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int q0 = qt * 32;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, hl = lane >> 5, ql = lane & 31;
    const int T = a.T;
    const int len = a.len ? (int)min((long long)T, a.len[b]) : T;
    const OpT* QK = (const OpT*)a.qk + (size_t)b * a.qk_bstride;
    const OpT* VT = (const OpT*)a.vt + (size_t)b * a.vt_bstride + (size_t)h * DK * a.Tp;
    const int H2 = 2 * a.H;

    frag Bq[KS];
    {
        const OpT* qp = QK + (size_t)min(q0 + ql, T - 1) * H2 + h * DK + 8 * hl;
#pragma unroll
        for (int s = 0; s < KS; ++s) Bq[s] = *(const frag*)(qp + 16 * s);
    }
    if (wave == 0) {  // R[q][r] = q . E_k[r]
        f32x16 r = {0};
#pragma unroll
        for (int s = 0; s < KS; ++s) r = Op<OpT>::mfma(*(const frag*)((const OpT*)a.relk + (size_t)s * 512 + lane * 8), Bq[s], r);
#pragma unroll
        for (int i = 0; i < 16; ++i) Rl[ql * 33 + (i & 3) + 8 * (i >> 2) + 4 * hl] = r[i];
    }
    __syncthreads();

    const int q = q0 + ql;
    const bool qok = q < len;
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 O[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) O[d][i] = 0.f;
    const int nkt = (T + 31) / 32;
    for (int kt = wave; kt < nkt; kt += 4) {
        const int j0 = kt * 32;
        f32x16 S = {0};
        {
            const OpT* kp = QK + (size_t)min(j0 + ql, T - 1) * H2 + a.H + h * DK + 8 * hl;
#pragma unroll
            for (int s = 0; s < KS; ++s) S = Op<OpT>::mfma(*(const frag*)(kp + 16 * s), Bq[s], S);
        }
        const bool near = (j0 >= q0 - 32 - a.ws) && (j0 <= q0 + 32 + a.ws);  // wave-uniform
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int j = j0 + (i & 3) + 8 * (i >> 2) + 4 * hl;
            float s = S[i];
            if (near) {
                const int rel = j - q + a.ws;
                const float rv = Rl[ql * 33 + min(max(rel, 0), 2 * a.ws)];
                s += (rel >= 0 && rel <= 2 * a.ws) ? rv : 0.f;
            }
            s = (qok && j < len) ? s : -1e4f;  // masked_fill(mask == 0, -1e4), attentions.py:115
            s = j < T ? s : -INFINITY;         // tile padding: not a key at all
            S[i] = s;
            mx = fmaxf(mx, s);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float sc = __expf(m_run - m_new);  // 0 on the first tile (m_run = -inf)
        float ps = 0.f;
        frag Bp[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float p = __expf(S[i] - m_new);
            ps += p;
            Bp[i >> 3][i & 7] = to_op<OpT>(p);
        }
        l_run = l_run * sc + ps;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int i = 0; i < 16; ++i) O[d][i] *= sc;
#pragma unroll
        for (int d = 0; d < DT; ++d) {
            const OpT* vp = VT + (size_t)(d * 32 + ql) * a.Tp + j0 + 4 * hl;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                union { uint2 u[2]; frag f; } av;
                av.u[0] = *(const uint2*)(vp + 16 * s2);
                av.u[1] = *(const uint2*)(vp + 16 * s2 + 8);
                O[d] = Op<OpT>::mfma(av.f, Bp[s2], O[d]);
            }
        }
    }
    l_run += __shfl_xor(l_run, 32, 64);
    if (hl == 0) {
        Ml[wave * 32 + ql] = m_run;
        Ll[wave * 32 + ql] = l_run;
    }
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) Ol[(wave * 32 + ql) * OS + d * 32 + (i & 3) + 8 * (i >> 2) + 4 * hl] = O[d][i];
    __syncthreads();
    if (threadIdx.x < 32) {
        const int x = threadIdx.x;
        float M = fmaxf(fmaxf(Ml[x], Ml[32 + x]), fmaxf(Ml[64 + x], Ml[96 + x]));
        float L = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) L += Ll[w * 32 + x] * __expf(Ml[w * 32 + x] - M);
        Mf[x] = M;
        Lf[x] = L;
    }
    __syncthreads();
    // band of p (relative values, attentions.py:127-135), recomputed in fp32 from the same operands
    const int nband = 2 * a.ws + 1;
    for (int idx = threadIdx.x; idx < 32 * nband; idx += 256) {
        const int x = idx / nband, r = idx - x * nband;
        const int qq = q0 + x, j = qq + r - a.ws;
        float p = 0.f;
        if (qq < T && j >= 0 && j < T) {
            const OpT* qp = QK + (size_t)qq * H2 + h * DK;
            const OpT* kp = QK + (size_t)j * H2 + a.H + h * DK;
            float s = 0.f;
            for (int d8 = 0; d8 < DK / 8; ++d8) {
                const frag qa = *(const frag*)(qp + d8 * 8), ka = *(const frag*)(kp + d8 * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) s = fmaf((float)qa[e], (float)ka[e], s);
            }
            s += Rl[x * 33 + r];
            s = (qq < len && j < len) ? s : -1e4f;
            p = __expf(s - Mf[x]) / Lf[x];
        }
        Pb[x * 32 + r] = p;
    }
    __syncthreads();
    OpT* out = (OpT*)a.out + (size_t)b * a.out_bstride;
    for (int idx = threadIdx.x; idx < 32 * DK; idx += 256) {
        const int x = idx / DK, d = idx - x * DK;
        if (q0 + x >= T) continue;
        const float M = Mf[x];
        float o = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) o += Ol[(w * 32 + x) * OS + d] * __expf(Ml[w * 32 + x] - M);
        o /= Lf[x];
        for (int r = 0; r < nband; ++r) o = fmaf(Pb[x * 32 + r], a.relv[r * DK + d], o);
        out[(size_t)(q0 + x) * a.H + h * DK + d] = to_op<OpT>(o);
    }
}

// z * x_mask, channels-last [B][T][C] -> the generator's channel-first [B][C][T]   (synthesizers.py:192)
static __global__ void __launch_bounds__(256) k_fr_out(const float* __restrict__ x, float* __restrict__ out, int T, int C, int t_off,
                                                const long long* __restrict__ len) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int t = t0 + r, c = c0 + tx;
        float v = 0.f;
        if (t < T && c < C) {
            v = x[((size_t)b * T + t) * C + c];
            if (len && (long long)(t + t_off) >= len[b]) v = 0.f;
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, t = t0 + tx;
        if (t < T && c < C) out[((size_t)b * C + c) * T + t] = tile[tx][r];
    }
}

}  // namespace rvcmi
