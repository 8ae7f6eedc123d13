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
    FR_COUPLE = 6,    // x1 <- (x1 - (acc + b) * mask) * mask, in place in the flow stream               residuals.py:224-236
    // round 6: one WN layer as TWO launches for small grids (a single clip = 38 time tiles on 256 CUs), so that the in_layer's weight
    // stream (737 KB, 84 % of the layer's) is split over 3x the blocks -- each block a third of the (tanh, sigmoid) channel pairs:
    FR_GATE = 7,      // tanh(acc_t + b_t + gc_t) * sigmoid(acc_s + b_s + gc_s) of a paired (tanh, sigmoid) tile -> OpT   norms.py:111-114, utils.py:47-55
    FR_WN_RS = 8,     // paired (res, skip) tiles: x' = (x + acc_r + b_r) * mask -> out; skip (+)= acc_s + b_s             norms.py:116-122
    FR_WN_RS_LAST = 9 // last layer: skip tiles only: skip (+)= acc + b                                                   norms.py:121-122
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
    void* vt;            // FR_QKV: v tiles, OpT [B][heads][Tp/32][dk/32][2][64][8] (A-fragment order of P.V)
    void* kf;            // FR_QKV: k tiles, OpT [B][heads][Tp/32][dk/16][64][8] (A-fragment order of K.Q^T)
    int Tp;
    float qdiv;          // sqrt(dk)
    int H;               // hidden channels (FR_QKV split point, FR_PROJ_ZP pairing)
    const float* noise;  // FR_PROJ_ZP: [B][C][T] channel-first (what randn_like(m_p) is)
    int phys_base;       // FR_COUPLE: first physical channel of the x1 half (flip folded)
    const float* gc;     // FR_GATE: cond_layer(g) slice of this layer, [B][gc_bstride] -> 2H values (or nullptr)
    long gc_bstride;
    float* skip;         // FR_WN_RS / _LAST: running sum of the skip parts, layout of `out`
    int first;           // FR_WN_RS / _LAST: 1 = skip is written, not accumulated
};

constexpr int FR_NB = 4;  // weight ring depth of the front kernels (k-groups)

// tanh / sigmoid through the hardware exp: the result is rounded to a 16-bit operand right after
__device__ __forceinline__ float fast_sigmoid(float v) { return 1.f / (1.f + __expf(-v)); }
__device__ __forceinline__ float fast_tanh(float v) { return 2.f * fast_sigmoid(2.f * v) - 1.f; }

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
    constexpr int SB = 8;  // independent chunk loads in flight per thread (a serial load->convert->store loop is latency-bound)
    const int total = rows * C8;
    for (int base = threadIdx.x; base < total; base += SB * NT) {
        frag v[SB];
        float4 lo[SB], hi4[SB];
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int idx = min(base + u * NT, total - 1);
            const int r = idx / C8;
            const int c8 = idx - r * C8;
            const int grc = min(max(g0 + r, 0), T - 1);  // clamped address: unconditional loads
            if (in_op) {
                v[u] = *(const frag*)((const OpT*)in + boff + (size_t)grc * CIN + c8 * 8);
            } else {
                const float4* p = (const float4*)((const float*)in + boff + (size_t)grc * CIN + c8 * 8);
                lo[u] = p[0];
                hi4[u] = p[1];
            }
        }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int idx = base + u * NT;
            if (idx < total) {
                const int r = idx / C8;
                const int c8 = idx - r * C8;
                const int gr = g0 + r;
                const bool ok = gr >= 0 && gr < hi;
                frag o;
                if (in_op) {
                    o = v[u];
                } else {
                    const float f[8] = {lo[u].x, lo[u].y, lo[u].z, lo[u].w, hi4[u].x, hi4[u].y, hi4[u].z, hi4[u].w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = to_op<OpT>(f[e]);
                }
                if (!ok) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (OpT)0.f;
                }
                *(frag*)(smem + (size_t)r * STRIDE + c8 * 16) = o;
            }
        }
    }
}

// fr_stage with the thread count as a runtime value (k_fr_gate_ks with a kernel size other than the shipped 5)
template <typename OpT, int CIN>
__device__ __forceinline__ void fr_stage_dyn(char* smem, const void* in, int in_op, long boff, int T, int g0, int rows, int lenrow, int NT) {
    using frag = typename Op<OpT>::frag;
    constexpr int STRIDE = Tile<CIN>::STRIDE, C8 = CIN / 8;
    const int hi = min(T, lenrow);
    for (int idx = threadIdx.x; idx < rows * C8; idx += NT) {
        const int r = idx / C8, c8 = idx - r * C8, gr = g0 + r, grc = min(max(gr, 0), T - 1);
        frag o;
        if (in_op) {
            o = *(const frag*)((const OpT*)in + boff + (size_t)grc * CIN + c8 * 8);
        } else {
            const float4* p = (const float4*)((const float*)in + boff + (size_t)grc * CIN + c8 * 8);
            const float4 lo = p[0], h4 = p[1];
            const float f[8] = {lo.x, lo.y, lo.z, lo.w, h4.x, h4.y, h4.z, h4.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = to_op<OpT>(f[e]);
        }
        if (!(gr >= 0 && gr < hi)) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (OpT)0.f;
        }
        *(frag*)(smem + (size_t)r * STRIDE + c8 * 16) = o;
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
    // these launches are small (tens of blocks at B = 1), so a wave's speed is set by how many weight bytes it keeps in
    // flight: a 4-deep ring (12 k-steps ahead), requested BEFORE the activation tile is staged
    const int ct0 = ((int)blockIdx.y * NW + wave) * MI;
    const OpT* wlane = (const OpT*)a.w + (size_t)ct0 * a.ct_stride + lane * 8;
    typename Op<OpT>::frag Aw[FR_NB][KGROUP][MI];
    conv_prefetch<OpT, CIN, MI, KGROUP, FR_NB>(Aw, wlane, a.ct_stride, a.ntaps);
    fr_stage<OpT, CIN, NT>(smem, a.in, a.in_op, (long)b * a.in_bstride, a.T, q0 - a.pad, rows, lenrow);
    __syncthreads();

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][jt][e] = 0.f;
    const char* lds_lane = smem + (size_t)(lane & 31) * STRIDE + (lane >> 5) * 16;
    conv_run<OpT, CIN, MI, NJ, KGROUP, FR_NB>(acc, Aw, lds_lane, wlane, a.ct_stride, a.ntaps, 0, 1);

    // ---- epilogue.  Every load is issued up front with clamped indices (hipcc turns a conditional load into an
    // exec-masked branch with its own wait; measured here: 15k cycles for a 24-load epilogue) ----
    const int hl = lane >> 5;
    const long long lenb = a.len ? a.len[b] : (long long)a.T + a.t_off;
    int tt[NJ], tcl[NJ];
    float mk[NJ];
#pragma unroll
    for (int jt = 0; jt < NJ; ++jt) {
        tt[jt] = q0 + jt * 32 + (lane & 31);
        tcl[jt] = min(tt[jt], a.T - 1);
        mk[jt] = (long long)(tcl[jt] + a.t_off) < lenb ? 1.f : 0.f;
    }
    if constexpr (EPI == FR_RES_LN) {
        // the block holds all cout (= NW*32) channels of its rows: LayerNorm statistics go through LDS
        static_assert(MI == 1, "LN epilogue: one tile per wave");
        const int cb = ct0 * 32 + 4 * hl;
        f32x4 bv[4], ga[4], be[4], rv[NJ][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bv[g] = *(const f32x4*)(a.bias + cb + 8 * g);
            ga[g] = *(const f32x4*)(a.gamma + cb + 8 * g);
            be[g] = *(const f32x4*)(a.beta + cb + 8 * g);
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt)
                rv[jt][g] = *(const f32x4*)(a.res + (size_t)b * a.out_bstride + (size_t)tcl[jt] * a.out_C + cb + 8 * g);
        }
        __syncthreads();  // staging tile is dead
        float* red = (float*)smem;  // [2][NW][TT]
        float v[NJ][16];
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) {
            const float pm = a.postmask ? mk[jt] : 1.f;
            float sum = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[jt][4 * g + e] = rv[jt][g][e] + (acc[0][jt][4 * g + e] + bv[g][e]) * pm;
                    sum += v[jt][4 * g + e];
                }
            sum += __shfl_xor(sum, 32, 64);
            if (hl == 0) red[(wave * NJ + jt) * 32 + (lane & 31)] = sum;
        }
        __syncthreads();
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += red[(w * NJ + jt) * 32 + (lane & 31)];
            const float mean = tot / (float)a.cout;
            float s2 = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                v[jt][e] -= mean;
                s2 += v[jt][e] * v[jt][e];
            }
            s2 += __shfl_xor(s2, 32, 64);
            if (hl == 0) red[NW * NJ * 32 + (wave * NJ + jt) * 32 + (lane & 31)] = s2;
        }
        __syncthreads();
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += red[NW * NJ * 32 + (w * NJ + jt) * 32 + (lane & 31)];
            const float rstd = 1.f / sqrtf(tot / (float)a.cout + 1e-5f);
            if (tt[jt] < a.T) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = v[jt][4 * g + e] * rstd * ga[g][e] + be[g][e];
                    *(f32x4*)(a.out + (size_t)b * a.out_bstride + (size_t)tt[jt] * a.out_C + cb + 8 * g) = o;
                }
            }
        }
    } else if constexpr (EPI == FR_PROJ_ZP) {
        static_assert(MI == 2, "paired (m, logs) tiles");
        const int pc = (ct0 / 2) * 32 + 4 * hl;  // first channel of this lane's groups
        f32x4 bm[4], bl[4];
        float nz[NJ][16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bm[g] = *(const f32x4*)(a.bias + pc + 8 * g);
            bl[g] = *(const f32x4*)(a.bias + a.H + pc + 8 * g);
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
                for (int e = 0; e < 4; ++e) nz[jt][4 * g + e] = a.noise[((size_t)b * a.H + pc + 8 * g + e) * a.T + tcl[jt]];
        }
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) {
            if (tt[jt] >= a.T) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float m = (acc[0][jt][4 * g + e] + bm[g][e]) * mk[jt];
                    const float lg = (acc[1][jt][4 * g + e] + bl[g][e]) * mk[jt];
                    o[e] = (m + expf(lg) * nz[jt][4 * g + e] * 0.66666f) * mk[jt];
                }
                *(f32x4*)(a.out + (size_t)b * a.out_bstride + (size_t)tt[jt] * a.out_C + pc + 8 * g) = o;
            }
        }
    } else if constexpr (EPI == FR_GATE) {
        static_assert(MI == 2, "paired (tanh, sigmoid) tiles");
        using o4 = __attribute__((ext_vector_type(4))) OpT;
        const int pc = (ct0 / 2) * 32 + 4 * hl;
        f32x4 bt[4], bs[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bt[g] = *(const f32x4*)(a.bias + pc + 8 * g);
            bs[g] = *(const f32x4*)(a.bias + a.H + pc + 8 * g);
            if (a.gc) {  // (uniform; the same sum the fused kernel stages in LDS: b_in + cond slice)
                bt[g] += *(const f32x4*)(a.gc + (size_t)b * a.gc_bstride + pc + 8 * g);
                bs[g] += *(const f32x4*)(a.gc + (size_t)b * a.gc_bstride + a.H + pc + 8 * g);
            }
        }
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) {
            if (tt[jt] >= a.T) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                o4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[e] = to_op<OpT>(fast_tanh(acc[0][jt][4 * g + e] + bt[g][e]) * fast_sigmoid(acc[1][jt][4 * g + e] + bs[g][e]));
                *(o4*)((OpT*)a.out_op + (size_t)b * a.out_op_bstride + (size_t)tt[jt] * a.H + pc + 8 * g) = o;
            }
        }
    } else if constexpr (EPI == FR_WN_RS || EPI == FR_WN_RS_LAST) {
        static_assert(MI == (EPI == FR_WN_RS ? 2 : 1), "paired (res, skip) tiles, or skip tiles alone on the last layer");
        const int pc = (EPI == FR_WN_RS ? ct0 / 2 : ct0) * 32 + 4 * hl;
        f32x4 br[MI][4], xv[NJ][4], sk[NJ][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) br[mi][g] = *(const f32x4*)(a.bias + mi * a.H + pc + 8 * g);
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) {
                const size_t o = (size_t)b * a.out_bstride + (size_t)tcl[jt] * a.out_C + pc + 8 * g;
                if constexpr (EPI == FR_WN_RS) xv[jt][g] = *(const f32x4*)(a.res + o);
                sk[jt][g] = *(const f32x4*)(a.skip + o);  // garbage on the first layer, not used there
            }
        }
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) {
            if (tt[jt] >= a.T) continue;
            const size_t o = (size_t)b * a.out_bstride + (size_t)tt[jt] * a.out_C + pc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if constexpr (EPI == FR_WN_RS) {
                    f32x4 r = {acc[0][jt][4 * g + 0], acc[0][jt][4 * g + 1], acc[0][jt][4 * g + 2], acc[0][jt][4 * g + 3]};
                    *(f32x4*)(a.out + o + 8 * g) = (xv[jt][g] + (r + br[0][g])) * mk[jt];
                }
                f32x4 sv = {acc[MI - 1][jt][4 * g + 0], acc[MI - 1][jt][4 * g + 1], acc[MI - 1][jt][4 * g + 2], acc[MI - 1][jt][4 * g + 3]};
                sv += br[MI - 1][g];
                if (!a.first) sv += sk[jt][g];
                *(f32x4*)(a.skip + o + 8 * g) = sv;
            }
        }
    } else {
        f32x4 bv[MI][4];
        int cog[MI][4];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                cog[mi][g] = (ct0 + mi) * 32 + 8 * g + 4 * hl;
                bv[mi][g] = *(const f32x4*)(a.bias + min(cog[mi][g], a.cout - 4));
            }
        f32x4 ex[MI][NJ][4];  // epilogue operand fetched up front: pitch embedding row / the x1 half of the flow stream
        if constexpr (EPI == FR_EMB || EPI == FR_COUPLE) {
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) {
                long long pi = 0;
                if constexpr (EPI == FR_EMB) pi = a.pitch ? a.pitch[(size_t)b * a.T + tcl[jt]] : 0;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int co = min(cog[mi][g], a.cout - 4);
                        if constexpr (EPI == FR_EMB)
                            ex[mi][jt][g] = a.pitch ? *(const f32x4*)(a.emb_pitch + (size_t)pi * a.cout + co) : f32x4{0.f, 0.f, 0.f, 0.f};
                        else
                            ex[mi][jt][g] = *(const f32x4*)(a.out + (size_t)b * a.out_bstride + (size_t)tcl[jt] * a.out_C + a.phys_base + co);
                    }
            }
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) {
                const int t = tt[jt];
                if (t >= a.T) continue;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = cog[mi][g];
                    if (co >= a.cout) continue;
                    f32x4 v = {acc[mi][jt][4 * g + 0], acc[mi][jt][4 * g + 1], acc[mi][jt][4 * g + 2], acc[mi][jt][4 * g + 3]};
                    v += bv[mi][g];
                    if constexpr (EPI == FR_EMB) {
                        v += ex[mi][jt][g];
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = lrelu(v[e] * a.scale, 0.1f) * mk[jt];
                        *(f32x4*)(a.out + (size_t)b * a.out_bstride + (size_t)t * a.out_C + co) = o;
                    } else if constexpr (EPI == FR_QKV) {
                        // q -> [T][H] (pre-scaled); k and v -> per (head, 32-key tile) blocks in MFMA FRAGMENT order, so
                        // that the attention kernel's operand loads are contiguous 1 KiB per instruction
                        using o4 = __attribute__((ext_vector_type(4))) OpT;
                        constexpr int DKc = 96, KSc = DKc / 16, DTc = DKc / 32;
                        const int nh = a.H / DKc, ntl = a.Tp / 32, tile = t >> 5, kk = t & 31;
                        if (co < a.H) {
                            o4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = to_op<OpT>(v[e] / a.qdiv);
                            *(o4*)((OpT*)a.out_op + (size_t)b * a.out_op_bstride + (size_t)t * a.H + co) = o;
                        } else if (co < 2 * a.H) {
                            const int c = co - a.H, head = c / DKc, cc = c - head * DKc;
                            o4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = to_op<OpT>(v[e]);
                            const size_t off = ((((size_t)b * nh + head) * ntl + tile) * KSc + cc / 16) * 512 +
                                               ((((cc & 15) >> 3) * 32 + kk) * 8) + (cc & 7);
                            *(o4*)((OpT*)a.kf + off) = o;
                        } else {
                            const int c = co - 2 * a.H, head = c / DKc, d = c - head * DKc;
                            const int s2 = kk >> 4, r = kk & 15, hv = (r >> 2) & 1, ev = (r & 3) + 4 * (r >> 3);
                            OpT* vp = (OpT*)a.vt + (((((size_t)b * nh + head) * ntl + tile) * DTc + d / 32) * 2 + s2) * 512 +
                                      (size_t)(hv * 32 + (d & 31)) * 8 + ev;
#pragma unroll
                            for (int e = 0; e < 4; ++e) vp[e * 8] = to_op<OpT>(v[e]);
                        }
                    } else if constexpr (EPI == FR_RELU_OP) {
                        using o4 = __attribute__((ext_vector_type(4))) OpT;
                        o4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = to_op<OpT>(fmaxf(v[e], 0.f) * mk[jt]);
                        *(o4*)((OpT*)a.out_op + (size_t)b * a.out_op_bstride + (size_t)t * a.cout + co) = o;
                    } else if constexpr (EPI == FR_F32_MASK) {
                        *(f32x4*)(a.out + (size_t)b * a.out_bstride + (size_t)t * a.out_C + co) = v * mk[jt];
                    } else if constexpr (EPI == FR_COUPLE) {
                        *(f32x4*)(a.out + (size_t)b * a.out_bstride + (size_t)t * a.out_C + a.phys_base + co) =
                            (ex[mi][jt][g] - v * mk[jt]) * mk[jt];
                    }
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------
// in_layer + gate of one WN layer with the TAPS split over the block's waves (round 6; the K-split idea of conv_ks_body, nsf_kernels.hpp).
// Measured first (ABAB, B = 1): splitting the layer's CHANNELS over 3x the blocks (FR_GATE epilogue of k_fr_conv + a res_skip launch)
// changes nothing -- 15.0 + 7.8 us against 22.3 us fused: a wave's K loop runs at the latency of ITS cold weight stream (60 k-steps with
// 16 in flight = four dependent trips to the far memory side), however few blocks share a CU.  Here block = (time tile, one (tanh, sigmoid)
// channel pair), wave w = tap w: its 12 k-steps are ALL requested by conv_prefetch before the tile is staged -- one trip --, the partial
// sums of waves 1.. meet in LDS and wave 0 adds them in the fixed order (((w0 + w1) + w2) + w3) + w4 and runs the gate epilogue.
// Not bit-identical to k_fr_wn (another summation order over the taps; option FR_WN_SPLIT pins the form).
// ------------------------------------------------------------------------------------------------
template <typename OpT, int H, int NJ>
static __global__ void __launch_bounds__(512) k_fr_gate_ks(FrConvArgs a) {
    using TL = Tile<H>;
    constexpr int STRIDE = TL::STRIDE, CC = TL::CC, TT = NJ * 32;
    static_assert(CC % KGROUP == 0 && CC / KGROUP <= FR_NB - 1, "one tap = at most NB - 1 k-groups: requested whole by conv_prefetch");
    using o4 = __attribute__((ext_vector_type(4))) OpT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.z, q0 = blockIdx.x * TT, pair = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, hl = lane >> 5;
    const int rows = TT + a.ntaps - 1 + 2;
    const OpT* wlane = (const OpT*)a.w + (size_t)(2 * pair) * a.ct_stride + (size_t)wave * CC * 512 + lane * 8;  // this wave's tap
    typename Op<OpT>::frag Aw[FR_NB][KGROUP][2];
    conv_prefetch<OpT, H, 2, KGROUP, FR_NB>(Aw, wlane, a.ct_stride, 1);
    // gate operands of wave 0, requested before the staging
    const int pc = pair * 32 + 4 * hl;
    f32x4 bt[4], bs[4];
    if (wave == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bt[g] = *(const f32x4*)(a.bias + pc + 8 * g);
            bs[g] = *(const f32x4*)(a.bias + a.H + pc + 8 * g);
            if (a.gc) {
                bt[g] += *(const f32x4*)(a.gc + (size_t)b * a.gc_bstride + pc + 8 * g);
                bs[g] += *(const f32x4*)(a.gc + (size_t)b * a.gc_bstride + a.H + pc + 8 * g);
            }
        }
    }
    // (blockDim = 64 * ntaps threads stage the tile; fr_stage's thread count is a template parameter: the shipped k = 5)
    if (blockDim.x == 320) fr_stage<OpT, H, 320>(smem, a.in, a.in_op, (long)b * a.in_bstride, a.T, q0 - a.pad, rows, a.T);
    else fr_stage_dyn<OpT, H>(smem, a.in, a.in_op, (long)b * a.in_bstride, a.T, q0 - a.pad, rows, a.T, (int)blockDim.x);
    __syncthreads();
    f32x16 acc[2][NJ];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][jt][e] = 0.f;
    const char* lds_lane = smem + (size_t)(lane & 31) * STRIDE + hl * 16;
    conv_run<OpT, H, 2, NJ, KGROUP, FR_NB>(acc, Aw, lds_lane, wlane, a.ct_stride, 1, wave, 1);
    float* red = (float*)(smem + (size_t)rows * STRIDE);  // [ntaps - 1][2][NJ][16][64]
    if (wave > 0) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
                for (int e = 0; e < 16; ++e) red[((((wave - 1) * 2 + mi) * NJ + jt) * 16 + e) * 64 + lane] = acc[mi][jt][e];
    }
    __syncthreads();
    if (wave != 0) return;
    for (int w = 1; w < a.ntaps; ++w)  // fixed order: ((w0 + w1) + w2) + ...
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mi][jt][e] += red[((((w - 1) * 2 + mi) * NJ + jt) * 16 + e) * 64 + lane];
#pragma unroll
    for (int jt = 0; jt < NJ; ++jt) {
        const int t = q0 + jt * 32 + (lane & 31);
        if (t >= a.T) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            o4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                o[e] = to_op<OpT>(fast_tanh(acc[0][jt][4 * g + e] + bt[g][e]) * fast_sigmoid(acc[1][jt][4 * g + e] + bs[g][e]));
            *(o4*)((OpT*)a.out_op + (size_t)b * a.out_op_bstride + (size_t)t * a.H + pc + 8 * g) = o;
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
    unsigned long long* stamps;  // dev only (RVCMI_FR_STAMPS): phase time stamps of block (1, 0)
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
    float* bg = (float*)(ACT + (size_t)(TT + 2) * STRIDE);  // [2H] in_layer bias + cond slice of this utterance
    const long boff = (long)b * a.bstride;
    const bool st = a.stamps && blockIdx.x == 1 && blockIdx.y == 0 && threadIdx.x == 0;
#define FR_STAMP(i) do { if (st) { a.stamps[2 * (i)] = wall_clock64(); a.stamps[2 * (i) + 1] = __builtin_amdgcn_s_memtime(); } } while (0)
    FR_STAMP(0);
    const OpT* wlane1 = (const OpT*)a.w_in + (size_t)(2 * wave) * a.ct_in + lane * 8;
    constexpr int NBW = NJ == 1 ? FR_NB : 2;  // 64-row tiles (large batches) hold twice the accumulators: a 2-deep ring avoids spills
    typename Op<OpT>::frag Aw[NBW][KGROUP][2];
    conv_prefetch<OpT, H, 2, KGROUP, NBW>(Aw, wlane1, a.ct_in, a.ntaps);
    // epilogue operands requested now, used ~10 us later: x (fp32 residual), running skip sum, res_skip bias
    constexpr int MI2 = LAST ? 1 : 2;
    const long long lenb = a.len ? a.len[b] : (long long)a.T + a.t_off;
    const int cb = wave * 32 + 4 * hl;
    f32x4 xv[NJ][4], sk[NJ][4], br[MI2][4];
    int tt[NJ];
#pragma unroll
    for (int jt = 0; jt < NJ; ++jt) {
        tt[jt] = q0 + jt * 32 + (lane & 31);
        const size_t o = boff + (size_t)min(tt[jt], a.T - 1) * H + cb;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            xv[jt][g] = *(const f32x4*)(a.x + o + 8 * g);
            sk[jt][g] = *(const f32x4*)(a.skip + o + 8 * g);  // garbage on the first layer, multiplied by 0 below
        }
    }
#pragma unroll
    for (int mi = 0; mi < MI2; ++mi)
#pragma unroll
        for (int g = 0; g < 4; ++g) br[mi][g] = *(const f32x4*)(a.b_rs + mi * H + cb + 8 * g);
    for (int i = threadIdx.x; i < 2 * H; i += NT) bg[i] = a.b_in[i] + (a.gc ? a.gc[(size_t)b * a.gc_bstride + i] : 0.f);
    fr_stage<OpT, H, NT>(smem, a.x, 0, boff, a.T, q0 - a.pad, xrows, a.T);
    // slack rows of the ACT tile (read one k-step ahead by the second K loop) must hold finite values
    for (int i = threadIdx.x; i < 2 * STRIDE / 4; i += NT) ((unsigned*)(ACT + (size_t)TT * STRIDE))[i] = 0u;
    __syncthreads();
    FR_STAMP(1);

    const char* lds_lane = smem + (size_t)(lane & 31) * STRIDE + hl * 16;
    const OpT* wlane = (const OpT*)a.w_rs + (size_t)(MI2 * wave) * a.ct_rs + lane * 8;
    typename Op<OpT>::frag Aw2[NBW][KGROUP][MI2];
    {
        f32x16 acc[2][NJ];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mi][jt][e] = 0.f;
        conv_run<OpT, H, 2, NJ, KGROUP, NBW>(acc, Aw, lds_lane, wlane1, a.ct_in, a.ntaps, 0, 1);
        FR_STAMP(2);
        conv_prefetch<OpT, H, MI2, KGROUP, NBW>(Aw2, wlane, a.ct_rs, 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 bt = *(const f32x4*)(bg + cb + 8 * g), bs = *(const f32x4*)(bg + H + cb + 8 * g);
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) {
                o4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[e] = to_op<OpT>(fast_tanh(acc[0][jt][4 * g + e] + bt[e]) * fast_sigmoid(acc[1][jt][4 * g + e] + bs[e]));
                *(o4*)(ACT + (size_t)(jt * 32 + (lane & 31)) * STRIDE + (cb + 8 * g) * 2) = o;
            }
        }
    }
    FR_STAMP(3);
    __syncthreads();
    FR_STAMP(4);

    f32x16 acc[MI2][NJ];
#pragma unroll
    for (int mi = 0; mi < MI2; ++mi)
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][jt][e] = 0.f;
    const char* act_lane = ACT + (size_t)(lane & 31) * STRIDE + hl * 16;
    conv_run<OpT, H, MI2, NJ, KGROUP, NBW>(acc, Aw2, act_lane, wlane, a.ct_rs, 1, 0, 1);
    FR_STAMP(5);
#pragma unroll
    for (int jt = 0; jt < NJ; ++jt) {
        if (tt[jt] >= a.T) continue;
        const float mk = (long long)(tt[jt] + a.t_off) < lenb ? 1.f : 0.f;
        const size_t o = boff + (size_t)tt[jt] * H + cb;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if constexpr (!LAST) {
                f32x4 r = {acc[0][jt][4 * g + 0], acc[0][jt][4 * g + 1], acc[0][jt][4 * g + 2], acc[0][jt][4 * g + 3]};
                *(f32x4*)(a.x_out + o + 8 * g) = (xv[jt][g] + (r + br[0][g])) * mk;
            }
            f32x4 s = {acc[MI2 - 1][jt][4 * g + 0], acc[MI2 - 1][jt][4 * g + 1], acc[MI2 - 1][jt][4 * g + 2], acc[MI2 - 1][jt][4 * g + 3]};
            s += br[MI2 - 1][g];
            if (!a.first) s += sk[jt][g];
            *(f32x4*)(a.skip + o + 8 * g) = s;
        }
    }
    FR_STAMP(6);
#undef FR_STAMP
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
    const void* q;      // OpT [B][T][H], pre-scaled by 1/sqrt(dk)
    const void* kf;     // OpT [B][heads][Tp/32][dk/16][64][8]
    const void* vf;     // OpT [B][heads][Tp/32][dk/32][2][64][8]
    void* out;          // OpT [B][T][H]
    const void* relk;   // packed E_k: [dk/16][64][8] OpT (A-fragment order, rows >= 2ws+1 zero)
    const float* relv;  // [2ws+1][dk] fp32
    const long long* len;
    int T, Tp, H, ws;
    unsigned long long* stamps;  // dev only
};

template <typename OpT, int DK, int NBAND>  // NBAND >= 2*ws + 1 (21 for the shipped window of 10)
static __global__ void __launch_bounds__(256) k_fr_attn(FrAttnArgs a) {
    using frag = typename Op<OpT>::frag;
    using o4 = __attribute__((ext_vector_type(4))) OpT;
    constexpr int KS = DK / 16;   // k-steps of the score product
    constexpr int DT = DK / 32;   // 32-row tiles of O^T
    constexpr int OS = DK + 1;    // LDS row stride of the merge buffers
    constexpr int DG = DK / 8;    // channels per thread in the output phase (256 threads = 32 queries x 8 groups)
    static_assert(DK % 32 == 0, "head dim must be a multiple of 32");
    __shared__ float Rl[32 * 33];
    __shared__ float Ml[4 * 32], Ll[4 * 32];
    __shared__ float Ol[4 * 32 * OS];
    __shared__ float Sb[32 * 32 + 32];  // masked scores of the relative band, Sb[q][j - q + ws] (+ 32 dump slots)
    __shared__ float Mf[32], Lf[32];
    __shared__ float Wl[4 * 32];       // merge weights exp(m_w - M) / L
    __shared__ __attribute__((aligned(16))) float Ev[32 * DK];  // relative value embeddings (2ws+1 <= 31 rows, rest zero)
    const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int q0 = qt * 32;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, hl = lane >> 5, ql = lane & 31;
    const int T = a.T;
    const int len = a.len ? (int)min((long long)T, a.len[b]) : T;
    const int nh = a.H / DK, ntl = a.Tp / 32;
    const OpT* KF = (const OpT*)a.kf + ((size_t)b * nh + h) * ntl * (KS * 512) + lane * 8;
    const OpT* VF = (const OpT*)a.vf + ((size_t)b * nh + h) * ntl * (DT * 2 * 512) + lane * 8;
    const bool st = a.stamps && blockIdx.x == 5 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0;
#define FR_STAMP(i) do { if (st) { a.stamps[2 * (i)] = wall_clock64(); a.stamps[2 * (i) + 1] = __builtin_amdgcn_s_memtime(); } } while (0)
    FR_STAMP(0);

    frag Bq[KS];
    {
        const OpT* qp = (const OpT*)a.q + ((size_t)b * T + min(q0 + ql, T - 1)) * a.H + h * DK + 8 * hl;
#pragma unroll
        for (int s = 0; s < KS; ++s) Bq[s] = *(const frag*)(qp + 16 * s);
    }
    // relative value table: requested now (unconditional, clamped), parked in registers, written to LDS after the key loop
    constexpr int EVN = 32 * DK / 256;
    float evr[EVN];
    {
        const int nev = (2 * a.ws + 1) * DK;
#pragma unroll
        for (int u = 0; u < EVN; ++u) {
            const int i = threadIdx.x + u * 256;
            const float v = a.relv[min(i, nev - 1)];
            evr[u] = i < nev ? v : 0.f;
        }
    }
    for (int i = threadIdx.x; i < 32 * 32; i += 256) Sb[i] = -INFINITY;
    if (wave == 0) {  // R[q][r] = q . E_k[r]
        f32x16 r = {0};
#pragma unroll
        for (int s = 0; s < KS; ++s) r = Op<OpT>::mfma(*(const frag*)((const OpT*)a.relk + (size_t)s * 512 + lane * 8), Bq[s], r);
#pragma unroll
        for (int i = 0; i < 16; ++i) Rl[ql * 33 + (i & 3) + 8 * (i >> 2) + 4 * hl] = r[i];
    }
    __syncthreads();
    FR_STAMP(1);

    const int q = q0 + ql;
    const bool qok = q < len;
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 O[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) O[d][i] = 0.f;
    const int nkt = (T + 31) / 32;
    // K / V fragments of the NEXT key tile are requested before the current tile is multiplied (register double buffer,
    // clamped tile index => unconditional loads): at B = 1 only ~80 blocks exist, so per-wave latency is what counts.
    struct KV {
        frag k[KS];
        frag v[DT][2];
    };
    auto load_tile = [&](int kt, KV& t) {
        const int kc = min(kt, nkt - 1);
#pragma unroll
        for (int s = 0; s < KS; ++s) t.k[s] = *(const frag*)(KF + ((size_t)kc * KS + s) * 512);
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) t.v[d][s2] = *(const frag*)(VF + (((size_t)kc * DT + d) * 2 + s2) * 512);
    };
    const int qrel = 4 * hl - q + a.ws;  // rel(i) = j0 + c_i + qrel with the compile-time row offset c_i of element i
    auto compute_tile = [&](int kt, const KV& t) {
        const int j0 = kt * 32;
        f32x16 S = {0};
#pragma unroll
        for (int s = 0; s < KS; ++s) S = Op<OpT>::mfma(t.k[s], Bq[s], S);
        // wave-uniform classification: interior tiles (no relative band, every key and query valid) skip all per-element
        // index work -- on wave64 every VALU instruction costs 4 cycles, and the general path below is ~35 of them per score
        const bool near = (j0 >= q0 - 32 - a.ws) && (j0 <= q0 + 32 + a.ws);
        // (query rows >= T of the last tile are never stored, so they do not force the general path when len == T)
        const bool plain = !near && j0 + 32 <= len && (q0 + 32 <= len || len == T);
        float mx = -INFINITY;
        if (plain) {
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, S[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ci = (i & 3) + 8 * (i >> 2);
                const int j = j0 + ci + 4 * hl;
                float s = S[i];
                const int rel = j0 + ci + qrel;
                const bool inband = near && rel >= 0 && rel <= 2 * a.ws;
                const float rv = Rl[ql * 33 + min(max(rel, 0), 2 * a.ws)];
                s += inband ? rv : 0.f;
                s = (qok && j < len) ? s : -1e4f;  // masked_fill(mask == 0, -1e4), attentions.py:115
                s = j < T ? s : -INFINITY;         // tile padding: not a key at all
                Sb[inband ? ql * 32 + rel : 32 * 32 + (lane & 31)] = s;  // band scores are needed again (relative values); else a dump slot
                S[i] = s;
                mx = fmaxf(mx, s);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        float ps = 0.f;
        frag Bp[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float p = __expf(S[i] - m_new);  // in [0, 1]: no fp16 clamp needed
            ps += p;
            Bp[i >> 3][i & 7] = (OpT)p;
        }
        if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {  // some query's running max moved: rescale
            const float sc = __expf(m_run - m_new);             // 0 on the first tile (m_run = -inf)
            l_run *= sc;
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int i = 0; i < 16; ++i) O[d][i] *= sc;
        }
        l_run += ps;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) O[d] = Op<OpT>::mfma(t.v[d][s2], Bp[s2], O[d]);
    };
    {
        // three tiles in flight (the K / V blocks come from another XCD's writes: ~2 us away); sched_barrier keeps the
        // compiler from sinking the prefetch loads down to their first use
        KV t0, t1, t2;
        int kt = wave;
        load_tile(kt, t0);
        load_tile(kt + 4, t1);
        while (kt < nkt) {
            load_tile(kt + 8, t2);
            __builtin_amdgcn_sched_barrier(0);
            compute_tile(kt, t0);
            kt += 4;
            if (kt >= nkt) break;
            load_tile(kt + 8, t0);
            __builtin_amdgcn_sched_barrier(0);
            compute_tile(kt, t1);
            kt += 4;
            if (kt >= nkt) break;
            load_tile(kt + 8, t1);
            __builtin_amdgcn_sched_barrier(0);
            compute_tile(kt, t2);
            kt += 4;
        }
    }
    FR_STAMP(2);
#pragma unroll
    for (int u = 0; u < EVN; ++u) Ev[threadIdx.x + u * 256] = evr[u];
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
#pragma unroll
        for (int w = 0; w < 4; ++w) Wl[w * 32 + x] = __expf(Ml[w * 32 + x] - M) / L;
    }
    __syncthreads();
    FR_STAMP(3);
    // output: thread = (query x, group of DG channels); the band probabilities of x stay in registers.  Every LDS read is
    // unconditional (rows >= 2ws+1 of Ev are zero and their p is 0).
    {
        const int x = threadIdx.x & 31, dg = threadIdx.x >> 5;
        const float M = Mf[x], Li = 1.f / Lf[x];
        float pb[NBAND];
#pragma unroll
        for (int r = 0; r < NBAND; ++r) pb[r] = __expf(Sb[x * 32 + r] - M) * Li;
        float w4[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) w4[w] = Wl[w * 32 + x];
        OpT* out = (OpT*)a.out + ((size_t)b * T + min(q0 + x, T - 1)) * a.H + h * DK + dg * DG;
#pragma unroll
        for (int c4 = 0; c4 < DG / 4; ++c4) {
            const int d = dg * DG + c4 * 4;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += Ol[(w * 32 + x) * OS + d + e] * w4[w];
#pragma unroll
            for (int r = 0; r < NBAND; ++r) acc += *(const f32x4*)(Ev + r * DK + d) * pb[r];  // relative values, attentions.py:127-135
            o4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = to_op<OpT>(acc[e]);
            if (q0 + x < T) *(o4*)(out + c4 * 4) = o;
        }
    }
    FR_STAMP(4);
#undef FR_STAMP
}

// ------------------------------------------------------------------------------------------------
// FFN (attentions.py:262-272) + residual + LayerNorm (encoders.py:78-80) in ONE launch:
//   conv_1 (k taps, H -> F) -> relu -> * mask -> LDS (fp16, F channels x 32*NJ1 rows) -> conv_2 (k taps, F -> H) -> * mask
//   -> + x -> LayerNorm.  The hidden activation never leaves the CU.  Tiles are "overlap-save" in time: a block computes
//   32*NJ1 hidden rows and from them 32*NJ1 - (k-1) output rows (6 % redundant conv_1 work at NJ1 = 1, k = 3).
// Reads x with a halo and writes another buffer (xo): in place would race with the neighbouring tiles' halo reads.
// ------------------------------------------------------------------------------------------------
struct FrFfnArgs {
    const float* x;     // [B][T][H] fp32
    float* xo;          // [B][T][H] fp32 (different buffer)
    long bstride;
    int T;
    const long long* len;
    const void* w1;     // packed conv_1
    long ct1;
    const float* b1;
    const void* w2;     // packed conv_2
    long ct2;
    const float* b2;
    int ntaps;          // kernel size (odd)
    const float* gamma;
    const float* beta;
    // split form (k_fr_ffn_part + k_fr_ffn_ln): conv_2 re-packed per slice of the hidden channels, fp32 partial sums [S][B][T][H]
    const void* w2s[4];
    long ct2s;
    float* part;
    int nsplit;
};

template <typename OpT, int H, int F, int NJ1>
static __global__ void __launch_bounds__(64 * (H / 32)) k_fr_ffn(FrFfnArgs a) {
    using TLH = Tile<H>;
    using TLF = Tile<F>;
    using o4 = __attribute__((ext_vector_type(4))) OpT;
    constexpr int NW = H / 32;
    constexpr int NT = 64 * NW;
    constexpr int HR = 32 * NJ1;                 // hidden rows per block
    constexpr int FT = F / 32 / NW;              // hidden-channel tiles per wave (4)
    static_assert(F % (32 * NW) == 0 && FT % 2 == 0, "hidden channels must split into pairs of tiles per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.y;
    const int pad = (a.ntaps - 1) / 2;
    const int TV = HR - 2 * pad;                 // valid output rows per block
    const int q0 = blockIdx.x * TV;              // first output row
    const int h0 = q0 - pad;                     // global row of hidden row 0
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, hl = lane >> 5;
    const int xrows = HR + a.ntaps - 1 + 2;
    char* XS = smem;                                          // x operand tile, rows h0 - pad ...
    char* HS = smem + (size_t)xrows * TLH::STRIDE;            // hidden tile [HR + ntaps - 1 + 2][F]
    const long boff = (long)b * a.bstride;
    const long long lenb = a.len ? a.len[b] : (long long)a.T;
    const int lenrow = (int)(lenb < (long long)a.T ? lenb : (long long)a.T);

    // conv_1 weights of this wave's first tile pair, then the x tile (x * x_mask: rows >= len are zero)
    const OpT* w1lane = (const OpT*)a.w1 + (size_t)(wave * FT) * a.ct1 + lane * 8;
    constexpr int NBF = NJ1 == 1 ? FR_NB : 2;  // see k_fr_wn
    typename Op<OpT>::frag Aw[NBF][KGROUP][2];
    conv_prefetch<OpT, H, 2, KGROUP, NBF>(Aw, w1lane, a.ct1, a.ntaps);
    // epilogue operands of the LayerNorm, requested early
    const int cb = wave * 32 + 4 * hl;
    int tt[NJ1], tcl[NJ1];
    f32x4 rv[NJ1][4], bv2[4], ga[4], be[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        bv2[g] = *(const f32x4*)(a.b2 + cb + 8 * g);
        ga[g] = *(const f32x4*)(a.gamma + cb + 8 * g);
        be[g] = *(const f32x4*)(a.beta + cb + 8 * g);
    }
#pragma unroll
    for (int jt = 0; jt < NJ1; ++jt) {
        tt[jt] = q0 + jt * 32 + (lane & 31);
        tcl[jt] = min(tt[jt], a.T - 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) rv[jt][g] = *(const f32x4*)(a.x + boff + (size_t)tcl[jt] * H + cb + 8 * g);
    }
    fr_stage<OpT, H, NT>(XS, a.x, 0, boff, a.T, h0 - pad, xrows, lenrow);
    // rows of the hidden tile past HR (read by conv_2's look-ahead and by the discarded last output rows) must be finite
    for (int i = threadIdx.x; i < (a.ntaps - 1 + 2) * (TLF::STRIDE / 4); i += NT) ((unsigned*)(HS + (size_t)HR * TLF::STRIDE))[i] = 0u;
    __syncthreads();

    // ---- conv_1 + relu + mask -> hidden tile, two output-channel tiles at a time ----
    const char* x_lane = XS + (size_t)(lane & 31) * TLH::STRIDE + hl * 16;
#pragma unroll 1
    for (int pass = 0; pass < FT / 2; ++pass) {
        const int ct = wave * FT + 2 * pass;
        const OpT* wl = (const OpT*)a.w1 + (size_t)ct * a.ct1 + lane * 8;
        if (pass > 0) conv_prefetch<OpT, H, 2, KGROUP, NBF>(Aw, wl, a.ct1, a.ntaps);
        f32x16 acc[2][NJ1];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int jt = 0; jt < NJ1; ++jt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mi][jt][e] = 0.f;
        conv_run<OpT, H, 2, NJ1, KGROUP, NBF>(acc, Aw, x_lane, wl, a.ct1, a.ntaps, 0, 1);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = (ct + mi) * 32 + 8 * g + 4 * hl;
                const f32x4 bb = *(const f32x4*)(a.b1 + c);
#pragma unroll
                for (int jt = 0; jt < NJ1; ++jt) {
                    const int hr = jt * 32 + (lane & 31);
                    const int t = h0 + hr;
                    const float mk = (t >= 0 && t < lenrow) ? 1.f : 0.f;  // zero padding of conv_2 and x_mask in one
                    o4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = to_op<OpT>(fmaxf(acc[mi][jt][4 * g + e] + bb[e], 0.f) * mk);
                    *(o4*)(HS + (size_t)hr * TLF::STRIDE + c * 2) = o;
                }
            }
    }
    // ---- conv_2 over the hidden tile ----
    const OpT* w2lane = (const OpT*)a.w2 + (size_t)wave * a.ct2 + lane * 8;
    typename Op<OpT>::frag Aw2[NBF][KGROUP][1];
    conv_prefetch<OpT, F, 1, KGROUP, NBF>(Aw2, w2lane, a.ct2, a.ntaps);
    __syncthreads();
    f32x16 acc2[1][NJ1];
#pragma unroll
    for (int jt = 0; jt < NJ1; ++jt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[0][jt][e] = 0.f;
    const char* h_lane = HS + (size_t)(lane & 31) * TLF::STRIDE + hl * 16;
    conv_run<OpT, F, 1, NJ1, KGROUP, NBF>(acc2, Aw2, h_lane, w2lane, a.ct2, a.ntaps, 0, 1);

    // ---- (* mask) + x -> LayerNorm over the H channels (statistics through LDS, two passes) ----
    __syncthreads();  // x tile is dead
    float* red = (float*)XS;  // [2][NW][HR]
    float v[NJ1][16];
#pragma unroll
    for (int jt = 0; jt < NJ1; ++jt) {
        const float pm = (long long)tcl[jt] < lenb ? 1.f : 0.f;
        float sum = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[jt][4 * g + e] = rv[jt][g][e] + (acc2[0][jt][4 * g + e] + bv2[g][e]) * pm;
                sum += v[jt][4 * g + e];
            }
        sum += __shfl_xor(sum, 32, 64);
        if (hl == 0) red[(wave * NJ1 + jt) * 32 + (lane & 31)] = sum;
    }
    __syncthreads();
#pragma unroll
    for (int jt = 0; jt < NJ1; ++jt) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) tot += red[(w * NJ1 + jt) * 32 + (lane & 31)];
        const float mean = tot / (float)H;
        float s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            v[jt][e] -= mean;
            s2 += v[jt][e] * v[jt][e];
        }
        s2 += __shfl_xor(s2, 32, 64);
        if (hl == 0) red[NW * NJ1 * 32 + (wave * NJ1 + jt) * 32 + (lane & 31)] = s2;
    }
    __syncthreads();
#pragma unroll
    for (int jt = 0; jt < NJ1; ++jt) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) tot += red[NW * NJ1 * 32 + (w * NJ1 + jt) * 32 + (lane & 31)];
        const float rstd = 1.f / sqrtf(tot / (float)H + 1e-5f);
        const int j = jt * 32 + (lane & 31);
        if (j < TV && tt[jt] < a.T) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = v[jt][4 * g + e] * rstd * ga[g][e] + be[g][e];
                *(f32x4*)(a.xo + boff + (size_t)tt[jt] * H + cb + 8 * g) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The FFN for SMALL grids (one clip: 38 time tiles on 256 CUs).  In k_fr_ffn every block streams ALL of a layer's FFN weights
// (1.77 MB at H = 192, F = 768, k = 3) through one CU's L1 at 64 B/clk: 27.6k cycles before anything else, 43 us per layer for
// 2.1 GFLOP.  Here grid.z = S slices of the hidden channels: block (tile, b, s) computes ITS F / S hidden channels (conv_1 + relu
// + mask -> LDS), multiplies them with its slice of conv_2 (re-packed per slice on the host) and writes the fp32 partial sums
// [rows][H]; k_fr_ffn_ln adds the S partials in slice order, bias, mask, residual and applies the LayerNorm.  4x the blocks, a
// quarter of the weight stream each.  (The partial sums add in a different order than one long K loop: last-bit differences.)
// ------------------------------------------------------------------------------------------------
template <typename OpT, int H, int FS, int NJ1>
static __global__ void __launch_bounds__(64 * (H / 32)) k_fr_ffn_part(FrFfnArgs a) {
    using TLH = Tile<H>;
    using TLF = Tile<FS>;
    using o4 = __attribute__((ext_vector_type(4))) OpT;
    constexpr int NW = H / 32;
    constexpr int NT = 64 * NW;
    constexpr int HR = 32 * NJ1;
    static_assert(FS == 32 * NW, "one hidden-channel tile per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.y, sl = blockIdx.z;
    const int pad = (a.ntaps - 1) / 2;
    const int TV = HR - 2 * pad;
    const int q0 = blockIdx.x * TV;
    const int h0 = q0 - pad;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, hl = lane >> 5;
    const int xrows = HR + a.ntaps - 1 + 2;
    char* XS = smem;
    char* HS = smem + (size_t)xrows * TLH::STRIDE;  // hidden tile [HR + ntaps - 1 + 2][FS]
    const long boff = (long)b * a.bstride;
    const long long lenb = a.len ? a.len[b] : (long long)a.T;
    const int lenrow = (int)(lenb < (long long)a.T ? lenb : (long long)a.T);

    const OpT* w1lane = (const OpT*)a.w1 + (size_t)(sl * NW + wave) * a.ct1 + lane * 8;
    constexpr int NBF = NJ1 == 1 ? FR_NB : 2;
    typename Op<OpT>::frag Aw[NBF][KGROUP][1];
    conv_prefetch<OpT, H, 1, KGROUP, NBF>(Aw, w1lane, a.ct1, a.ntaps);
    fr_stage<OpT, H, NT>(XS, a.x, 0, boff, a.T, h0 - pad, xrows, lenrow);
    for (int i = threadIdx.x; i < (a.ntaps - 1 + 2) * (TLF::STRIDE / 4); i += NT) ((unsigned*)(HS + (size_t)HR * TLF::STRIDE))[i] = 0u;
    __syncthreads();

    // ---- conv_1 (this slice's hidden channels) + relu + mask -> hidden tile ----
    const char* x_lane = XS + (size_t)(lane & 31) * TLH::STRIDE + hl * 16;
    {
        f32x16 acc[1][NJ1];
#pragma unroll
        for (int jt = 0; jt < NJ1; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[0][jt][e] = 0.f;
        conv_run<OpT, H, 1, NJ1, KGROUP, NBF>(acc, Aw, x_lane, w1lane, a.ct1, a.ntaps, 0, 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cl = wave * 32 + 8 * g + 4 * hl;  // channel inside the slice
            const f32x4 bb = *(const f32x4*)(a.b1 + sl * FS + cl);
#pragma unroll
            for (int jt = 0; jt < NJ1; ++jt) {
                const int hr = jt * 32 + (lane & 31);
                const int t = h0 + hr;
                const float mk = (t >= 0 && t < lenrow) ? 1.f : 0.f;
                o4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = to_op<OpT>(fmaxf(acc[0][jt][4 * g + e] + bb[e], 0.f) * mk);
                *(o4*)(HS + (size_t)hr * TLF::STRIDE + cl * 2) = o;
            }
        }
    }
    // ---- this slice's part of conv_2 ----
    const OpT* w2lane = (const OpT*)a.w2s[sl] + (size_t)wave * a.ct2s + lane * 8;
    typename Op<OpT>::frag Aw2[NBF][KGROUP][1];
    conv_prefetch<OpT, FS, 1, KGROUP, NBF>(Aw2, w2lane, a.ct2s, a.ntaps);
    __syncthreads();
    f32x16 acc2[1][NJ1];
#pragma unroll
    for (int jt = 0; jt < NJ1; ++jt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[0][jt][e] = 0.f;
    const char* h_lane = HS + (size_t)(lane & 31) * TLF::STRIDE + hl * 16;
    conv_run<OpT, FS, 1, NJ1, KGROUP, NBF>(acc2, Aw2, h_lane, w2lane, a.ct2s, a.ntaps, 0, 1);
    const int cb = wave * 32 + 4 * hl;
    float* pp = a.part + ((size_t)sl * gridDim.y + b) * (size_t)a.T * H;
#pragma unroll
    for (int jt = 0; jt < NJ1; ++jt) {
        const int j = jt * 32 + (lane & 31);
        const int t = q0 + j;
        if (j < TV && t < a.T) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 o = {acc2[0][jt][4 * g + 0], acc2[0][jt][4 * g + 1], acc2[0][jt][4 * g + 2], acc2[0][jt][4 * g + 3]};
                *(f32x4*)(pp + (size_t)t * H + cb + 8 * g) = o;
            }
        }
    }
}

// y = LayerNorm(x + (sum_s part[s] + b2) * mask)   (encoders.py:78-80); one wave per row, H = 64 * CPL channels
template <int H>
static __global__ void __launch_bounds__(256) k_fr_ffn_ln(const float* __restrict__ part, int nsplit, const float* __restrict__ x, float* __restrict__ xo,
                                                          const float* __restrict__ b2, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const long long* __restrict__ len, int T, int B) {
    constexpr int CPL = H / 64;
    static_assert(H % 64 == 0, "channels per lane");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= (long)B * T) return;
    const int b = (int)(row / T), t = (int)(row - (long)b * T);
    const float pm = (len ? (long long)t < len[b] : true) ? 1.f : 0.f;
    float v[CPL];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int c = lane + 64 * i;
        float p = part[(size_t)row * H + c];
        for (int s = 1; s < nsplit; ++s) p += part[((size_t)s * B * T + row) * H + c];
        v[i] = x[(size_t)row * H + c] + (p + b2[c]) * pm;
        sum += v[i];
    }
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
    const float mean = sum / (float)H;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        v[i] -= mean;
        s2 += v[i] * v[i];
    }
    for (int off = 32; off >= 1; off >>= 1) s2 += __shfl_xor(s2, off, 64);
    const float rstd = 1.f / sqrtf(s2 / (float)H + 1e-5f);
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int c = lane + 64 * i;
        xo[(size_t)row * H + c] = v[i] * rstd * gamma[c] + beta[c];
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
