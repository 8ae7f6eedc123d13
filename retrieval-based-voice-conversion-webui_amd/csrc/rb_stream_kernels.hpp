// Streaming fused ResBlock1 (rvc/layers/residuals.py:68-85) for gfx950: a persistent block walks a STRIP of consecutive
// time rows of one (utterance, resblock) and keeps every halo on the CU, so nothing is re-read or re-computed:
//
//   * the block advances R = 32*NJ rows per step; per (conv1, conv2) pair m it holds in LDS the last Hx_m = 32 + p1 - p2
//     activated input rows ("X history") and the last 2*p2 activated h rows ("H history") of the previous step;
//   * each pair's output window lags its input window by exactly 32 rows (>= p1 + p2, the rows a pair cannot produce
//     yet).  32 rows = one MFMA column tile, so the fp32 residual  x' = conv2(h) + x  needs no data movement: output tile
//     jt adds input tile jt-1, which the SAME lane of the same wave already holds in registers (tile NJ-1 is carried to
//     the next step).  The fp32 residual stream therefore never leaves the register file between the pairs;
//   * with ND = 3 all pairs of a resblock run back to back on the resident rows: ONE read and ONE write of the stream
//     per resblock, no overlap-save recomputation (k_rb_full computes R rows to keep R - 2*HL).  With ND = 1 the same
//     code is the pair-level kernel for channel counts whose three histories do not fit (C = 256).
//
// One wave per SIMD (512 registers): waves split OUTPUT CHANNELS only (NCO slices of 32*MI), all on the same time
// window, so the residual tile shift never crosses a wave.  LDS: one operand tile M of HEAD + R rows used alternately as
// X (new rows at HEAD) and H (new rows at HROW), the per-pair history side buffers, and the biases.
//
// The schedule (row arithmetic, masks, history copies, tile shift) is modelled line by line in tools/model_rb_stream.py
// and checked there against a direct evaluation of the resblock.
#pragma once
#include "nsf_kernels.hpp"

namespace rvcmi {

constexpr int RS_HEAD = 62;   // first M row of the new X rows: 2*p2 (<= 10) rows of H head + Hx (<= 52) rows of X head in front
constexpr int RS_HROW = 10;   // first M row of the new H rows
constexpr int RS_SLACK = 4;   // zero rows behind the tile: the last padded tap of conv1 may read p1 + p2 + dil - 32 <= 3 rows beyond
constexpr int RS_MAXND = 3;

struct RbStreamJob {
    const float* src;
    float* dst;
    const void* w1[RS_MAXND];
    const void* w2[RS_MAXND];
    const float* b1[RS_MAXND];
    const float* b2[RS_MAXND];
    long ct1, ct2;       // packed elements per 32-channel output tile
    int k, k_p;          // real / padded taps
    int dil[RS_MAXND];
    int blk0;            // first blockIdx.x of this job
    int nstrips;
    int strip_len;       // output rows per strip
    int sx_off[RS_MAXND];  // LDS row offset of pair m's X history inside the side area
    int sh_off[RS_MAXND];  // ... and of its H history
};
struct RbStreamArgs {
    RbStreamJob job[3];
    int njobs;
    int B;          // utterances; the grid is (sum of nstrips) * B blocks
    int L;
    const int* lens;  // ragged batch (nsf_kernels.hpp item_rows)
    int lmul;
    long bstride;
    int side_rows;  // rows of the side area (max over jobs)
    int flags;      // bit 2 (4) = lean K loop instantiation; bit 3 (8) = the dst streams are fp16 (pack4_h, nsf_kernels.hpp), same
                    // element layout; bit 4 (16) = the src stream is fp16 (KL = 2 only)
    unsigned long long* ts;  // dev only (RVCMI_RS_STAMPS=1): per-wave cycle sums per phase, [block][wave][16]
};

// cooperative LDS -> LDS copy of `rows` operand rows (whole 16-byte words, pad included)
// (a batched variant -- 4 reads in flight, then 4 predicated writes -- measured SLOWER: phase A 3.5k -> 5.7k cycles)
template <int STRIDE, int NT>
__device__ __forceinline__ void rs_copy_rows(char* dst, const char* src, int rows) {
    constexpr int W = STRIDE / 16;
    const int total = rows * W;
    for (int i = threadIdx.x; i < total; i += NT) *(uint4*)(dst + (size_t)i * 16) = *(const uint4*)(src + (size_t)i * 16);
}

// KL = 2: the lean K loop kconv (nsf_kernels.hpp) instead of conv_prefetch / conv_run
// XH = 1 (with KL = 2): the src stream is fp16 (a.flags bit 4) -- its own instantiation: as a runtime switch the fp32 and the fp16 row
// registers of the step prefetch were both live (472 -> 512 registers + 16 spilled)
template <typename OpT, int C, int MI, int NJ, int NCO, int ND, int KG, int NB, int OCC = 1, int KL = 1, int XH = 0>
static __global__ void __launch_bounds__(64 * NCO, OCC) k_rb_stream(RbStreamArgs a) {
    using TL = Tile<C>;
    using frag = typename Op<OpT>::frag;
    using o4 = __attribute__((ext_vector_type(4))) OpT;
    constexpr int STRIDE = TL::STRIDE;
    constexpr int NT = 64 * NCO;
    constexpr int R = 32 * NJ;
    constexpr int CP = 32 * MI * NCO;
    static_assert(CP == C, "waves must tile the channels exactly");
    constexpr int MROWS = RS_HEAD + R + RS_SLACK;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // KL = 1: [M | side | dump row | biases].  KL = 2: [side | dump row | biases | M ...]: the fp32 transposition tile T of the
    // coalesced step IO (R rows x (4 C + 16) bytes = 101 KB at C = 128) starts at M and runs over M's end into the unused LDS.
    char* side = KL == 2 ? smem : smem + (size_t)MROWS * STRIDE;
    // after the side area: one dump row (discarded history writes), then the biases [ND][2][CP]
    char* dump = side + (size_t)a.side_rows * STRIDE;
    float* bias_l = (float*)(dump + STRIDE);
    char* M = KL == 2 ? (char*)(bias_l + ND * 2 * CP) : smem;

    // ---- which job / utterance / strip ---------------------------------------------------------------------------------
    // The launch is one 1-D grid of (strips of all jobs) x B blocks.  Hardware places block i on XCD i % 8; the logical
    // order is JOB-major and each XCD takes one contiguous chunk of it, so an XCD's 4 MB L2 mostly serves ONE resblock's
    // weights (2.2 MB at k = 11) instead of all three (4.1 MB: with the round-robin order the weight loads missed L2 and the
    // kernel fetched 2.4x its activation bytes).  Placement only affects speed, never results.
    int lb;
    {
        const int nb = (int)gridDim.x, q = nb >> 3, r = nb & 7, xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
        lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    int ji = 0;
#pragma unroll
    for (int j = 1; j < 3; ++j)
        if (j < a.njobs && lb >= a.job[j].blk0 * a.B) ji = j;
    const RbStreamJob& J = a.job[ji];
    const int rem = lb - J.blk0 * a.B;
    const int b = rem / J.nstrips;
    const int strip = rem - b * J.nstrips;
    if (b >= a.B) return;
    const float* src = J.src + (size_t)b * a.bstride;
    float* dst = J.dst + (size_t)b * a.bstride;
    const int L = item_rows(a.lens, b, a.lmul, a.L);
    const int S0 = strip * J.strip_len;
    const int S1 = min(L, S0 + J.strip_len);
    if (S0 >= L) return;  // (ragged batch) a strip behind the item's end: block-uniform, before any barrier
    const int p2 = (J.k - 1) / 2;
    int HL = ND * p2;
#pragma unroll
    for (int m = 0; m < ND; ++m) HL += J.dil[m] * (J.k - 1) / 2;
    const int r0 = S0 - HL;                                       // first row loaded by step 0
    const int nsteps = (S1 - r0 + 32 * ND + R - 1) / R;

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;  // wave: scalar (SGPR weight bases)
    const int ct0 = wave * MI;               // first 32-channel output tile of this wave
    const int half4 = 4 * (lane >> 5);
    const int lrow = lane & 31;

    // ---- zero the whole LDS image once (histories start empty, slack rows stay zero), stage the biases ----------------
    {
        const int total16 = (int)(((size_t)(MROWS + a.side_rows + 1) * STRIDE + (KL == 2 ? ND * 2 * CP * 4 : 0)) / 16);
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (int i = threadIdx.x; i < total16; i += NT) *(uint4*)(smem + (size_t)i * 16) = z;
        if constexpr (KL == 2) __syncthreads();
        for (int i = threadIdx.x; i < ND * 2 * CP; i += NT) {
            const int m = i / (2 * CP), w = (i / CP) & 1, c = i % CP;
            bias_l[i] = (w ? J.b2[m] : J.b1[m])[c];
        }
    }

    // dev-only phase timing: wave-uniform s_memtime deltas summed over all steps / pairs (scalar registers)
    unsigned long long tprev = 0, tsum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const bool stamps = a.ts != nullptr;
    auto stamp = [&](int ph) {
        if (stamps) {
            const unsigned long long t = __builtin_readcyclecounter();
            tsum[ph] += t - tprev;
            tprev = t;
        }
    };
    if (stamps) tprev = __builtin_readcyclecounter();

    auto bar = [] {  // KL = 2: a barrier that does not drain the prefetched weight fragments (lds_barrier, below)
        if constexpr (KL == 2) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        } else {
            __syncthreads();
        }
    };
    f32x16 carry[ND][MI];
#pragma unroll
    for (int m = 0; m < ND; ++m)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) carry[m][mi][e] = 0.f;

    frag A[NB][KG][MI];  // weight register ring, requested one phase ahead of its use
    frag A8[8][MI];      // ... of the lean loop (KL = 2)
    const unsigned loff = (unsigned)lane * 16u;
    const unsigned ctb1 = (unsigned)(J.ct1 * 2), ctb2 = (unsigned)(J.ct2 * 2);
    auto prefetch = [&](const OpT* wl, long ct, unsigned ctb) {  // wl: this wave's first output-channel tile (wave-uniform)
        if constexpr (KL == 2) kconv_prefetch<OpT, MI>(A8, weight_rsrc(wl), loff, ctb);
        else conv_prefetch<OpT, C, MI, KG, NB>(A, wl + lane * 8, ct, J.k_p);
    };
    auto run = [&](f32x16 (&acc)[MI][NJ], const char* lds_lane, const OpT* wl, long ct, unsigned ctb, int dstep) {
        if constexpr (KL == 2) kconv<OpT, C, MI, NJ, STRIDE>(acc, A8, lds_address(lds_lane), weight_rsrc(wl), loff, ctb, J.k, dstep);
        else conv_run<OpT, C, MI, NJ, KG, NB>(acc, A, lds_lane, wl + lane * 8, ct, J.k_p, 0, dstep);
    };
    prefetch((const OpT*)J.w1[0] + (size_t)ct0 * J.ct1, J.ct1, ctb1);
    __syncthreads();

    // ---- coalesced step IO (KL = 2) ---------------------------------------------------------------------------------------
    // The D-layout global accesses of KL = 1 (every instruction touches 32 rows x 32 bytes) are TA-issue-bound: 96 loads +
    // 96 stores = 10.6k cycles per step, the largest non-MFMA item left.  Here a step's R rows move as whole rows -- 1 KB
    // contiguous per wave instruction, 24 per wave each way -- and change layout through an fp32 tile T in LDS (row stride
    // 4 C + 16 bytes: conflict-free for both the row-wise and the D-layout 16-byte accesses).  The next step's rows are
    // requested BEFORE the finished rows are stored, so their HBM latency runs under the store phase.
    // One instruction per 16-byte chunk in every IO loop (a lone wave issues ~4.5 cycles per instruction: the first version, with
    // per-chunk address arithmetic, row clamps and store predicates, cost as much as the D-layout accesses it replaced):
    //   * global rows go through RAW BUFFER descriptors whose range check does the masking: the load descriptor covers the
    //     utterance (rows outside [0, L) read as zeros), the store descriptor covers THIS STRIP (rows outside [S0, S1) are dropped);
    //     the row offset lives in the VGPR offset (the range check covers VGPR + immediate offset);
    //   * thread (g8, c) = (tid / 32, tid % 32) owns chunk c of the 24 rows 24 g8 .. 24 g8 + 23: chunk `it` is an immediate offset
    //     (it % 8) * 512 on one of three offset registers in global memory and it * TSTR on one base register in LDS.
    constexpr int CHR = C / 4;            // 16-byte chunks per fp32 row
    constexpr int NG8 = NT / CHR;         // thread groups (8)
    constexpr int NCH = R / NG8;          // rows (= chunks) per thread per step (24)
    constexpr int TSTR = C * 4 + 16;
    static_assert(KL != 2 || (NT % CHR == 0 && R % NG8 == 0 && NCH % 8 == 0 && C * 4 * 8 == 4096), "IO chunk geometry");
    char* T = M;
    f32x4 xr[(KL == 2 && !XH) ? NCH : 1];
    u32x4_t xrh[(KL == 2 && XH) ? R / (NT / (CHR / 2)) : 1];  // fp16 rows: NCHH chunks per thread
    constexpr bool xhf = XH != 0;  // fp16 input stream (half the bytes, half the load instructions and LDS round trip)
    static_assert(!xhf || KL == 2, "fp16 input rows come through the coalesced step IO");
    const __amdgpu_buffer_rsrc_t rs_src =
        xhf ? __builtin_amdgcn_make_buffer_rsrc((void*)((const _Float16*)J.src + (size_t)b * a.bstride), 0, (int)((size_t)L * C * 2), 0x00020000)
            : __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)((size_t)L * C * 4), 0x00020000);
    const bool yh = (a.flags & 8) != 0;  // block-uniform: fp16 output stream
    _Float16* dsth = (_Float16*)J.dst + (size_t)b * a.bstride;
    const __amdgpu_buffer_rsrc_t rs_dst =
        yh ? __builtin_amdgcn_make_buffer_rsrc((void*)(dsth + (size_t)S0 * C), 0, (int)((size_t)max(S1 - S0, 0) * C * 2), 0x00020000)
           : __builtin_amdgcn_make_buffer_rsrc((void*)(dst + (size_t)S0 * C), 0, (int)((size_t)max(S1 - S0, 0) * C * 4), 0x00020000);
    // fp16 rows: 2 C bytes = CHR / 2 chunks; thread (g16, c) = (tid / (CHR / 2), tid % (CHR / 2)) owns chunk c of NCH / 2 rows
    constexpr int CHRH = CHR / 2, NG16 = NT / CHRH, NCHH = R / NG16, TSTRH = C * 2 + 16;
    static_assert(KL != 2 || (R % NG16 == 0 && (NCHH - 1) * C * 2 < 4096), "fp16 IO chunk geometry");
    const int g16 = (int)threadIdx.x / CHRH, cchh = (int)threadIdx.x % CHRH;
    const unsigned ioh_voff = (unsigned)(g16 * NCHH * C * 2 + cchh * 16);
    const unsigned ioh_lds = lds_address(T) + (unsigned)(g16 * NCHH * TSTRH + cchh * 16);
    const unsigned dlh_lds = lds_address(T) + (unsigned)(lrow * TSTRH + (ct0 * 32 + half4) * 2);
    const int g8 = (int)threadIdx.x / CHR, cch = (int)threadIdx.x % CHR;
    const unsigned io_voff = (unsigned)(g8 * NCH * C * 4 + cch * 16);           // this thread's chunk of its first row, in bytes
    const unsigned io_lds = lds_address(T) + (unsigned)(g8 * NCH * TSTR + cch * 16);
    auto issue_loads = [&](int w0) {
        if constexpr (xhf) {  // fp16 rows (2 C bytes): thread (g16, c) owns chunk c of NCHH rows, immediates it * 2 C < 4096 on one offset register
            unsigned v0 = ioh_voff + (unsigned)(w0 * (C * 2));  // (wraps for w0 < 0: far outside the descriptor => zeros)
            asm volatile("" : "+v"(v0));
#pragma unroll
            for (int it = 0; it < NCHH; ++it) xrh[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_src, v0 + (unsigned)(it * (C * 2)), 0, 0);
        } else {
        // (wraps for w0 < 0: far outside the descriptor => zeros.  Three opaque offset registers, 4096 bytes apart: the rest of a
        //  chunk's offset fits the instruction's 12-bit immediate; left transparent, the compiler hoists 24 per-chunk registers)
        unsigned v0[NCH / 8];
#pragma unroll
        for (int j = 0; j < NCH / 8; ++j) {
            v0[j] = io_voff + (unsigned)(w0 * (C * 4)) + (unsigned)(j * 4096);
            asm volatile("" : "+v"(v0[j]));
        }
#pragma unroll
        for (int it = 0; it < NCH; ++it)
            xr[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_src, v0[it / 8] + (unsigned)((it % 8) * (C * 4)), 0, 0));
        }
    };
    // D-layout addresses of this lane inside T: tiles 0..2 from one base register, 3..5 from a second (16-bit offset fields)
    const unsigned dl_lds = lds_address(T) + (unsigned)(lrow * TSTR + (ct0 * 32 + half4) * 4);
    using lds_f32x4 = __attribute__((address_space(3))) f32x4;
    if constexpr (KL == 2) issue_loads(r0);

    for (int step = 0; step < nsteps; ++step) {
        f32x16 xin[MI][NJ];
        if constexpr (KL == 2) {
            bar();  // every thread is done with T (the previous step's row-wise reads for its stores)
            if constexpr (xhf) {
                using lds_u4i = __attribute__((address_space(3))) u32x4_t;
                using h4 = __attribute__((ext_vector_type(4))) _Float16;
                using lds_h4 = __attribute__((address_space(3))) h4;
                {
                    unsigned bb = ioh_lds;
                    asm volatile("" : "+v"(bb));
#pragma unroll
                    for (int it = 0; it < NCHH; ++it) *(lds_u4i*)(size_t)(bb + (unsigned)(it * TSTRH)) = xrh[it];
                }
                bar();
                {
                    unsigned b0 = dlh_lds;
                    asm volatile("" : "+v"(b0));
#pragma unroll
                    for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const h4 v = *(const lds_h4*)(size_t)(b0 + (unsigned)(jt * 32 * TSTRH + (mi * 32 + 8 * g) * 2));
#pragma unroll
                                for (int e = 0; e < 4; ++e) xin[mi][jt][4 * g + e] = (float)v[e];
                            }
                }
            } else {
            {
                unsigned b = io_lds;
                asm volatile("" : "+v"(b));
#pragma unroll
                for (int it = 0; it < NCH; ++it) *(lds_f32x4*)(size_t)(b + (unsigned)(it * TSTR)) = xr[it];
            }
            bar();
            {
                unsigned b0 = dl_lds, b1 = dl_lds + (unsigned)(3 * 32 * TSTR);
                asm volatile("" : "+v"(b0), "+v"(b1));
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 v = *(const lds_f32x4*)(size_t)((jt < 3 ? b0 : b1) + (unsigned)((jt % 3) * 32 * TSTR + (mi * 32 + 8 * g) * 4));
#pragma unroll
                            for (int e = 0; e < 4; ++e) xin[mi][jt][4 * g + e] = v[e];
                        }
            }
            }
            bar();  // before phase A publishes into M (the same LDS)
        } else
        // ---- load the next R rows of x straight into the accumulator layout (clamped addresses, masked values) -------
        {
            const int w0 = r0 + step * R;
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) {
                const int tg = w0 + jt * 32 + lrow;
                const unsigned msk = (tg >= 0 && tg < L) ? 0xffffffffu : 0u;
                const int tgc = min(max(tg, 0), L - 1);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 v = *(const f32x4*)(src + (size_t)tgc * C + (ct0 + mi) * 32 + 8 * g + half4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) xin[mi][jt][4 * g + e] = mask_bits(v[e], msk);
                    }
            }
        }
        stamp(8);  // x loads issued
#pragma unroll
        for (int m = 0; m < ND; ++m) {
            const int dil = J.dil[m];
            const int p1 = dil * (J.k - 1) / 2;
            const int Hx = 32 + p1 - p2;
            const int wm = r0 - 32 * m + step * R;  // global row of xin tile 0 / row 0
            const bool interior = wm - 32 >= 0 && wm + R <= L;  // block-uniform: every row of the X and H windows is inside the utterance
            char* sideX = side + (size_t)J.sx_off[m] * STRIDE;
            char* sideH = side + (size_t)J.sh_off[m] * STRIDE;
            const OpT* w1l = (const OpT*)J.w1[m] + (size_t)ct0 * J.ct1;
            const OpT* w2l = (const OpT*)J.w2[m] + (size_t)ct0 * J.ct2;

            // ---- phase A: M <- [H head | X head | lrelu(x) new rows] ---------------------------------------------------
            // KL = 2: the X tail is not copied out after conv1 any more; the publish writes its last Hx rows a second time into the
            // history buffer ("dual write") and the X head is restored WAVE-PRIVATELY -- each wave moves its own
            // 32 * MI channels, reads issued before its own dual writes (a wave's LDS operations execute in order) -- so the two
            // do not race without a barrier.  Saves the Hx-row LDS -> LDS copy of phase B.
            constexpr int WB = 64 * MI;            // bytes of a row owned by one wave
            constexpr int WCH = WB / 16;           // ... in 16-byte chunks
            constexpr int RPI = 64 / WCH;          // history rows one wave moves per instruction
            constexpr int NIT = (52 + RPI - 1) / RPI;
            uint4 hb[KL == 2 ? NIT : 1];
            if constexpr (KL == 2) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int i = min(it * RPI + lane / WCH, Hx - 1);
                    hb[it] = *(const uint4*)(sideX + (size_t)i * STRIDE + ct0 * 64 + (lane % WCH) * 16);
                }
            } else {
                rs_copy_rows<STRIDE, NT>(M + (size_t)(RS_HEAD - Hx) * STRIDE, sideX, Hx);
            }
            rs_copy_rows<STRIDE, NT>(M + (size_t)(RS_HROW - 2 * p2) * STRIDE, sideH, 2 * p2);
            {
                unsigned rowmask[NJ];
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt) {
                    const int t = wm + jt * 32 + lrow;
                    rowmask[jt] = (t >= 0 && t < L) ? 0xffffffffu : 0u;
                }
                char* xw = M + (size_t)(RS_HEAD + lrow) * STRIDE + (ct0 * 32 + half4) * 2;
                if (interior) publish_operand<OpT, C, MI, NJ, STRIDE, false>(xw, xin, rowmask);
                else publish_operand<OpT, C, MI, NJ, STRIDE, true>(xw, xin, rowmask);
                if constexpr (KL == 2) {
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        const int i = min(it * RPI + lane / WCH, Hx - 1);
                        *(uint4*)(M + (size_t)(RS_HEAD - Hx + i) * STRIDE + ct0 * 64 + (lane % WCH) * 16) = hb[it];
                    }
                    // dual write: new rows [R - Hx, R) -> history rows [0, Hx)   (Hx <= 52 < 64: tiles NJ-2 and NJ-1 only)
#pragma unroll
                    for (int jt = NJ - 2; jt < NJ; ++jt) {
                        const int srow = jt * 32 + lrow - (R - Hx);
                        char* tw = (srow >= 0 ? sideX + (size_t)srow * STRIDE : dump) + (ct0 * 32 + half4) * 2;
                        // (the same expression as the publish above, mask variant included: the packed words are reused, not recomputed)
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const f32x16& t = xin[mi][jt];
                                *(uint2*)(tw + (mi * 32 + 8 * g) * 2) =
                                    interior ? pack4_lrelu<OpT, false>(t[4 * g + 0], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3], 0u)
                                             : pack4_lrelu<OpT, true>(t[4 * g + 0], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3], rowmask[jt]);
                            }
                    }
                }
            }
            // residual tiles of this pair's output window (= input window - 32 rows): pure register renaming
            f32x16 res[MI][NJ];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                res[mi][0] = carry[m][mi];
#pragma unroll
                for (int jt = 1; jt < NJ; ++jt) res[mi][jt] = xin[mi][jt - 1];
                carry[m][mi] = xin[mi][NJ - 1];
            }
            stamp(0);  // phase A (waits for the x loads when m == 0)
            bar();
            stamp(1);

            // ---- conv1 (dilated): h rows [a_m, a_m + R), a_m = wm - 32 + p2 ---------------------------------------------
            f32x16 hacc[MI][NJ];
            {
                const float* bl = bias_l + (m * 2 + 0) * CP + ct0 * 32 + half4;
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
            run(hacc, M + (size_t)(RS_HEAD - Hx + lrow) * STRIDE + (lane >> 5) * 16, w1l, J.ct1, ctb1, dil);
            prefetch(w2l, J.ct2, ctb2);  // in flight across the publish + barriers
            stamp(2);  // conv1
            bar();  // every wave is done reading X
            stamp(3);

            // ---- phase B: save the X tail, M <- lrelu(h) new rows (+ their tail into the H history) ----------------------
            if constexpr (KL != 2) rs_copy_rows<STRIDE, NT>(sideX, M + (size_t)(RS_HEAD + R - Hx) * STRIDE, Hx);
            {
                const int am = wm - 32 + p2;
                unsigned rowmask[NJ];
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt) {
                    const int t = am + jt * 32 + lrow;
                    rowmask[jt] = (t >= 0 && t < L) ? 0xffffffffu : 0u;
                }
                char* hw = M + (size_t)(RS_HROW + lrow) * STRIDE + (ct0 * 32 + half4) * 2;
                if (interior) publish_operand<OpT, C, MI, NJ, STRIDE, false>(hw, hacc, rowmask);
                else publish_operand<OpT, C, MI, NJ, STRIDE, true>(hw, hacc, rowmask);
                // the last 2*p2 rows of the new h rows are next step's H head
                const int srow = lrow - (32 - 2 * p2);
                char* tw = (srow >= 0 ? sideH + (size_t)srow * STRIDE : dump) + (ct0 * 32 + half4) * 2;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x16& t = hacc[mi][NJ - 1];
                        *(uint2*)(tw + (mi * 32 + 8 * g) * 2) =
                            pack4_lrelu<OpT, true>(t[4 * g + 0], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3], rowmask[NJ - 1]);
                    }
            }
            stamp(4);  // phase B
            bar();
            stamp(5);

            // ---- conv2 accumulates onto the residual: x' rows [wm - 32, wm - 32 + R) ------------------------------------
            {
                const float* bl = bias_l + (m * 2 + 1) * CP + ct0 * 32 + half4;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 bv = *(const f32x4*)(bl + mi * 32 + 8 * g);
#pragma unroll
                        for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
                            for (int e = 0; e < 4; ++e) res[mi][jt][4 * g + e] += bv[e];
                    }
            }
            run(res, M + (size_t)(RS_HROW - 2 * p2 + lrow) * STRIDE + (lane >> 5) * 16, w2l, J.ct2, ctb2, 1);
            {  // next conv1's weights (next pair, or pair 0 of the next step)
                const int mn = (m + 1 < ND) ? m + 1 : 0;
                prefetch((const OpT*)J.w1[mn] + (size_t)ct0 * J.ct1, J.ct1, ctb1);
            }
            stamp(6);  // conv2
            bar();  // every wave is done reading H
            stamp(7);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt) xin[mi][jt] = res[mi][jt];
        }
        // ---- store the rows of this strip ------------------------------------------------------------------------------
        if constexpr (KL == 2) {
            const int wout = r0 - 32 * ND + step * R;
            if (step + 1 < nsteps) issue_loads(r0 + (step + 1) * R);  // the next step's rows: in flight under the store phase (not past the strip's last step:
                                                                       // those rows belong to the next strip's block, 1.5x the algorithmic traffic in round 3)
            if (yh) {
                using u32x2_t = __attribute__((ext_vector_type(2))) unsigned;
                using lds_u2 = __attribute__((address_space(3))) u32x2_t;
                using lds_u4 = __attribute__((address_space(3))) u32x4_t;
                {
                    unsigned b0 = dlh_lds;
                    asm volatile("" : "+v"(b0));
#pragma unroll
                    for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                            for (int g = 0; g < 4; ++g)
                                *(lds_u2*)(size_t)(b0 + (unsigned)(jt * 32 * TSTRH + (mi * 32 + 8 * g) * 2)) = __builtin_bit_cast(
                                    u32x2_t, pack4_h(xin[mi][jt][4 * g + 0], xin[mi][jt][4 * g + 1], xin[mi][jt][4 * g + 2], xin[mi][jt][4 * g + 3]));
                }
                bar();
                {
                    unsigned bb = ioh_lds;
                    unsigned v0 = ioh_voff + (unsigned)((wout - S0) * (C * 2));  // (wraps for rows in front of the strip: dropped)
                    asm volatile("" : "+v"(bb), "+v"(v0));
#pragma unroll
                    for (int it = 0; it < NCHH; ++it) {
                        const u32x4_t v = *(const lds_u4*)(size_t)(bb + (unsigned)(it * TSTRH));
                        __builtin_amdgcn_raw_buffer_store_b128(v, rs_dst, v0 + (unsigned)(it * (C * 2)), 0, 0);
                    }
                }
            } else {
            {
                unsigned b0 = dl_lds, b1 = dl_lds + (unsigned)(3 * 32 * TSTR);
                asm volatile("" : "+v"(b0), "+v"(b1));
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 v = {xin[mi][jt][4 * g + 0], xin[mi][jt][4 * g + 1], xin[mi][jt][4 * g + 2], xin[mi][jt][4 * g + 3]};
                            *(lds_f32x4*)(size_t)((jt < 3 ? b0 : b1) + (unsigned)((jt % 3) * 32 * TSTR + (mi * 32 + 8 * g) * 4)) = v;
                        }
            }
            bar();
            {
                unsigned b = io_lds;
                asm volatile("" : "+v"(b));
                unsigned v0[NCH / 8];  // (wraps for rows in front of the strip: dropped)
#pragma unroll
                for (int j = 0; j < NCH / 8; ++j) {
                    v0[j] = io_voff + (unsigned)((wout - S0) * (C * 4)) + (unsigned)(j * 4096);
                    asm volatile("" : "+v"(v0[j]));
                }
#pragma unroll
                for (int it = 0; it < NCH; ++it) {
                    const f32x4 v = *(const lds_f32x4*)(size_t)(b + (unsigned)(it * TSTR));
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs_dst, v0[it / 8] + (unsigned)((it % 8) * (C * 4)), 0, 0);
                }
            }
            }
        } else
        {
            const int wout = r0 - 32 * ND + step * R;
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) {
                const int tg = wout + jt * 32 + lrow;
                if (tg >= S0 && tg < S1) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 v = {xin[mi][jt][4 * g + 0], xin[mi][jt][4 * g + 1], xin[mi][jt][4 * g + 2], xin[mi][jt][4 * g + 3]};
                            if (yh) *(uint2*)(dsth + (size_t)tg * C + (ct0 + mi) * 32 + 8 * g + half4) = pack4_h(v[0], v[1], v[2], v[3]);
                            else *(f32x4*)(dst + (size_t)tg * C + (ct0 + mi) * 32 + 8 * g + half4) = v;
                        }
                }
            }
        }
        stamp(9);  // stores issued
    }
    if (stamps && lane < 12) {
        unsigned long long v = 0;
#pragma unroll
        for (int i = 0; i < 12; ++i) v = (lane == i) ? tsum[i] : v;
        if (lane == 10) v = (unsigned long long)nsteps;
        if (lane == 11) v = (unsigned long long)ji;
        a.ts[((size_t)blockIdx.x * NCO + wave) * 16 + lane] = v;
    }
}

}  // namespace rvcmi
