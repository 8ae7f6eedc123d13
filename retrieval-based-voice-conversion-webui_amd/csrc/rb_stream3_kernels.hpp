// k_rb_stream3: the streaming fused ResBlock1 (rb_stream_kernels.hpp) with its publish / history phases moved INTO the MFMA
// shadow of the K loops, C = 128, one block per CU, one wave per SIMD.
//
// Why: k_rb_stream spends 21-45 % of a block's time in phases that issue no MFMA (activation -> fp16 -> LDS publishes, LDS -> LDS
// history copies), and with one wave per SIMD nothing else can use the matrix pipe meanwhile.  A second block per
// CU (k_rb_stream2) hides only part of it and doubles the strips' warm-up rows.  Here the SAME wave overlaps them: the R = 192
// rows of a step are two HALVES a / b of 96 rows (3 MFMA column tiles each), every conv runs as two K loops ("slots"), and the
// VALU / LDS work of the OTHER half rides in the issue slots between the MFMAs of the running K loop (a one-wave SIMD hides
// ~5 single-issue instructions per 32-cycle MFMA; the K loop itself needs 2-3):
//
//   slot 1  C1(m, a)  <- fillers: publish X(m, b) (+ its tail into the X history),  restore H history,  res[a] += b2
//   slot 2  C1(m, b)  <- fillers: publish H(m, a) = lrelu(hacc[a] + b1),  restore the NEXT pair's X history
//   slot 3  C2(m, a)  <- fillers: publish H(m, b) (+ tail into the H history),  res[b] += b2
//   slot 4  C2(m, b)  <- fillers: publish X(next pair, a) = lrelu(x'[a])   (pair 2: the next step's freshly loaded rows)
//
// one barrier after every slot.  A filler only ever writes rows the running K loop does not read: X and H live in SEPARATE
// buffers (256-byte rows with the time-keyed chunk swizzle of k_rb_stream2, so both fit: 247 + 203 + 156 history rows + 3 KB of
// biases = 158.5 KB), C1(m, a) reads X rows below XHEAD + 96 while X(m, b) lands above, and so on; the schedule with these
// exact row ranges is tools/model_rb_stream.py run_strip3, which evaluates every K loop before AND after its fillers.
// The price: every half streams the conv's weights again (A-fragment reuse 3 instead of 6: 42 B/clk/CU out of L2), which is why
// the weight ring is one whole tap (8 k-steps = 768 cycles) deep and refilled in place, and B fragments are read TWO k-steps
// ahead.  Global IO (x rows in, x' rows out, D-layout) stays outside the K loops: VMEM in a filler would put the weight ring's
// vmcnt behind HBM latencies.
//
// MEASURED (round 3): parity-green, 0.704 vs 0.641 ms per clip for k_rb_stream -- the plain taps run at 40.5 cycles per
// MFMA like k_rb_stream's loop, but the 72 gaps with fillers take ~70: the compiler needs 7-9 instructions per gap where a lone wave
// hides five, and the half tiles stream the weights out of L2 twice as fast.  Opt-in (RS_V3=1); kept as the reference for the
// slot / filler machinery (rs3_conv, rs3_gap) that k_rb_stream2x's K loop reuses.
#pragma once
#include <type_traits>

#include "rb_stream2_kernels.hpp"

namespace rvcmi {

constexpr int RS3_XHEAD = 52;   // first XB row of the new X rows (X history, <= 52 rows, in front)
constexpr int RS3_HHEAD = 10;   // first HB row of the new H rows (H history, <= 10 rows, in front)
constexpr int RS3_XSLACK = 3;   // the B prefetch past the last tap of conv1 reads <= p1 + p2 + dil - 32 <= 3 rows behind the tile
constexpr int RS3_HSLACK = 1;   // ... of conv2: one row
constexpr int RS3_R = 192;
constexpr int RS3_XROWS = RS3_XHEAD + RS3_R + RS3_XSLACK;
constexpr int RS3_HROWS = RS3_HHEAD + RS3_R + RS3_HSLACK;
constexpr int RS3_NPUB = 3 + 12 * 7;  // filler ops of publishing one half: 3 row masks + 12 x (4 activations, 2 packs, 1 store)

template <typename OpT>
__device__ __forceinline__ unsigned rs3_pack2(float a, float b) {
    using o2 = __attribute__((ext_vector_type(2))) OpT;
    o2 o = {(OpT)a, (OpT)b};
    return __builtin_bit_cast(unsigned, o);
}

// Op I of publishing the three tiles T0 .. T0+2 of a stream (src(ic<T>) -> that tile's accumulator):
//   I < 3        row mask of tile T0 + I  (trow = time of this lane's row in tile 0)
//   then 12 items (tile jt, channel group g) of 7 ops: 4 x [v = lrelu(x (+ bias))], 2 x [pack 2, mask], 1 x [8-byte store]
// TAIL: 0 none, 1 = X (tiles 4 and 5 also go to tail4 / tail5), 2 = H (tile 5 only).
template <typename OpT, int T0, bool BIAS, int TAIL, int I, typename SRC>
__device__ __forceinline__ void rs3_pub_op(SRC&& src, const f32x4 (&bias)[4], float (&v)[4], unsigned (&pk)[2], unsigned (&rowmask)[3], int trow, int L,
                                           unsigned rowaddr, const unsigned (&off)[4], unsigned tail4, unsigned tail5) {
    if constexpr (I < 3) {
        const int t = trow + 32 * (T0 + I);
        rowmask[I] = ((unsigned)t < (unsigned)L) ? 0xffffffffu : 0u;
    } else {
        constexpr int p = (I - 3) / 7, u = (I - 3) % 7, jt = p / 4, g = p % 4, T = T0 + jt;
        if constexpr (u < 4) {
            float x = src(ic_t<T>{})[4 * g + u];
            if constexpr (BIAS) x += bias[g][u];
            v[u] = lrelu_op<OpT>(x);
        } else if constexpr (u == 4) {
            pk[0] = rs3_pack2<OpT>(v[0], v[1]) & rowmask[jt];
        } else if constexpr (u == 5) {
            pk[1] = rs3_pack2<OpT>(v[2], v[3]) & rowmask[jt];
        } else {
            const u32x2 w = {pk[0], pk[1]};
            lds_st<u32x2>(rowaddr + (unsigned)(T * 32 * RS2_STRIDE) + off[g], w);
            if constexpr (TAIL == 1 && T == 4) lds_st<u32x2>(tail4 + off[g], w);
            if constexpr (TAIL != 0 && T == 5) lds_st<u32x2>(tail5 + off[g], w);
        }
    }
}

template <typename OpT>
static __global__ void __launch_bounds__(256, 1) k_rb_stream3(RbStreamArgs a) {
    using frag = typename Op<OpT>::frag;
    constexpr int C = 128, NT = 256, ND = 3, R = RS3_R, STRIDE = RS2_STRIDE;
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const unsigned XB = lds_addr(smem);
    const unsigned HB = XB + RS3_XROWS * STRIDE;
    const unsigned side = HB + RS3_HROWS * STRIDE;
    const unsigned dump = side + (unsigned)a.side_rows * STRIDE;
    const unsigned bias_a = dump + STRIDE;  // [ND][2][C] floats
    float* bias_l = (float*)(smem + (size_t)(RS3_XROWS + RS3_HROWS + a.side_rows + 1) * STRIDE);

    // ---- which job / utterance / strip: job-major logical order, one contiguous chunk per XCD (see k_rb_stream) --------
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
    const int L = a.L;
    const int S0 = strip * J.strip_len;
    const int S1 = min(L, S0 + J.strip_len);
    const int p2 = (J.k - 1) / 2;
    int HL = ND * p2;
#pragma unroll
    for (int m = 0; m < ND; ++m) HL += J.dil[m] * (J.k - 1) / 2;
    const int r0 = S0 - HL;
    const int nsteps = (S1 - r0 + 32 * ND + R - 1) / R;

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int hh = lane >> 5, lr = lane & 31;
    const int ch0 = wave * 32;
    const unsigned loff = (unsigned)lane * 16u;

    // ---- zero the LDS image once (histories start empty, slack rows stay zero), stage the biases --------------------------
    {
        const int total16 = (RS3_XROWS + RS3_HROWS + a.side_rows + 1) * STRIDE / 16;
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (int i = threadIdx.x; i < total16; i += NT) *(uint4*)(smem + (size_t)i * 16) = z;
        for (int i = threadIdx.x; i < ND * 2 * C; i += NT) {
            const int m = i / (2 * C), w = (i / C) & 1, c = i % C;
            bias_l[i] = (w ? J.b2[m] : J.b1[m])[c];
        }
    }

    unsigned long long tprev = 0, tsum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const bool stamps = a.ts != nullptr;
    auto stamp = [&](int ph) {
        if (stamps) {
            const unsigned long long t = __builtin_readcyclecounter();
            tsum[ph] += t - tprev;
            tprev = t;
        }
    };

    // Everything a slot derives from the lane id is loop-invariant over the steps; LICM would hoist ~60 addresses PER SLOT out of
    // the step loop (the first build: 536 spilled VGPRs, reloads in every MFMA gap and -- scratch being VMEM -- vmcnt(0) in front
    // of every weight fragment).  An empty asm makes the lane id opaque per slot, so the few address computations are redone
    // where they are used, in the MFMA shadow.
    auto fresh = [](int vv) {
        asm volatile("" : "+v"(vv));
        return vv;
    };
    // publish geometry of a lane: row lr of tile 0 of the new rows; 8-byte piece of chunk 4 * wave + g, swizzled by the row's time
    auto pub_geo = [&](int lr_, int hh_, int key, unsigned (&off)[4]) {
#pragma unroll
        for (int g = 0; g < 4; ++g) off[g] = (unsigned)((((4 * wave + g) ^ (key + lr_)) & 15) << 4) + 8u * (unsigned)hh_;
    };

    auto load_x = [&](f32x16 (&x)[3], int w0) {  // three tiles from time w0 on (clamped addresses, masked values)
#pragma unroll
        for (int T = 0; T < 3; ++T) {
            const int tg = w0 + T * 32 + lr;
            const unsigned msk = (tg >= 0 && tg < L) ? 0xffffffffu : 0u;
            const int tgc = min(max(tg, 0), L - 1);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 vv = *(const f32x4*)(src + (size_t)tgc * C + ch0 + 8 * g + 4 * hh);
#pragma unroll
                for (int e = 0; e < 4; ++e) x[T][4 * g + e] = mask_bits(vv[e], msk);
            }
        }
    };

    f32x16 carry[ND];
#pragma unroll
    for (int m = 0; m < ND; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) carry[m][e] = 0.f;

    frag A[8];
    rs3_prefetch<OpT>(A, (const OpT*)J.w1[0] + (size_t)wave * J.ct1, loff);
    f32x16 xin[2][3];
    load_x(xin[0], r0);
    load_x(xin[1], r0 + 96);
    __syncthreads();  // LDS image zeroed
    {   // prologue (exposed once per strip): half a of the first step's X; its history is the zero image
        f32x4 nb4[4];
        float v[4];
        unsigned pk[2], rmask[3], off[4];
        pub_geo(lr, hh, r0, off);
        const unsigned xrow = XB + (unsigned)(RS3_XHEAD + lr) * STRIDE;
        auto xs = [&](auto T) -> const f32x16& { return xin[0][decltype(T)::value]; };
        rs3_for<0, RS3_NPUB>([&](auto I) {
            rs3_pub_op<OpT, 0, false, 0, decltype(I)::value>(xs, nb4, v, pk, rmask, r0 + lr, L, xrow, off, 0u, 0u);
        });
    }
    __syncthreads();
    if (stamps) tprev = __builtin_readcyclecounter();

    for (int step = 0; step < nsteps; ++step) {
        f32x16 xnext[3];  // half a of the next step's rows (half b is loaded at the step boundary)
        auto pair = [&](auto MM) {
            constexpr int m = decltype(MM)::value;
            constexpr int mn = (m + 1 < ND) ? m + 1 : 0;
            const int dil = J.dil[m];
            const int p1 = dil * (J.k - 1) / 2;
            const int Hx = 32 + p1 - p2;
            const int Hxn = 32 + J.dil[mn] * (J.k - 1) / 2 - p2;
            const int wm = r0 - 32 * m + step * R;                       // time of the pair's X tile 0 / row 0  (== r0 mod 16)
            const int wn = (m + 1 < ND) ? wm - 32 : r0 + (step + 1) * R;  // ... of the NEXT pair's (next step's pair 0)
            const unsigned sideX = side + (unsigned)J.sx_off[m] * STRIDE;
            const unsigned sideH = side + (unsigned)J.sh_off[m] * STRIDE;
            const unsigned sideXn = side + (unsigned)J.sx_off[mn] * STRIDE;
            const OpT* w1l = (const OpT*)J.w1[m] + (size_t)wave * J.ct1;
            const OpT* w2l = (const OpT*)J.w2[m] + (size_t)wave * J.ct2;
            const OpT* w1n = (const OpT*)J.w1[mn] + (size_t)wave * J.ct1;

            // the pair's residual tiles (its output window = its input window - 32 rows): pure register renaming
            f32x16 res[2][3];
            res[0][0] = carry[m];
            res[0][1] = xin[0][0];
            res[0][2] = xin[0][1];
            res[1][0] = xin[0][2];
            res[1][1] = xin[1][0];
            res[1][2] = xin[1][1];
            carry[m] = xin[1][2];
            f32x16 hacc[2][3];
            f32x4 bv[4];
            float v[4];
            unsigned pk[2], rmask[3], off[4], rowa = 0, tail4 = 0, tail5 = 0;
            u32x4 hb[4];

            // ================= slot 1: C1(m, a)  |  X(m, b) + tail, H history restore, res[a] += b2 ==============================
            {
                const int ln = fresh(lane), l1 = ln & 31, h1 = ln >> 5;
                // X tile T of this pair: tiles 0..4 are res tiles 1..5, tile 5 is the new carry
                auto xs = [&](auto T) -> const f32x16& {
                    constexpr int t = decltype(T)::value;
                    if constexpr (t < 5) return res[(t + 1) / 3][(t + 1) % 3];
                    else return carry[m];
                };
                unsigned hpos = 0;
                constexpr int P0 = 5, N1 = P0 + RS3_NPUB + 2 + 12;
                auto op = [&](auto I) {
                    constexpr int i = decltype(I)::value;
                    if constexpr (i < 4) {
                        bv[i] = lds_ld<f32x4>(bias_a + (unsigned)(ch0 + 4 * h1) * 4u + (unsigned)(((m * 2 + 1) * C + 8 * i) * 4));
                    } else if constexpr (i == 4) {
                        pub_geo(l1, h1, r0, off);
                        rowa = XB + (unsigned)(RS3_XHEAD + l1) * STRIDE;
                        const int srow4 = l1 + Hx - 64, srow5 = l1 + Hx - 32;  // X tail: new rows [R - Hx, R) -> history rows [0, Hx)
                        tail4 = srow4 >= 0 ? sideX + (unsigned)srow4 * STRIDE : dump;
                        tail5 = sideX + (unsigned)srow5 * STRIDE;
                        // H history: 2 * p2 <= 10 rows, wave-private chunks, one b128 per lane (clamped duplicates)
                        const int hi = min(ln >> 2, 2 * p2 - 1);
                        hpos = (unsigned)hi * STRIDE + (unsigned)((((4 * wave + (ln & 3)) ^ (r0 - p2 + hi)) & 15) << 4);
                    } else if constexpr (i < P0 + RS3_NPUB) {
                        rs3_pub_op<OpT, 3, false, 1, i - P0>(xs, bv, v, pk, rmask, wm + l1, L, rowa, off, tail4, tail5);
                    } else if constexpr (i == P0 + RS3_NPUB) {
                        hb[0] = lds_ld<u32x4>(sideH + hpos);
                    } else if constexpr (i == P0 + RS3_NPUB + 1) {
                        lds_st<u32x4>(HB + (unsigned)(RS3_HHEAD - 2 * p2) * STRIDE + hpos, hb[0]);
                    } else {
                        constexpr int j = i - (P0 + RS3_NPUB + 2), jt = j / 4, g = j % 4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) res[0][jt][4 * g + e] += bv[g][e];
                    }
                };
                rs3_conv<OpT, true>(hacc[0], A, XB + (unsigned)(RS3_XHEAD - Hx + l1) * STRIDE, (r0 - Hx + l1) & 15, h1, w1l, (unsigned)ln * 16u,
                                    J.k, dil, [&](auto G) { rs3_gap<N1, decltype(G)::value>(op); });
                rs3_prefetch<OpT>(A, w1l, (unsigned)ln * 16u);
            }
            stamp(0);
            lds_barrier();
            stamp(1);

            // ================= slot 2: C1(m, b)  |  H(m, a), next pair's X history restore =====================================
            {
                const int ln = fresh(lane), l1 = ln & 31, h1 = ln >> 5;
                auto hs = [&](auto T) -> const f32x16& { return hacc[0][decltype(T)::value]; };
                unsigned rpos[4] = {0, 0, 0, 0};
                constexpr int P0 = 5, N2 = P0 + RS3_NPUB + 9;
                auto op = [&](auto I) {
                    constexpr int i = decltype(I)::value;
                    if constexpr (i < 4) {
                        bv[i] = lds_ld<f32x4>(bias_a + (unsigned)(ch0 + 4 * h1) * 4u + (unsigned)(((m * 2 + 0) * C + 8 * i) * 4));
                    } else if constexpr (i == 4) {
                        pub_geo(l1, h1, r0 + p2, off);
                        rowa = HB + (unsigned)(RS3_HHEAD + l1) * STRIDE;
                    } else if constexpr (i < P0 + RS3_NPUB) {
                        rs3_pub_op<OpT, 0, true, 0, i - P0>(hs, bv, v, pk, rmask, wm - 32 + p2 + l1, L, rowa, off, 0u, 0u);
                    } else if constexpr (i == P0 + RS3_NPUB) {
                        const int key0 = r0 - Hxn;  // time of the next pair's history row 0 (mod 16)
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            const int hi = min(it * 16 + (ln >> 2), Hxn - 1);
                            rpos[it] = (unsigned)hi * STRIDE + (unsigned)((((4 * wave + (ln & 3)) ^ (key0 + hi)) & 15) << 4);
                        }
                    } else if constexpr (i < P0 + RS3_NPUB + 5) {
                        hb[i - (P0 + RS3_NPUB + 1)] = lds_ld<u32x4>(sideXn + rpos[i - (P0 + RS3_NPUB + 1)]);
                    } else {
                        lds_st<u32x4>(XB + (unsigned)(RS3_XHEAD - Hxn) * STRIDE + rpos[i - (P0 + RS3_NPUB + 5)], hb[i - (P0 + RS3_NPUB + 5)]);
                    }
                };
                rs3_conv<OpT, true>(hacc[1], A, XB + (unsigned)(RS3_XHEAD - Hx + 96 + l1) * STRIDE, (r0 - Hx + l1) & 15, h1, w1l, (unsigned)ln * 16u,
                                    J.k, dil, [&](auto G) { rs3_gap<N2, decltype(G)::value>(op); });
                rs3_prefetch<OpT>(A, w2l, (unsigned)ln * 16u);
            }
            stamp(2);
            lds_barrier();
            stamp(3);
            if constexpr (m == ND - 1) {
                load_x(xnext, r0 + (step + 1) * R);  // half a of the next step's rows (clamped + masked past the end)
                stamp(8);
            }

            // ================= slot 3: C2(m, a)  |  H(m, b) + tail, res[b] += b2 ================================================
            {
                const int ln = fresh(lane), l1 = ln & 31, h1 = ln >> 5;
                auto hs = [&](auto T) -> const f32x16& { return hacc[1][decltype(T)::value - 3]; };
                constexpr int P0 = 5, N3 = P0 + RS3_NPUB + 4 + 12;
                auto op = [&](auto I) {
                    constexpr int i = decltype(I)::value;
                    if constexpr (i < 4) {
                        bv[i] = lds_ld<f32x4>(bias_a + (unsigned)(ch0 + 4 * h1) * 4u + (unsigned)(((m * 2 + 0) * C + 8 * i) * 4));
                    } else if constexpr (i == 4) {
                        pub_geo(l1, h1, r0 + p2, off);
                        rowa = HB + (unsigned)(RS3_HHEAD + l1) * STRIDE;
                        const int srow = l1 - (32 - 2 * p2);  // H tail: the last 2 * p2 rows of tile 5
                        tail5 = srow >= 0 ? sideH + (unsigned)srow * STRIDE : dump;
                    } else if constexpr (i < P0 + RS3_NPUB) {
                        rs3_pub_op<OpT, 3, true, 2, i - P0>(hs, bv, v, pk, rmask, wm - 32 + p2 + l1, L, rowa, off, 0u, tail5);
                    } else if constexpr (i < P0 + RS3_NPUB + 4) {  // b1 is dead: the same registers take b2
                        bv[i - (P0 + RS3_NPUB)] =
                            lds_ld<f32x4>(bias_a + (unsigned)(ch0 + 4 * h1) * 4u + (unsigned)(((m * 2 + 1) * C + 8 * (i - (P0 + RS3_NPUB))) * 4));
                    } else {
                        constexpr int j = i - (P0 + RS3_NPUB + 4), jt = j / 4, g = j % 4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) res[1][jt][4 * g + e] += bv[g][e];
                    }
                };
                rs3_conv<OpT, false>(res[0], A, HB + (unsigned)(RS3_HHEAD - 2 * p2 + l1) * STRIDE, (r0 - p2 + l1) & 15, h1, w2l, (unsigned)ln * 16u,
                                     J.k, 1, [&](auto G) { rs3_gap<N3, decltype(G)::value>(op); });
                rs3_prefetch<OpT>(A, w2l, (unsigned)ln * 16u);
            }
            stamp(4);
            lds_barrier();
            stamp(5);

            // ================= slot 4: C2(m, b)  |  X(next pair, a) ===============================================================
            {
                const int ln = fresh(lane), l1 = ln & 31, h1 = ln >> 5;
                // pairs 0, 1: the next pair's X is this pair's x' (tiles 0..2 are final after slot 3);  pair 2: the next step's rows
                auto xs = [&](auto T) -> const f32x16& {
                    if constexpr (m + 1 < ND) return res[0][decltype(T)::value];
                    else return xnext[decltype(T)::value];
                };
                constexpr int P0 = 1, N4 = P0 + RS3_NPUB;
                auto op = [&](auto I) {
                    constexpr int i = decltype(I)::value;
                    if constexpr (i == 0) {
                        pub_geo(l1, h1, r0, off);
                        rowa = XB + (unsigned)(RS3_XHEAD + l1) * STRIDE;
                    } else {
                        rs3_pub_op<OpT, 0, false, 0, i - P0>(xs, bv, v, pk, rmask, wn + l1, L, rowa, off, 0u, 0u);
                    }
                };
                rs3_conv<OpT, false>(res[1], A, HB + (unsigned)(RS3_HHEAD - 2 * p2 + 96 + l1) * STRIDE, (r0 - p2 + l1) & 15, h1, w2l,
                                     (unsigned)ln * 16u, J.k, 1, [&](auto G) { rs3_gap<N4, decltype(G)::value>(op); });
                rs3_prefetch<OpT>(A, w1n, (unsigned)ln * 16u);
            }
            stamp(6);
            lds_barrier();
            stamp(7);
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int jt = 0; jt < 3; ++jt) xin[q][jt] = res[q][jt];
        };
        pair(ic_t<0>{});
        pair(ic_t<1>{});
        pair(ic_t<2>{});
        // ---- store the rows of this strip, take over the next step's rows ---------------------------------------------------
        {
            const int wout = r0 - 32 * ND + step * R;
#pragma unroll
            for (int T = 0; T < 6; ++T) {
                const int tg = wout + T * 32 + lr;
                if (tg >= S0 && tg < S1) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x16& t = xin[T / 3][T % 3];
                        const f32x4 vv = {t[4 * g + 0], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]};
                        *(f32x4*)(dst + (size_t)tg * C + ch0 + 8 * g + 4 * hh) = vv;
                    }
                }
            }
        }
#pragma unroll
        for (int jt = 0; jt < 3; ++jt) xin[0][jt] = xnext[jt];
        load_x(xin[1], r0 + (step + 1) * R + 96);
        stamp(9);
    }
    if (stamps && lane < 12) {
        unsigned long long vv = 0;
#pragma unroll
        for (int i = 0; i < 12; ++i) vv = (lane == i) ? tsum[i] : vv;
        if (lane == 10) vv = (unsigned long long)nsteps;
        if (lane == 11) vv = (unsigned long long)ji;
        a.ts[((size_t)blockIdx.x * 4 + wave) * 16 + lane] = vv;
    }
}

}  // namespace rvcmi
