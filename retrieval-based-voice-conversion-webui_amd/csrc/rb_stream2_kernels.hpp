// k_rb_stream2: the streaming fused ResBlock1 (rb_stream_kernels.hpp) re-laid for TWO resident blocks per CU at C = 128.
//
// Why: with one wave per SIMD (k_rb_stream, NJ = 6) nothing fills the matrix pipe while a block is in its publish / history /
// IO phases (21-45 % of its time).  Two independent strips per CU do: tools/ubench/kloop.hip measures 34.9
// cycles per SIMD-MFMA for two NJ = 3 waves sharing a SIMD (37.0 for the one NJ = 6 wave), and a wave in its phases leaves the
// pipe to its partner.  What has to shrink for that is the block's footprint: 256 registers per wave (NJ = 3: R = 96 rows per
// step) and 80 KB of LDS (160 KB / 2) for a whole resblock (ND = 3) at k = 11:
//
//   * rows are 256 B, NOT padded: the 16-byte chunk c of the row that holds time t is stored at chunk  c ^ (t & 15)  (an XOR
//     swizzle keyed on the row's TIME, so a row keeps its layout when it moves between the tile and a history buffer).  The 16
//     lanes of a ds_read_b128 group read 16 consecutive-ish times => 16 distinct chunk slots => no bank conflict, same for the
//     ds_write_b64 groups of a publish.  In the K loop the swizzle costs one v_xor per k-step (6% of the LDS image saved);
//   * X and H use the SAME rows of the operand tile M (new rows at RS2_HEAD = 52 = the largest X history); k_rb_stream kept H
//     at row 10 so that the X tail could still be copied out after conv1 -- here no tail is ever copied:
//   * a publish writes its last rows twice, into M and into the pair's history buffer ("dual write"), and the history is
//     restored into the head of M by the SAME wave for the SAME channels (each wave owns 32 channels = 4 chunks of every row),
//     reads issued before that wave's dual writes: LDS executes a wave's operations in order, so no extra barrier is needed;
//   * M 151 rows + histories 156 rows (k = 11) + 1 dump row = 308 rows x 256 B + 3 KB of biases = 81 920 B = exactly half a CU.
//
// Everything else -- strips, the 32-row lag, the register-resident fp32 residual, job-major XCD order -- is k_rb_stream's.
// Schedule model: tools/model_rb_stream.py (run_strip2).
#pragma once
#include <type_traits>

#include "rb_stream_kernels.hpp"

namespace rvcmi {

constexpr int RS2_STRIDE = 256;
constexpr int RS2_HEAD = 52;   // first M row of the new X / H rows; X history (<= 52 rows) / H history (<= 10) in front
constexpr int RS2_SLACK = 3;   // the B prefetch past the last tap reads up to p1 + p2 + dil - 32 <= 3 rows behind the tile (unused)

using lds_cptr = const __attribute__((address_space(3))) char*;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;  // (uint4 is a class type: no address-space-qualified copies)
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
__device__ __forceinline__ u32x2 to_u32x2(uint2 v) { return u32x2{v.x, v.y}; }
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(lds_cptr)p; }
template <typename T>
__device__ __forceinline__ T lds_ld(unsigned a) {
    return *(const __attribute__((address_space(3))) T*)(size_t)a;
}
template <typename T>
__device__ __forceinline__ void lds_st(unsigned a, T v) {
    *(__attribute__((address_space(3))) T*)(size_t)a = v;
}

// Swizzled K loop, C_in = 128, one 32-channel output tile per wave: per tap 8 k-steps = two weight groups of 4 (ring A[2][4]:
// A[0] holds group 0 on entry -- rs2_prefetch --, the first half of a tap multiplies A[0] while A[1] loads, the second half the
// reverse).  row_addr: LDS byte address (256-aligned) of this lane's row at tap 0; s: (time of that row) & 15; h = lane >> 5.
// Weights: wbase is WAVE-UNIFORM (scalar registers: the tap loop advances it with scalar adds), loff = lane * 16 bytes.
template <typename OpT>
__device__ __forceinline__ typename Op<OpT>::frag rs2_wload(const OpT* wbase, unsigned loff) {
    return *(const typename Op<OpT>::frag*)((const char*)wbase + loff);
}
template <typename OpT>
__device__ __forceinline__ void rs2_prefetch(typename Op<OpT>::frag (&A)[2][4], const OpT* wbase, unsigned loff) {
#pragma unroll
    for (int k = 0; k < 4; ++k) A[0][k] = rs2_wload<OpT>(wbase + k * 512, loff);
}

template <typename OpT, int NJ>
__device__ __forceinline__ void rs2_conv(f32x16 (&acc)[NJ], typename Op<OpT>::frag (&A)[2][4], unsigned row_addr, int s, int h,
                                         const OpT* wbase, unsigned loff, int ntaps, int dil) {
    using frag = typename Op<OpT>::frag;
    auto tap_base = [&](unsigned ra, int sv) { return ra | (unsigned)(((h ^ sv) & 15) << 4); };
    unsigned base = tap_base(row_addr, s);
    frag Bf[2][NJ];
#pragma unroll
    for (int jt = 0; jt < NJ; ++jt) {
        Bf[0][jt] = lds_ld<frag>(base + jt * 32 * RS2_STRIDE);
        __builtin_amdgcn_sched_barrier(0);
    }
    const OpT* an = wbase + 4 * 512;            // group 1
    const OpT* const alast = wbase + (size_t)(2 * ntaps - 1) * 4 * 512;
    for (int t = 0; t < ntaps; ++t) {
        row_addr += (unsigned)(dil * RS2_STRIDE);
        s = (s + dil) & 15;
        const unsigned nbase = tap_base(row_addr, s);  // next tap (past the last one: rows behind the window, never used)
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int u = kk >> 2, k = kk & 3;
            const unsigned nb = (kk < 7) ? (base ^ (unsigned)((kk + 1) << 5)) : nbase;
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) {
                acc[jt] = Op<OpT>::mfma(A[u][k], Bf[kk & 1][jt], acc[jt]);
                Bf[(kk + 1) & 1][jt] = lds_ld<frag>(nb + jt * 32 * RS2_STRIDE);
                if (jt == NJ - 1) A[u ^ 1][k] = rs2_wload<OpT>(an + k * 512, loff);  // same k-step of the next group
                __builtin_amdgcn_sched_barrier(0);
            }
            if (k == 3) an = (an < alast) ? an + 4 * 512 : an;  // clamped: the tail re-requests the last group
        }
        base = nbase;
    }
}

// acc (MFMA D layout) -> lrelu -> OpT -> swizzled LDS rows.  rowbase: LDS address of this lane's row in tile 0 (256-aligned row
// start), off[g]: byte offset inside the row of this lane's 8-byte piece of chunk g (swizzle folded in).
template <typename OpT, int NJ, bool MASK>
__device__ __forceinline__ void rs2_publish(unsigned rowbase, const unsigned (&off)[4], const f32x16 (&acc)[NJ], const unsigned (&rowmask)[NJ]) {
#pragma unroll
    for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x16& t = acc[jt];
            lds_st<u32x2>(rowbase + jt * 32 * RS2_STRIDE + off[g],
                          to_u32x2(pack4_lrelu<OpT, MASK>(t[4 * g + 0], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3], rowmask[jt])));
        }
}

// the same values of ONE tile once more, into a history buffer (or the dump row)
template <typename OpT, bool MASK>
__device__ __forceinline__ void rs2_publish_tail(unsigned rowaddr, const unsigned (&off)[4], const f32x16& acc, unsigned rowmask) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
        lds_st<u32x2>(rowaddr + off[g], to_u32x2(pack4_lrelu<OpT, MASK>(acc[4 * g + 0], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3], rowmask)));
}

constexpr int RS3_GAPS = 72;    // MFMA gaps that carry fillers: the first 3 taps of a K loop (every kernel size has >= 3 taps)
template <int N>
using ic_t = std::integral_constant<int, N>;
template <int B, int E, typename F>
__device__ __forceinline__ void rs3_for(F&& f) {
    if constexpr (B < E) {
        f(ic_t<B>{});
        rs3_for<B + 1, E>(f);
    }
}
// ops [g * N / GAPS, (g + 1) * N / GAPS) of an N-op filler program go into gap g
template <int N, int G, typename F>
__device__ __forceinline__ void rs3_gap(F&& op) {
    rs3_for<(G * N) / RS3_GAPS, ((G + 1) * N) / RS3_GAPS>(op);
}

template <typename OpT>
__device__ __forceinline__ void rs3_prefetch(typename Op<OpT>::frag (&A)[8], const OpT* wbase, unsigned loff) {
#pragma unroll
    for (int k = 0; k < 8; ++k) A[k] = rs2_wload<OpT>(wbase + k * 512, loff);
}

// Swizzled K loop over one half (3 column tiles), C_in = 128, one 32-channel output tile per wave.  A holds tap 0 on entry
// (rs3_prefetch) and is refilled IN PLACE: right after the three MFMAs of k-step kk, A[kk] is requested for the next tap (the
// MFMAs have read it; the data lands a tap later).  Bf is a ring of 4 k-steps, read two k-steps ahead.  fill(gap) is called
// with a compile-time gap index after every MFMA of the first three taps.
template <typename OpT, bool ZERO, bool MW = false, int LA = 2, bool PEEL = true, typename F>
__device__ __forceinline__ void rs3_conv(f32x16 (&acc)[3], typename Op<OpT>::frag (&A)[8], unsigned row_addr, int s, int h, const OpT* wbase,
                                         unsigned loff, int ntaps, int dil, F&& fill) {
    using frag = typename Op<OpT>::frag;
    constexpr unsigned TS = 32u * RS2_STRIDE;
    auto tap_base = [&](unsigned ra, int sv) { return ra | (unsigned)(((h ^ sv) & 15) << 4); };
    unsigned base = tap_base(row_addr, s);
    static_assert(LA == 1 || LA == 2, "B fragments are read one or two k-steps ahead");
    constexpr int BM = LA == 2 ? 3 : 1;  // ring of 4 / 2 k-steps
    frag Bf[BM + 1][3];
#pragma unroll
    for (int q = 0; q < LA; ++q)
#pragma unroll
        for (int jt = 0; jt < 3; ++jt) {
            Bf[q][jt] = lds_ld<frag>((base ^ (unsigned)(q << 5)) + jt * TS);
            __builtin_amdgcn_sched_barrier(0);
        }
    const OpT* an = wbase + 8 * 512;  // tap 1
    const OpT* const alast = wbase + (size_t)(ntaps - 1) * 8 * 512;
    auto tap = [&](auto TI) {
        constexpr int ti = decltype(TI)::value;  // >= 0: one of the first three taps (fillers), -1: the plain loop body
        row_addr += (unsigned)(dil * RS2_STRIDE);
        s = (s + dil) & 15;
        const unsigned nbase = tap_base(row_addr, s);  // next tap (past the last one: rows behind the window, never used)
        rs3_for<0, 8>([&](auto KK) {
            constexpr int kk = decltype(KK)::value;
            const unsigned nb = (kk + LA < 8) ? (base ^ (unsigned)((kk + LA) << 5)) : (nbase ^ (unsigned)((kk + LA - 8) << 5));
            rs3_for<0, 3>([&](auto JT) {
                constexpr int jt = decltype(JT)::value;
                // MW (no fillers only): ONE s_waitcnt per k-step -- everything but the next k-step's three reads has landed --
                // instead of the compiler's one per MFMA (tools/ubench/kloop2.hip: 36.4 -> 35.3 cycles per MFMA for a lone wave)
                if constexpr (MW && LA == 2 && jt == 0) __builtin_amdgcn_s_waitcnt(0xC07F | (3 << 8));
                if constexpr (ZERO && ti == 0 && kk == 0) {
                    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    acc[jt] = Op<OpT>::mfma(A[kk], Bf[kk & BM][jt], z);
                } else {
                    acc[jt] = Op<OpT>::mfma(A[kk], Bf[kk & BM][jt], acc[jt]);
                }
                Bf[(kk + LA) & BM][jt] = lds_ld<frag>(nb + jt * TS);
                if constexpr (jt == 2) A[kk] = rs2_wload<OpT>(an + kk * 512, loff);  // same k-step of the next tap
                if constexpr (ti >= 0) fill(ic_t<ti * 24 + kk * 3 + jt>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        base = nbase;
        an = (an < alast) ? an + 8 * 512 : an;  // clamped: the last tap re-requests itself
    };
    if constexpr (PEEL) {
        tap(ic_t<0>{});
        tap(ic_t<1>{});
        tap(ic_t<2>{});
        for (int t = 3; t < ntaps; ++t) tap(ic_t<-1>{});
    } else {
        static_assert(PEEL || !ZERO, "the zero-initialising first k-step lives in the peeled tap 0");
        for (int t = 0; t < ntaps; ++t) tap(ic_t<-1>{});
    }
}

// One GROUP of 4 waves walks one strip.  GROUPS = 1: the group is the block (k_rb_stream2).  GROUPS = 2 (k_rb_stream2x): two
// groups share a block and every barrier is block-wide, so that the two strips can be held in ANTI-PHASE (see k_rb_stream2x).
// smem: this group's LDS image; lb: logical block (strip) index; tid: 0..255 inside the group.
template <typename OpT, int NJ, int ND, int GROUPS, int KL = 1>
__device__ __forceinline__ void rs2_body(const RbStreamArgs& a, char* smem, int lb, int tid) {
    using frag = typename Op<OpT>::frag;
    constexpr int C = 128, NT = 256, R = 32 * NJ, STRIDE = RS2_STRIDE;
    constexpr int MROWS = RS2_HEAD + R + RS2_SLACK;
    static_assert(R % 16 == 0, "the swizzle key of a lane's row must not change from step to step");
    auto bar = [] {
        if constexpr (GROUPS == 1) __syncthreads();
        else lds_barrier();
    };
    const unsigned M = lds_addr(smem);
    const unsigned side = M + MROWS * STRIDE;
    const unsigned dump = side + (unsigned)a.side_rows * STRIDE;
    float* bias_l = (float*)(smem + (size_t)(MROWS + a.side_rows + 1) * STRIDE);  // [ND][2][C]

    int ji = 0;
#pragma unroll
    for (int j = 1; j < 3; ++j)
        if (j < a.njobs && lb >= a.job[j].blk0 * a.B) ji = j;
    const RbStreamJob& J = a.job[ji];
    const int rem = lb - J.blk0 * a.B;
    const int b = rem / J.nstrips;
    const int strip = rem - b * J.nstrips;
    if (GROUPS == 1 && b >= a.B) return;  // (two groups per block: the launcher makes the grid exact)
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
    // two groups per block must execute the same number of barriers: the step count of a FULL strip of this job, also for a
    // shorter last strip (its extra steps load clamped rows and store nothing)
    const int nsteps = GROUPS == 1 ? (S1 - r0 + 32 * ND + R - 1) / R : (J.strip_len + HL + 32 * ND + R - 1) / R;

    const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));  // scalar: everything derived from it stays in SGPRs
    const int lane = tid & 63;
    const int h = lane >> 5, lrow = lane & 31;
    const int half4 = 4 * h;
    const int ch0 = wave * 32;  // this wave's output channels = chunks 4*wave .. 4*wave+3 of every row
    const unsigned loff = (unsigned)lane * 16u;

    // ---- zero the LDS image once, stage the biases -----------------------------------------------------------------------
    {
        const int total16 = (MROWS + a.side_rows + 1) * STRIDE / 16;
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (int i = tid; i < total16; i += NT) *(uint4*)(smem + (size_t)i * 16) = z;
        for (int i = tid; i < ND * 2 * C; i += NT) {
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
    if (stamps) tprev = __builtin_readcyclecounter();

    f32x16 carry[ND];
#pragma unroll
    for (int m = 0; m < ND; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) carry[m][e] = 0.f;

    // byte offsets of this lane's 8-byte pieces inside a row whose time is == key (mod 16)
    auto piece_offsets = [&](int key, int hh, unsigned (&off)[4]) {
#pragma unroll
        for (int g = 0; g < 4; ++g) off[g] = (unsigned)((((4 * wave + g) ^ key) & 15) << 4) + 8u * (unsigned)hh;
    };
    // wave-private history restore: rows [0, nrows) of `from` -> `to`, this wave's 4 chunks of each row; times key0 + i
    auto restore_read = [&](u32x4 (&buf)[4], int ln, unsigned from, int nrows, int key0, int it_n) {
#pragma unroll
        for (int it = 0; it < 4; ++it)
            if (it < it_n) {
                const int i = min(it * 16 + (ln >> 2), nrows - 1);
                const unsigned pos = (unsigned)((((4 * wave + (ln & 3)) ^ (key0 + i)) & 15) << 4);
                buf[it] = lds_ld<u32x4>(from + (unsigned)i * STRIDE + pos);
            }
    };
    auto restore_write = [&](const u32x4 (&buf)[4], int ln, unsigned to, int nrows, int key0, int it_n) {
#pragma unroll
        for (int it = 0; it < 4; ++it)
            if (it < it_n) {
                const int i = min(it * 16 + (ln >> 2), nrows - 1);
                const unsigned pos = (unsigned)((((4 * wave + (ln & 3)) ^ (key0 + i)) & 15) << 4);
                lds_st<u32x4>(to + (unsigned)i * STRIDE + pos, buf[it]);
            }
    };
    // The pair loop is unrolled (ND = 3) and everything a pair derives from the lane id is loop-invariant over the steps:
    // LICM hoisted ~45 such registers PER PAIR out of the step loop (94 spilled VGPRs at a 256-register budget).  An empty asm
    // makes the lane id opaque per pair-phase, so the few dozen address computations are redone where they are used.
    auto fresh = [](int v) {
        asm volatile("" : "+v"(v));
        return v;
    };

    // GROUPS = 2: the K-loop wave is alone on its SIMD's matrix pipe, so its loop is the lone-wave one of k_rb_stream3: the weight
    // ring is one whole tap deep and refilled in place (twice the look-ahead of A[2][4] in the same 32 registers); B fragments stay
    // one k-step ahead (two would need 24 more registers: 56 spills at the 256-register budget of two waves per SIMD)
    frag A[2][4];
    auto nofill = [](auto) {};
    auto prefetch = [&](const OpT* wb, unsigned lo) {
        if constexpr (GROUPS == 2) rs3_prefetch<OpT>(reinterpret_cast<frag(&)[8]>(A), wb, lo);
        else rs2_prefetch<OpT>(A, wb, lo);
    };
    auto conv = [&](f32x16 (&acc)[NJ], unsigned row_addr, int sv, int hv, const OpT* wb, unsigned lo, int ntaps, int dl) {
        if constexpr (GROUPS == 2 && KL == 2)  // B fragments two k-steps ahead + one s_waitcnt per k-step (236 registers)
            rs3_conv<OpT, false, true, 2, false>(acc, reinterpret_cast<frag(&)[8]>(A), row_addr, sv, hv, wb, lo, ntaps, dl, nofill);
        else if constexpr (GROUPS == 2)
            rs3_conv<OpT, false, false, 1, false>(acc, reinterpret_cast<frag(&)[8]>(A), row_addr, sv, hv, wb, lo, ntaps, dl, nofill);
        else rs2_conv<OpT, NJ>(acc, A, row_addr, sv, hv, wb, lo, ntaps, dl);
    };
    prefetch((const OpT*)J.w1[0] + (size_t)wave * J.ct1, loff);
    bar();
    // Two blocks share a CU; dispatched together with identical work they would walk in lockstep -- both in their publish
    // phases, then both in their K loops at half rate each (measured: 69 cycles per MFMA per wave in EVERY K loop) -- and the
    // second context would hide nothing.  The block of the second dispatch round (XCD-local index >= 32: a heuristic, it only
    // affects speed) therefore starts about half a pair-step late; nothing re-synchronises the two afterwards.
    if (GROUPS == 1 && a.skew > 0 && (((int)blockIdx.x >> 3) & 32)) {
        const int n = a.skew * (J.k + 3);
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(16);
    }
    const bool prio = (a.flags & 1) != 0;

    for (int step = 0; step < nsteps; ++step) {
        // ---- the next R rows of x, straight into the accumulator layout ---------------------------------------------------
        f32x16 xin[NJ];
        {
            const int w0 = r0 + step * R;
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) {
                const int tg = w0 + jt * 32 + lrow;
                const unsigned msk = (tg >= 0 && tg < L) ? 0xffffffffu : 0u;
                const int tgc = min(max(tg, 0), L - 1);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = *(const f32x4*)(src + (size_t)tgc * C + ch0 + 8 * g + half4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) xin[jt][4 * g + e] = mask_bits(v[e], msk);
                }
            }
        }
        stamp(8);
#pragma unroll
        for (int m = 0; m < ND; ++m) {
            const int dil = J.dil[m];
            const int p1 = dil * (J.k - 1) / 2;
            const int Hx = 32 + p1 - p2;
            const int wm = r0 - 32 * m + step * R;  // time of xin tile 0 / row 0  (== r0 mod 16)
            const bool interior = wm - 32 >= 0 && wm + R <= L;
            const unsigned sideX = side + (unsigned)J.sx_off[m] * STRIDE;
            const unsigned sideH = side + (unsigned)J.sh_off[m] * STRIDE;
            const OpT* w1l = (const OpT*)J.w1[m] + (size_t)wave * J.ct1;
            const OpT* w2l = (const OpT*)J.w2[m] + (size_t)wave * J.ct2;

            // ---- phase A: M <- [X history | lrelu(x) new rows]; the new tail also goes to the history buffer -----------------
            {
                const int ln = fresh(lane), lr = ln & 31, hh = ln >> 5;
                u32x4 hb[4];
                const int itn = (Hx + 15) >> 4;
                restore_read(hb, ln, sideX, Hx, r0 - Hx, itn);  // issued first: in flight during the publish
                unsigned rowmask[NJ];
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt) {
                    const int t = wm + jt * 32 + lr;
                    rowmask[jt] = (t >= 0 && t < L) ? 0xffffffffu : 0u;
                }
                unsigned off[4];
                piece_offsets(r0 + lr, hh, off);
                const unsigned xw = M + (unsigned)(RS2_HEAD + lr) * STRIDE;
                if (interior) rs2_publish<OpT, NJ, false>(xw, off, xin, rowmask);
                else rs2_publish<OpT, NJ, true>(xw, off, xin, rowmask);
                restore_write(hb, ln, M + (unsigned)(RS2_HEAD - Hx) * STRIDE, Hx, r0 - Hx, itn);
                // dual write: new rows [R - Hx, R) are next step's X history (Hx <= 52 < 64: tiles NJ-2 and NJ-1 only)
#pragma unroll
                for (int jt = NJ - 2; jt < NJ; ++jt) {
                    const int srow = jt * 32 + lr - (R - Hx);
                    const unsigned ta = srow >= 0 ? sideX + (unsigned)srow * STRIDE : dump;
                    if (interior) rs2_publish_tail<OpT, false>(ta, off, xin[jt], 0u);
                    else rs2_publish_tail<OpT, true>(ta, off, xin[jt], rowmask[jt]);
                }
            }
            f32x16 res[NJ];
            res[0] = carry[m];
#pragma unroll
            for (int jt = 1; jt < NJ; ++jt) res[jt] = xin[jt - 1];
            carry[m] = xin[NJ - 1];
            stamp(0);
            bar();
            stamp(1);

            // ---- conv1 (dilated): h times [am, am + R), am = wm - 32 + p2; lane row at tap 0 = M row HEAD - Hx + lrow ----------
            f32x16 hacc[NJ];
            {
                const int ln = fresh(lane), lr = ln & 31, hh = ln >> 5;
                const float* bl = bias_l + (m * 2 + 0) * C + ch0 + 4 * hh;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bv = *(const f32x4*)(bl + 8 * g);
#pragma unroll
                    for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
                        for (int e = 0; e < 4; ++e) hacc[jt][4 * g + e] = bv[e];
                }
                if (prio) __builtin_amdgcn_s_setprio(2);
                conv(hacc, M + (unsigned)(RS2_HEAD - Hx + lr) * STRIDE, (r0 - Hx + lr) & 15, hh, w1l, (unsigned)ln * 16u, J.k, dil);
                if (prio) __builtin_amdgcn_s_setprio(0);
                prefetch(w2l, (unsigned)ln * 16u);
            }
            stamp(2);
            bar();  // every wave is done reading X
            stamp(3);

            // ---- phase B: M <- [H history | lrelu(h) new rows]; the new tail also goes to the history buffer ------------------
            {
                const int ln = fresh(lane), lr = ln & 31, hh = ln >> 5;
                u32x4 hb[4];
                restore_read(hb, ln, sideH, 2 * p2, r0 - p2, 1);
                const int am = wm - 32 + p2;
                unsigned rowmask[NJ];
#pragma unroll
                for (int jt = 0; jt < NJ; ++jt) {
                    const int t = am + jt * 32 + lr;
                    rowmask[jt] = (t >= 0 && t < L) ? 0xffffffffu : 0u;
                }
                unsigned off[4];
                piece_offsets(r0 + p2 + lr, hh, off);
                const unsigned hw = M + (unsigned)(RS2_HEAD + lr) * STRIDE;
                if (interior) rs2_publish<OpT, NJ, false>(hw, off, hacc, rowmask);
                else rs2_publish<OpT, NJ, true>(hw, off, hacc, rowmask);
                restore_write(hb, ln, M + (unsigned)(RS2_HEAD - 2 * p2) * STRIDE, 2 * p2, r0 - p2, 1);
                const int srow = lr - (32 - 2 * p2);
                const unsigned ta = srow >= 0 ? sideH + (unsigned)srow * STRIDE : dump;
                if (interior) rs2_publish_tail<OpT, false>(ta, off, hacc[NJ - 1], 0u);
                else rs2_publish_tail<OpT, true>(ta, off, hacc[NJ - 1], rowmask[NJ - 1]);
            }
            stamp(4);
            bar();
            stamp(5);

            // ---- conv2 accumulates onto the residual: x' times [wm - 32, wm - 32 + R); lane row at tap 0 = HEAD - 2*p2 + lrow ---
            {
                const int ln = fresh(lane), lr = ln & 31, hh = ln >> 5;
                const float* bl = bias_l + (m * 2 + 1) * C + ch0 + 4 * hh;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bv = *(const f32x4*)(bl + 8 * g);
#pragma unroll
                    for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
                        for (int e = 0; e < 4; ++e) res[jt][4 * g + e] += bv[e];
                }
                if (prio) __builtin_amdgcn_s_setprio(2);
                conv(res, M + (unsigned)(RS2_HEAD - 2 * p2 + lr) * STRIDE, (r0 - p2 + lr) & 15, hh, w2l, (unsigned)ln * 16u, J.k, 1);
                if (prio) __builtin_amdgcn_s_setprio(0);
                const int mn = (m + 1 < ND) ? m + 1 : 0;
                prefetch((const OpT*)J.w1[mn] + (size_t)wave * J.ct1, (unsigned)ln * 16u);
            }
            stamp(6);
            bar();  // every wave is done reading H
            stamp(7);
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) xin[jt] = res[jt];
        }
        // ---- store the rows of this strip -------------------------------------------------------------------------------------
        {
            const int wout = r0 - 32 * ND + step * R;
#pragma unroll
            for (int jt = 0; jt < NJ; ++jt) {
                const int tg = wout + jt * 32 + lrow;
                if (tg >= S0 && tg < S1) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 v = {xin[jt][4 * g + 0], xin[jt][4 * g + 1], xin[jt][4 * g + 2], xin[jt][4 * g + 3]};
                        *(f32x4*)(dst + (size_t)tg * C + ch0 + 8 * g + half4) = v;
                    }
                }
            }
        }
        stamp(9);
    }
    if (stamps && lane < 12) {
        unsigned long long v = 0;
#pragma unroll
        for (int i = 0; i < 12; ++i) v = (lane == i) ? tsum[i] : v;
        if (lane == 10) v = (unsigned long long)nsteps;
        if (lane == 11) v = (unsigned long long)ji;
        a.ts[((size_t)lb * 4 + wave) * 16 + lane] = v;
    }
}


template <typename OpT, int NJ, int ND>
static __global__ void __launch_bounds__(256, 2) k_rb_stream2(RbStreamArgs a) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    // job-major logical order, one contiguous chunk per XCD (see k_rb_stream)
    const int nb = (int)gridDim.x, q = nb >> 3, r = nb & 7, xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    const int lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    rs2_body<OpT, NJ, ND, 1>(a, smem, lb, (int)threadIdx.x);
}

// k_rb_stream2x: the two strips that k_rb_stream2 runs as two independent blocks per CU, as two GROUPS of 4 waves inside ONE
// 8-wave block, held in anti-phase.  Measured (tools/ubench/kloop2.hip): a lone wave per SIMD hides ~5
// instructions per MFMA and runs the K loop at 34-37 cycles per MFMA, but its publish / history / IO phases issue no MFMA at
// all; two independent blocks per CU drift into lockstep (both in their K loops at 70 cycles per MFMA each, then both in
// their phases: k_rb_stream2 hid only a third of the phases).  Here every barrier is block-wide and group 1 starts one segment
// late: a group's segments alternate  phase A | conv1 | phase B | conv2 , so in every inter-barrier interval exactly ONE group
// runs a K loop -- alone on its SIMD's matrix pipe -- while the other runs a phase (VALU / LDS / global IO) in its shadow.
// Both groups take strips of the same resblock and utterance (the launcher makes the strip count even), i.e. K loops of the
// same length, and execute the same number of barriers (rs2_body: full-strip step count).
template <typename OpT, int NJ, int ND, int KL>
static __global__ void __launch_bounds__(512, 1) k_rb_stream2x(RbStreamArgs a) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int nb = (int)gridDim.x, q = nb >> 3, r = nb & 7, xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    const int lp = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;  // logical PAIR of strips
    const int grp = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    const size_t image = (size_t)(RS2_HEAD + 32 * NJ + RS2_SLACK + a.side_rows + 1) * RS2_STRIDE + (size_t)ND * 2 * 128 * sizeof(float);
    if (grp == 1) lds_barrier();  // one segment late
    rs2_body<OpT, NJ, ND, 2, KL>(a, smem + (size_t)grp * image, 2 * lp + grp, (int)(threadIdx.x & 255));
    if (grp == 0) lds_barrier();  // ... and group 0 waits for group 1's last segment
}

}  // namespace rvcmi
