// IVF-Flat (faiss IndexIVFFlat, METRIC_L2) retrieval for gfx950: nearest-centroid probe, in-list
// squared-L2 scan with a per-wave top-k, and the reference's inverse-square blend
// (infer/modules/vc/pipeline.py:126-138, infer/lib/rtrvc.py:172-185; index recipe web.py:544-563).
//
// This half of the hot path is HBM/L2-bound byte movement (a few dozen 1-3 KB rows per query), so it
// is written as coalesced 16-byte row reads + wavefront-level reductions, NOT reshaped into a GEMM.
// Distances are evaluated in fp64 on the fp32 inputs as sum((q-v)^2) -- the arithmetic cost is
// irrelevant next to the row traffic, and it makes the ranking independent of summation order, which
// is what allows bit-exact indices against the CPU oracle (ties -> lowest id).
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

#include "common.hpp"
#include "glue_kernels.hpp"

namespace rvcmi {

struct BlobHeader {  // first 128 bytes of the device blob; everything the kernels need is offset-addressed
    uint64_t magic;  // "RVCMIIVF"
    uint32_t version;
    int32_t d;
    int32_t nprobe;
    int32_t pad0;
    int64_t ntotal;
    int64_t nlist;
    int64_t pos_last;  // list-major position of the row whose id == ntotal-1 (numpy's big_npy[-1])
    uint64_t off_centroids, off_list_offsets, off_ids, off_vecs;
    uint64_t total_bytes;
    uint64_t off_centroids_t;  // [d/4][nlist] float4: the coarse pass reads it lane-per-centroid, coalesced
    uint64_t off_cnorm;        // [nlist] fp32(|c|^2) (rounded from fp64) for the fp32 prefilter
    double cmax;               // max |c| over the centroids (error bound of the prefilter)
    uint64_t reserved[2];
};
static_assert(sizeof(BlobHeader) == 128, "blob header must be 128 bytes");
static const uint64_t kMagic = 0x465649494d435652ull;  // "RVCMIIVF" little-endian

__host__ __device__ static inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------

constexpr int QT = 4;  // queries per coarse block

// Coarse quantiser, nprobe == 1: for QT queries (LDS, broadcast reads) every lane owns one centroid
// at a time and accumulates QT fp64 distances for it -- no cross-lane reduction in the hot loop.
// Each lane keeps its running best; one wave-level (dist, id) argmin per query at the end.
__global__ void __launch_bounds__(256) k_coarse1(const float* __restrict__ q, const float4* __restrict__ cent_t,
                                                 int64_t nq, int64_t nlist, int d, int64_t* __restrict__ assign) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* qs = (float*)smem_raw;                               // [QT][d]
    double* bd = (double*)(smem_raw + align_up((size_t)QT * d * 4, 16));  // [4 waves][QT]
    int64_t* bi = (int64_t*)(bd + 4 * QT);
    const int64_t q0 = (int64_t)blockIdx.x * QT;
    const int nqb = (int)min((int64_t)QT, nq - q0);
    for (int i = threadIdx.x; i < QT * d; i += 256) {
        int qi = i / d;
        qs[i] = qi < nqb ? q[(q0 + qi) * d + (i - qi * d)] : 0.f;
    }
    __syncthreads();
    double best[QT];
    int64_t besti[QT];
#pragma unroll
    for (int k = 0; k < QT; ++k) { best[k] = INFINITY; besti[k] = INT64_MAX; }
    const int d4 = d >> 2;
    for (int64_t c = threadIdx.x; c < nlist; c += 256) {
        const float4* row = cent_t + c;  // element e of centroid c lives at cent_t[e*nlist + c]
        double acc[QT];
#pragma unroll
        for (int k = 0; k < QT; ++k) acc[k] = 0.0;
#pragma unroll 8
        for (int e = 0; e < d4; ++e) {
            const float4 v = row[(size_t)e * nlist];
#pragma unroll
            for (int k = 0; k < QT; ++k) {
                const float4 qq = *(const float4*)(qs + k * d + e * 4);
                double t0 = (double)qq.x - (double)v.x, t1 = (double)qq.y - (double)v.y;
                double t2 = (double)qq.z - (double)v.z, t3 = (double)qq.w - (double)v.w;
                acc[k] = fma(t0, t0, acc[k]);
                acc[k] = fma(t1, t1, acc[k]);
                acc[k] = fma(t2, t2, acc[k]);
                acc[k] = fma(t3, t3, acc[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < QT; ++k)
            if (acc[k] < best[k]) { best[k] = acc[k]; besti[k] = c; }  // c ascending per lane: ties keep the lower id
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < QT; ++k) {
        double bdv = best[k];
        int64_t biv = besti[k];
        for (int off = 32; off >= 1; off >>= 1) {
            double od = __shfl_xor(bdv, off, 64);
            int64_t oi = __shfl_xor(biv, off, 64);
            if (od < bdv || (od == bdv && oi < biv)) { bdv = od; biv = oi; }
        }
        if (lane == 0) { bd[wave * QT + k] = bdv; bi[wave * QT + k] = biv; }
    }
    __syncthreads();
    if (threadIdx.x < nqb) {
        const int k = threadIdx.x;
        double bdv = bd[k];
        int64_t biv = bi[k];
        for (int w = 1; w < 4; ++w) {
            double od = bd[w * QT + k];
            int64_t oi = bi[w * QT + k];
            if (od < bdv || (od == bdv && oi < biv)) { bdv = od; biv = oi; }
        }
        assign[q0 + k] = biv;
    }
}

// ---- nprobe == 1 fast path: fp32 prefilter on the exact-fp32 MFMA, fp64 verification of the near-ties -------------
//
// k_coarse_gemm: S[q][c] = fl32(|c|^2) - 2*dot32(q, c) with v_mfma_f32_32x32x2_f32 (an fmaf chain, so the classic
// dot-product error bound holds: |err| <= d*u*|q||c|, u = 2^-24).  Block = 64 queries x 128 centroids, K chunks of 32
// staged in LDS (row stride 33 floats: conflict-free ds_read_b32 for both operands).
constexpr int CG_K = 32, CG_S = 33;
using f32x16 = __attribute__((ext_vector_type(16))) float;
// MQ query tiles of 32 per block, NWC waves each owning 32 centroids.  (2, 4): 64 x 128 tiles for big nlist;
// (1, 1): one wave per 32 x 32 tile, so that small problems (nlist 256: 8 x 19 blocks) are not latency-bound.
template <int MQ, int NWC>
__global__ void __launch_bounds__(64 * NWC) k_coarse_gemm(const float* __restrict__ q, const float* __restrict__ cent,
                                                          const float* __restrict__ cn, int64_t nq, int64_t nlist, int d,
                                                          float* __restrict__ S) {
    constexpr int NT = 64 * NWC, QR = 32 * MQ, CR = 32 * NWC;
    __shared__ float Qs[QR * CG_S];
    __shared__ float Cs[CR * CG_S];
    const int64_t q0 = (int64_t)blockIdx.y * QR, c0 = (int64_t)blockIdx.x * CR;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x16 acc[MQ];
#pragma unroll
    for (int m = 0; m < MQ; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
    constexpr int SLOTS = (QR + CR) * (CG_K / 4);  // float4 slots per K chunk
    static_assert(SLOTS % NT == 0, "staging slots must divide evenly");
    constexpr int PER = SLOTS / NT;
    float4 pre[PER];
    auto fetch = [&](int k0) {  // next chunk -> registers (clamped rows: unconditional loads)
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int idx = threadIdx.x + it * NT;
            const int row = idx >> 3, c4 = idx & 7;
            const int kk = min(k0 + c4 * 4, d - 4);
            const float* src = row < QR ? q + min(q0 + row, nq - 1) * d : cent + min(c0 + (row - QR), nlist - 1) * d;
            const float4 v = *(const float4*)(src + kk);  // unconditional: no branch per load
            const bool ok = k0 + c4 * 4 < d;
            pre[it] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < d; k0 += CG_K) {
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int idx = threadIdx.x + it * NT;
            const int row = idx >> 3, c4 = idx & 7;
            float* dst = (row < QR ? Qs + row * CG_S : Cs + (row - QR) * CG_S) + c4 * 4;
            dst[0] = pre[it].x; dst[1] = pre[it].y; dst[2] = pre[it].z; dst[3] = pre[it].w;
        }
        __syncthreads();
        if (k0 + CG_K < d) fetch(k0 + CG_K);  // in flight while this chunk is multiplied
        const float* qa = Qs + (lane & 31) * CG_S + (lane >> 5);
        const float* cb = Cs + (wave * 32 + (lane & 31)) * CG_S + (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < CG_K / 2; ++ks) {
            const float bv = cb[2 * ks];
#pragma unroll
            for (int m = 0; m < MQ; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[m * 32 * CG_S + 2 * ks], bv, acc[m], 0, 0, 0);
        }
        __syncthreads();
    }
    const int64_t c = c0 + wave * 32 + (lane & 31);
    if (c < nlist) {
        const float cnc = cn[c];
#pragma unroll
        for (int m = 0; m < MQ; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t qr = q0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (qr < nq) S[qr * nlist + c] = cnc - 2.f * acc[m][r];
            }
    }
}

// One wave's share of a 32 x 32 score tile whose K dimension is split over the block's 4 waves (wave w: chunks w, w + 4, ...
// of CG_K floats).  `src[s]` = the row this lane stages in slot s (rows 0..31 of the A side, 32..63 of the B side; lane + 64 s
// = 8 row + float4 column).  DEPTH chunks are requested ahead: a wave walks d / 128 chunks (6 at d = 768), each a dependent
// memory round trip when fetched one ahead (13 us for a 152-tile launch); with DEPTH = 4 the walk is two round trips.
// NCH = chunks per wave (compile time: the ring is indexed statically); acc += A * B^T over this wave's chunks.
template <int NCH, int DEPTH>
__device__ __forceinline__ void ks_wave_tile(f32x16& acc, const float* const (&src)[8], float* st, int wave, int lane) {
    constexpr int D = DEPTH < NCH ? DEPTH : NCH;
    float4 pre[D][8];
    const int kstep = 4 * CG_K;
    const int kw = wave * CG_K;
    const int coff = (lane & 7) * 4;
#pragma unroll
    for (int c = 0; c < D; ++c)
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) pre[c][s2] = *(const float4*)(src[s2] + kw + c * kstep + coff);
    const float* qa = st + (lane & 31) * CG_S + (lane >> 5);
    const float* cb = st + (32 + (lane & 31)) * CG_S + (lane >> 5);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) {
            const int idx = lane + s2 * 64;
            float* dst = st + (idx >> 3) * CG_S + (idx & 7) * 4;
            const float4 v = pre[c % D][s2];
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        }
        if (c + D < NCH) {
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) pre[c % D][s2] = *(const float4*)(src[s2] + kw + (c + D) * kstep + coff);
        }
#pragma unroll
        for (int ks = 0; ks < CG_K / 2; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[2 * ks], cb[2 * ks], acc, 0, 0, 0);
    }
}

// Small problems (nlist 256 x 599 queries = 152 tiles): one 32 x 32 tile per block, the K dimension split over 4 waves
// (wave w takes chunks w, w+4, ...), each with its own staging area and a register prefetch of its next chunk; the four
// partial accumulators are added in a fixed order through LDS.  The single-wave version walked 24 chunks serially and was
// bound by 24 load latencies (34 us); a tree of partial sums stays inside the error bound k_coarse_pick assumes.
__global__ void __launch_bounds__(256) k_coarse_gemm_ks(const float* __restrict__ q, const float* __restrict__ cent,
                                                        const float* __restrict__ cn, int64_t nq, int64_t nlist, int d,
                                                        float* __restrict__ S) {
    __shared__ float St[4][64 * CG_S];   // per wave: 32 query rows then 32 centroid rows
    __shared__ float Red[3][64 * 16];
    const int64_t q0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* st = St[wave];
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    float4 pre[8];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = lane + it * 64;
            const int row = idx >> 3, c4 = idx & 7;
            const int kk = min(k0 + c4 * 4, d - 4);
            const float* src = row < 32 ? q + min(q0 + row, nq - 1) * d : cent + min(c0 + (row - 32), nlist - 1) * d;
            const float4 v = *(const float4*)(src + kk);
            const bool ok = k0 + c4 * 4 < d;
            pre[it] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
        }
    };
    const int kstep = 4 * CG_K;
    if (d == 768 || d == 256) {  // the shipped feature widths: deep prefetch ring (ks_wave_tile)
        const float* src[8];
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) {
            const int row = (lane + s2 * 64) >> 3;
            src[s2] = row < 32 ? q + min(q0 + row, nq - 1) * d : cent + min(c0 + (row - 32), nlist - 1) * d;
        }
        if (d == 768) ks_wave_tile<6, 4>(acc, src, st, wave, lane);
        else ks_wave_tile<2, 2>(acc, src, st, wave, lane);
    } else {
    if (wave * CG_K < d) fetch(wave * CG_K);
    for (int k0 = wave * CG_K; k0 < d; k0 += kstep) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = lane + it * 64;
            const int row = idx >> 3, c4 = idx & 7;
            float* dst = st + row * CG_S + c4 * 4;
            dst[0] = pre[it].x; dst[1] = pre[it].y; dst[2] = pre[it].z; dst[3] = pre[it].w;
        }
        if (k0 + kstep < d) fetch(k0 + kstep);  // in flight while this chunk is multiplied (wave-private LDS: no barrier)
        const float* qa = st + (lane & 31) * CG_S + (lane >> 5);
        const float* cb = st + (32 + (lane & 31)) * CG_S + (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < CG_K / 2; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[2 * ks], cb[2 * ks], acc, 0, 0, 0);
    }
    }
    if (wave > 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) Red[wave - 1][e * 64 + lane] = acc[e];
    }
    __syncthreads();
    if (wave == 0) {
        const int64_t c = c0 + (lane & 31);
        if (c < nlist) {
            const float cnc = cn[c];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float dot = ((acc[r] + Red[0][r * 64 + lane]) + Red[1][r * 64 + lane]) + Red[2][r * 64 + lane];
                const int64_t qr = q0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (qr < nq) S[qr * nlist + c] = cnc - 2.f * dot;
            }
        }
    }
}

// k_coarse_pick: one wave per query.  m = min_c S[q][c]; every centroid with S <= m + margin is a candidate
// (margin = 2 * rigorous rounding bound, so the exact winner is always among them -- usually alone); candidates are
// re-evaluated cooperatively in fp64 as sum((q-c)^2) and the exact (distance, id) minimum wins.
__global__ void __launch_bounds__(256) k_coarse_pick(const float* __restrict__ q, const float* __restrict__ cent,
                                                     const float* __restrict__ S, int64_t nq, int64_t nlist, int d, double cmax,
                                                     int64_t* __restrict__ assign) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t qi = (int64_t)blockIdx.x * 4 + wave;
    if (qi >= nq) return;
    const float* row = S + qi * nlist;
    const float* qp = q + qi * d;
    // the query slice of this lane (float4 chunks lane, lane + 64, ...; up to 4 of them = d <= 1024 stay in registers) and the
    // score row are requested together: one memory round trip
    const int d4 = d >> 2;
    float4 xq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xq[j] = lane + 64 * j < d4 ? ((const float4*)qp)[lane + 64 * j] : make_float4(0.f, 0.f, 0.f, 0.f);
    float m = INFINITY;
    for (int64_t c = lane; c < nlist; c += 64) m = fminf(m, row[c]);
    double qn2 = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        qn2 = fma((double)xq[j].x, (double)xq[j].x, qn2); qn2 = fma((double)xq[j].y, (double)xq[j].y, qn2);
        qn2 = fma((double)xq[j].z, (double)xq[j].z, qn2); qn2 = fma((double)xq[j].w, (double)xq[j].w, qn2);
    }
    for (int e = 256 * 4 + lane; e < d; e += 64) qn2 = fma((double)qp[e], (double)qp[e], qn2);  // (d > 1024 only)
    for (int off = 32; off >= 1; off >>= 1) {
        qn2 += __shfl_xor(qn2, off, 64);
        m = fminf(m, __shfl_xor(m, off, 64));
    }
    // |S_hat - S| <= (2d+4) u (|q| cmax + cmax^2), u = 2^-24; two such errors separate a false winner from the true one
    const double E = (2.0 * d + 4.0) * 5.9604644775390625e-08 * (sqrt(qn2) * cmax + cmax * cmax);
    const float thr = (float)((double)m + 2.0 * E + 1e-30) ;
    const float thr_up = __uint_as_float(__float_as_uint(fabsf(thr)) + 2u);  // round the threshold outwards
    const float lim = thr >= 0.f ? thr_up : -__uint_as_float(__float_as_uint(fabsf(thr)) - 2u);
    double best = INFINITY;
    int64_t besti = INT64_MAX;
    for (int64_t c0 = 0; c0 < nlist; c0 += 64) {
        const int64_t c = c0 + lane;
        const bool cand = c < nlist && row[c] <= lim;
        unsigned long long mask = __ballot(cand);
        while (mask) {
            // two candidates per trip (usually there are one or two in all): both centroids' loads are in flight together
            const int b0 = __builtin_ctzll(mask);
            mask &= mask - 1;
            const bool two = mask != 0;
            const int b1 = two ? __builtin_ctzll(mask) : b0;
            if (two) mask &= mask - 1;
            const float* cp0 = cent + (c0 + b0) * d;
            const float* cp1 = cent + (c0 + b1) * d;
            float4 y0[4], y1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e4 = min(lane + 64 * j, d4 - 1);
                y0[j] = ((const float4*)cp0)[e4];
                y1[j] = ((const float4*)cp1)[e4];
            }
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (lane + 64 * j < d4) {
                    const double x0 = (double)xq[j].x, x1 = (double)xq[j].y, x2 = (double)xq[j].z, x3 = (double)xq[j].w;
                    double t;
                    t = x0 - (double)y0[j].x; a0 = fma(t, t, a0); t = x1 - (double)y0[j].y; a0 = fma(t, t, a0);
                    t = x2 - (double)y0[j].z; a0 = fma(t, t, a0); t = x3 - (double)y0[j].w; a0 = fma(t, t, a0);
                    t = x0 - (double)y1[j].x; a1 = fma(t, t, a1); t = x1 - (double)y1[j].y; a1 = fma(t, t, a1);
                    t = x2 - (double)y1[j].z; a1 = fma(t, t, a1); t = x3 - (double)y1[j].w; a1 = fma(t, t, a1);
                }
            }
            for (int e = 256 * 4 + lane; e < d; e += 64) {  // (d > 1024 only)
                const double t0 = (double)qp[e] - (double)cp0[e], t1 = (double)qp[e] - (double)cp1[e];
                a0 = fma(t0, t0, a0);
                a1 = fma(t1, t1, a1);
            }
            for (int off = 32; off >= 1; off >>= 1) {
                a0 += __shfl_xor(a0, off, 64);
                a1 += __shfl_xor(a1, off, 64);
            }
            if (a0 < best || (a0 == best && c0 + b0 < besti)) { best = a0; besti = c0 + b0; }
            if (two && (a1 < best || (a1 == best && c0 + b1 < besti))) { best = a1; besti = c0 + b1; }
        }
    }
    if (lane == 0) assign[qi] = besti;
}

// General nprobe: full fp64 distance rows into scratch, then one wave per query extracts the nprobe
// smallest (dist, id) by repeated argmin.  Legacy indices only (tools/cmd/train-index.py used nprobe=9).
__global__ void __launch_bounds__(256) k_coarse_dist(const float* __restrict__ q, const float* __restrict__ cent,
                                                     int64_t nq, int64_t nlist, int d, double* __restrict__ dist) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nq * nlist) return;
    const int64_t qi = i / nlist, c = i - qi * nlist;
    const float* a = q + qi * d;
    const float* b = cent + c * d;
    double acc = 0.0;
    for (int e = 0; e < d; ++e) {
        double t = (double)a[e] - (double)b[e];
        acc = fma(t, t, acc);
    }
    dist[i] = acc;
}
__global__ void __launch_bounds__(64) k_coarse_select(double* __restrict__ dist, int64_t nlist, int nprobe,
                                                      int64_t* __restrict__ assign) {
    const int64_t qi = blockIdx.x;
    double* row = dist + qi * nlist;
    const int lane = threadIdx.x;
    for (int p = 0; p < nprobe; ++p) {
        double bdv = INFINITY;
        int64_t biv = INT64_MAX;
        for (int64_t c = lane; c < nlist; c += 64) {
            double v = row[c];
            if (v < bdv) { bdv = v; biv = c; }
        }
        for (int off = 32; off >= 1; off >>= 1) {
            double od = __shfl_xor(bdv, off, 64);
            int64_t oi = __shfl_xor(biv, off, 64);
            if (od < bdv || (od == bdv && oi < biv)) { bdv = od; biv = oi; }
        }
        if (lane == 0) {
            assign[qi * nprobe + p] = biv == INT64_MAX ? -1 : biv;
            if (biv != INT64_MAX) row[biv] = INFINITY;
        }
        __syncthreads();
    }
}

constexpr int KMAX = 8;

struct TopK {
    double d[KMAX];
    int64_t id[KMAX];
    int64_t pos[KMAX];
};
__device__ __forceinline__ bool before(double da, int64_t ia, double db, int64_t ib) { return da < db || (da == db && ia < ib); }
__device__ __forceinline__ void topk_insert(TopK& t, double dv, int64_t idv, int64_t posv) {
    // The list always carries KMAX sorted slots (static register indices); callers emit the first k.
    if (!before(dv, idv, t.d[KMAX - 1], t.id[KMAX - 1])) return;
    t.d[KMAX - 1] = dv;
    t.id[KMAX - 1] = idv;
    t.pos[KMAX - 1] = posv;
#pragma unroll
    for (int s = KMAX - 1; s >= 1; --s) {
        if (before(t.d[s], t.id[s], t.d[s - 1], t.id[s - 1])) {
            const double td = t.d[s]; t.d[s] = t.d[s - 1]; t.d[s - 1] = td;
            const int64_t ti = t.id[s]; t.id[s] = t.id[s - 1]; t.id[s - 1] = ti;
            const int64_t tp = t.pos[s]; t.pos[s] = t.pos[s - 1]; t.pos[s - 1] = tp;
        }
    }
}

// List scan: one 256-thread block per query = 16 groups of 16 lanes.  Group g takes rows g, g+16, ... of the
// probed list(s); inside a group lane s reads float4 chunks s, s+16, ... of the row (16 lanes x 16 B = 256
// contiguous bytes per load, 4 loads in flight per lane) and the 16 partial fp64 sums are combined with four
// xor-shuffles (wavefront-level reduction).  Every lane of a group carries the group's sorted top-8 in
// registers; thread 0 merges the 16 group lists.  ~40 rows per list (web.py:544) => 2-3 rows per group, so the
// whole list is in flight at once instead of being walked serially.
constexpr int SCAN_GROUPS = 16;
__global__ void __launch_bounds__(256) k_scan(const float* __restrict__ q, const int64_t* __restrict__ assign, int nprobe,
                                              const int64_t* __restrict__ list_off, const int64_t* __restrict__ ids,
                                              const float* __restrict__ vecs, int64_t nq, int d, int k,
                                              float* __restrict__ D, int64_t* __restrict__ I, int64_t* __restrict__ P,
                                              int* __restrict__ any_short, int dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int64_t qi = blockIdx.x;
    float* qs = (float*)smem_raw;
    TopK* merge = (TopK*)(smem_raw + align_up((size_t)d * 4, 16));
    for (int e = threadIdx.x; e < d; e += 256) qs[e] = q[qi * d + e];
    __syncthreads();
    TopK t;
#pragma unroll
    for (int s = 0; s < KMAX; ++s) { t.d[s] = INFINITY; t.id[s] = INT64_MAX; t.pos[s] = -1; }
    const int grp = threadIdx.x >> 4, sub = threadIdx.x & 15;
    const int d4 = d >> 2;
    for (int p = 0; p < nprobe; ++p) {
        const int64_t l = assign[qi * nprobe + p];
        if (l < 0) continue;
        const int64_t beg = list_off[l], end = list_off[l + 1];
        for (int64_t r = beg + grp; r < end; r += SCAN_GROUPS) {
            const float4* row = (const float4*)(vecs + r * d);
            double acc = 0.0;
            for (int c0 = sub; c0 < d4; c0 += 64) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c = c0 + 16 * u;
                    v[u] = (c < d4 && !(dbg & 2)) ? row[c] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c = c0 + 16 * u;
                    if (c < d4 && !(dbg & 1)) {
                        const float4 qq = *(const float4*)(qs + c * 4);
                        double t0 = (double)qq.x - (double)v[u].x, t1 = (double)qq.y - (double)v[u].y;
                        double t2 = (double)qq.z - (double)v[u].z, t3 = (double)qq.w - (double)v[u].w;
                        acc = fma(t0, t0, acc);
                        acc = fma(t1, t1, acc);
                        acc = fma(t2, t2, acc);
                        acc = fma(t3, t3, acc);
                    }
                }
            }
            for (int off = 8; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
            if (dbg & 1) acc = (double)(r & 1023);
            if (!(dbg & 4)) topk_insert(t, acc, ids[r], r);
            else if (acc < t.d[0]) { t.d[0] = acc; t.id[0] = r; t.pos[0] = r; }
        }
    }
    if (sub == 0) merge[grp] = t;
    __syncthreads();
    // 16 sorted lists x 8 = 128 candidates; thread i ranks candidate i against all others (broadcast LDS reads)
    // under the strict order (distance, id, slot) and, if it lands in the first k, writes that output slot.
    if (threadIdx.x < SCAN_GROUPS * KMAX && !(dbg & 8)) {
        const int me = threadIdx.x;
        const int mg = me / KMAX, ms = me % KMAX;
        const double md = merge[mg].d[ms];
        const int64_t mid = merge[mg].id[ms];
        int rank = 0;
        for (int o = 0; o < SCAN_GROUPS * KMAX; ++o) {
            const double od = merge[o / KMAX].d[o % KMAX];
            const int64_t oid = merge[o / KMAX].id[o % KMAX];
            const bool ob = od < md || (od == md && (oid < mid || (oid == mid && o < me)));
            rank += ob ? 1 : 0;
        }
        if (rank < k) {
            if (mid == INT64_MAX) {  // fewer than k candidates: faiss pads with -1 / FLT_MAX
                D[qi * k + rank] = FLT_MAX;
                I[qi * k + rank] = -1;
                P[qi * k + rank] = -1;
                atomicOr(any_short, 1);
            } else {
                D[qi * k + rank] = (float)md;
                I[qi * k + rank] = mid;
                P[qi * k + rank] = merge[mg].pos[ms];
            }
        }
    }
}

// ---- list-sorted query order (nprobe == 1) ------------------------------------------------------------------------------
// The scan is one block per query.  On the benchmark index 599 queries probe 26 lists (16.5 MB of list rows) and walk 568 MB
// of them; with queries in arrival order every XCD's 4 MB L2 sees all 26 lists, misses, and the launch runs at the ~7.5 TB/s the
// Infinity Cache delivers (76 us).  Sorting the queries by probed list (counting sort: histogram, one-block exclusive scan,
// scatter -- the order INSIDE a list is whatever the atomics give, results do not depend on it) and handing every XCD one
// contiguous range of the sorted order (hardware puts block b on XCD b % 8) leaves each L2 with an eighth of the lists.
// MEASURED (r3l): scan 77.1 -> 74.4 us on the benchmark index, 52.5 -> 52.2 us per clip at B = 16, for 17 us (3.7 us per clip at
// B = 16) of sorting launches: the scan is bound by its longest blocks (the size-biased lists: 416 rows = 26 dependent iterations),
// which the sorted order also packs onto the same XCD, not by where the rows come from.  Opt-in (IVF_SORT=1), off by default.
__global__ void __launch_bounds__(256) k_qsort_hist(const int64_t* __restrict__ assign, int64_t nq, int64_t nlist, int* __restrict__ cnt) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nq) return;
    const int64_t l = assign[i];
    atomicAdd(&cnt[(l < 0 || l >= nlist) ? nlist : l], 1);  // bucket nlist: queries without a list
}
__global__ void __launch_bounds__(1024) k_qsort_scan(int* __restrict__ cnt, int64_t n) {  // in place: counts -> exclusive offsets
    __shared__ int part[1024];
    const int64_t per = (n + 1023) / 1024;
    const int64_t b = (int64_t)threadIdx.x * per, e = b + per < n ? b + per : n;
    int s = 0;
    for (int64_t i = b; i < e; ++i) s += cnt[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - s;
    for (int64_t i = b; i < e; ++i) {
        const int c = cnt[i];
        cnt[i] = run;
        run += c;
    }
}
__global__ void __launch_bounds__(256) k_qsort_scatter(const int64_t* __restrict__ assign, int64_t nq, int64_t nlist, int* __restrict__ off,
                                                       int* __restrict__ perm) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nq) return;
    const int64_t l = assign[i];
    perm[atomicAdd(&off[(l < 0 || l >= nlist) ? nlist : l], 1)] = (int)i;
}
// logical index of block b when every XCD (b % 8) takes one contiguous range of [0, nb)
__device__ __forceinline__ int64_t xcd_contiguous(int64_t b, int64_t nb) {
    const int64_t q = nb >> 3, r = nb & 7, xcd = b & 7, slot = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// Specialised scan for d = 64*V (V = 12 for the 768-d v2 index, 4 for the 256-d v1 index): the query chunk of each
// lane lives in registers, every load of TWO rows is in flight before anything is consumed (one memory round trip per
// pair of rows instead of three per row), row indices are clamped so no load sits behind a branch.
// Specialised scan: d = LPR * 4 * VL, LPR lanes per row (16 or 32), G = 256 / LPR row groups, RU rows in flight per group.
// d = 768 uses 32 lanes per row: with 16 (48 floats of query + 96 of rows + fp64 temporaries per lane) the kernel needed
// 345 registers, ran one block per CU and turned the clip's 599 query blocks into three serial rounds of HBM-latency-bound
// work; at 32 lanes per row every block of the launch is resident at once.
template <int VL, int LPR, int RU>
// q and bfeats are NOT __restrict__: the fused blend runs in place (rvcmi_ivf_search_blend passes the same buffer for both).
// (3 blocks per CU: a clip's 599 query blocks must all be resident, see the register note above)
__global__ void __launch_bounds__(256, 3) k_scan_v(const float* q, const int64_t* __restrict__ assign, int nprobe,
                                                   const int64_t* __restrict__ list_off, const int64_t* __restrict__ ids,
                                                   const float* __restrict__ vecs, int64_t nq, int k, float* __restrict__ D,
                                                   int64_t* __restrict__ I, int64_t* __restrict__ P, int* __restrict__ any_short,
                                                   float* bfeats, float rate, float omr, int64_t pos_last, unsigned long long* ts = nullptr,
                                                   const int* __restrict__ perm = nullptr) {
    constexpr int d = LPR * 4 * VL;
    // dev only: wall-clock stamps of block 0's phases (RVCMI_IVF_STAMPS=1)
    auto stamp = [&](int i) {
        if (ts && blockIdx.x == 0 && threadIdx.x == 0) ts[i] = wall_clock64();
    };
    stamp(0);
    __shared__ float bd[KMAX];      // fused blend (bfeats != nullptr): the query's k results, by rank
    __shared__ long long bp[KMAX];
    constexpr int G = 256 / LPR;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    TopK* merge = (TopK*)smem_raw;
    const int64_t qi = perm ? (int64_t)perm[xcd_contiguous(blockIdx.x, gridDim.x)] : (int64_t)blockIdx.x;  // list-sorted order, one range per XCD
    const int grp = threadIdx.x / LPR, sub = threadIdx.x % LPR;
    // the query chunk of this lane, converted to fp64 ONCE (it used to be re-converted for every row: one of the four fp64-rate
    // instructions per element)
    double qd[VL][4];
#pragma unroll
    for (int i = 0; i < VL; ++i) {
        const float4 t4 = *(const float4*)(q + qi * d + (sub + LPR * i) * 4);
        qd[i][0] = (double)t4.x;
        qd[i][1] = (double)t4.y;
        qd[i][2] = (double)t4.z;
        qd[i][3] = (double)t4.w;
    }
    // the group's running top-8 lives in LDS and is maintained by the group's first lane: keeping it in registers (48 per
    // lane, all lanes) plus the unrolled compare-swap network cost ~100 VGPRs; a list has only 2-3 rows per group
    TopK& t = merge[grp];
    if (sub == 0) {
#pragma unroll
        for (int s = 0; s < KMAX; ++s) { t.d[s] = INFINITY; t.id[s] = INT64_MAX; t.pos[s] = -1; }
    }
    for (int p = 0; p < nprobe; ++p) {
        const int64_t l = assign[qi * nprobe + p];
        if (ts && threadIdx.x == 0 && blockIdx.x == 0) { asm volatile("" ::"v"((int)l)); ts[1] = wall_clock64(); }
        if (l < 0) continue;
        const int64_t beg = list_off[l], end = list_off[l + 1];
        if (ts && threadIdx.x == 0 && blockIdx.x == 0) { asm volatile("" ::"v"((int)beg), "v"((int)end)); ts[2] = wall_clock64(); }
        if (end <= beg) continue;
        // Row loop, software-pipelined: the loads of the NEXT 16 rows are in flight while the current ones are reduced (one
        // iteration used to be a full load -> fp64 -> shuffle -> LDS-insertion round trip, 3.5 us; lists are size-biased, the
        // longest one sets the kernel time).  Addresses are clamped to the list's last row, so every load is unconditional.
        auto load_rows = [&](float4(&v)[RU][VL], int64_t(&idv)[RU], int64_t r0) {
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int64_t rr = r0 + u * G;
                const int64_t rc = rr < end ? rr : end - 1;
                const float4* row = (const float4*)(vecs + rc * d);
#pragma unroll
                for (int i = 0; i < VL; ++i) v[u][i] = row[sub + LPR * i];
                idv[u] = ids[rc];
            }
        };
        auto proc_rows = [&](const float4(&v)[RU][VL], const int64_t(&idv)[RU], int64_t r0) {
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int64_t rr = r0 + u * G;
                double acc = 0.0;
#pragma unroll
                for (int i = 0; i < VL; ++i) {
                    const double t0 = qd[i][0] - (double)v[u][i].x, t1 = qd[i][1] - (double)v[u][i].y;
                    const double t2 = qd[i][2] - (double)v[u][i].z, t3 = qd[i][3] - (double)v[u][i].w;
                    acc = fma(t0, t0, acc);
                    acc = fma(t1, t1, acc);
                    acc = fma(t2, t2, acc);
                    acc = fma(t3, t3, acc);
                }
                for (int off = LPR / 2; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
                if (sub == 0 && rr < end && before(acc, idv[u], t.d[KMAX - 1], t.id[KMAX - 1])) {
                    int s = KMAX - 1;  // insertion into the sorted LDS list
                    while (s > 0 && before(acc, idv[u], t.d[s - 1], t.id[s - 1])) {
                        t.d[s] = t.d[s - 1];
                        t.id[s] = t.id[s - 1];
                        t.pos[s] = t.pos[s - 1];
                        --s;
                    }
                    t.d[s] = acc;
                    t.id[s] = idv[u];
                    t.pos[s] = rr;
                }
            }
        };
        float4 va[RU][VL], vb[RU][VL];
        int64_t ia[RU], ib[RU];
        int64_t r0 = beg + grp;
        stamp(3);
        load_rows(va, ia, r0);
        while (r0 < end) {
            load_rows(vb, ib, r0 + G * RU);
            proc_rows(va, ia, r0);
            r0 += G * RU;
            if (!(r0 < end)) break;
            load_rows(va, ia, r0 + G * RU);
            proc_rows(vb, ib, r0);
            r0 += G * RU;
        }
    }
    stamp(6);
    __syncthreads();
    stamp(7);
    if (threadIdx.x < G * KMAX) {
        const int me = threadIdx.x;
        const int mg = me / KMAX, ms = me % KMAX;
        const double md = merge[mg].d[ms];
        const int64_t mid = merge[mg].id[ms];
        int rank = 0;
#pragma unroll 8
        for (int o = 0; o < G * KMAX; ++o) {
            const double od = merge[o / KMAX].d[o % KMAX];
            const int64_t oid = merge[o / KMAX].id[o % KMAX];
            rank += (od < md || (od == md && (oid < mid || (oid == mid && o < me)))) ? 1 : 0;
        }
        if (rank < k) {
            if (mid == INT64_MAX) {
                D[qi * k + rank] = FLT_MAX;
                I[qi * k + rank] = -1;
                P[qi * k + rank] = -1;
                bd[rank] = FLT_MAX;
                bp[rank] = -1;
                atomicOr(any_short, 1);
            } else {
                D[qi * k + rank] = (float)md;
                I[qi * k + rank] = mid;
                P[qi * k + rank] = merge[mg].pos[ms];
                bd[rank] = (float)md;
                bp[rank] = merge[mg].pos[ms];
            }
        }
    }
    if (bfeats) {
        // pipeline.py:129-138 for this query's row, exactly as k_blend evaluates it (numpy's operation order); the rows were
        // just read by the scan, so the gather hits L2.  Only used without the realtime guard (which is a per-call decision).
#pragma clang fp contract(off)
        __syncthreads();
        float w[KMAX];
        for (int s = 0; s < k; ++s) {
            const float inv = div_rn(1.0f, bd[s]);
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
        long long pp[KMAX];  // gather positions in registers; all KMAX row loads of an element are issued before the sums
#pragma unroll
        for (int s = 0; s < KMAX; ++s) {
            const long long p = s < k ? bp[s] : 0;
            pp[s] = p < 0 ? pos_last : p;
        }
        for (int e = threadIdx.x; e < d; e += 256) {
            float gv[KMAX];
#pragma unroll
            for (int s = 0; s < KMAX; ++s) gv[s] = vecs[pp[s] * d + e];
            const float f = bfeats[qi * d + e];
            float acc = 0.f;
#pragma unroll
            for (int s = 0; s < KMAX; ++s) {
                if (s < k) {
                    const float prod = mul_rn(gv[s], w[s]);
                    acc = s == 0 ? prod : add_rn(acc, prod);
                }
            }
            bfeats[qi * d + e] = add_rn(mul_rn(acc, rate), mul_rn(omr, f));
        }
    }
    stamp(8);
}

// pipeline.py:129-138 with numpy's fp32 operation order:
//   weight = np.square(1/score); weight /= weight.sum(axis=1, keepdims=True)      (pairwise sum of 8)
//   npy = np.sum(big_npy[ix] * weight[..., None], axis=1)                          (sequential over k)
//   feats = npy*index_rate + (1-index_rate)*feats
// One block per query, threads over d.  id -1 gathers big_npy[-1] exactly like numpy does.
__global__ void __launch_bounds__(256) k_blend(float* __restrict__ feats, const float* __restrict__ D,
                                               const int64_t* __restrict__ P, const float* __restrict__ vecs, int d, int k,
                                               int64_t pos_last, float rate, float omr, const int* __restrict__ any_short,
                                               int skip_if_short) {
#pragma clang fp contract(off)
    if (skip_if_short && *any_short) return;
    const int64_t qi = blockIdx.x;
    float w[KMAX];
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
    int64_t pp[KMAX];  // all KMAX gathers of an element are issued before the sums (a serial load -> use chain per neighbour was 8 L2 round trips)
#pragma unroll
    for (int s = 0; s < KMAX; ++s) {
        const int64_t p = s < k ? P[qi * k + s] : 0;
        pp[s] = p < 0 ? pos_last : p;
    }
    for (int e = threadIdx.x; e < d; e += 256) {
        float gv[KMAX];
#pragma unroll
        for (int s = 0; s < KMAX; ++s) gv[s] = vecs[pp[s] * d + e];
        const float f = feats[qi * d + e];
        float acc = 0.f;
#pragma unroll
        for (int s = 0; s < KMAX; ++s) {
            if (s < k) {
                const float prod = mul_rn(gv[s], w[s]);
                acc = s == 0 ? prod : add_rn(acc, prod);
            }
        }
        feats[qi * d + e] = add_rn(mul_rn(acc, rate), mul_rn(omr, f));
    }
}

}  // namespace rvcmi
#include "ivf_lm_kernels.hpp"
namespace rvcmi {

// ---- index build (web.py:544-563: index.train = k-means for the nlist centroids, index.add = nearest-centroid lists) ----
// Lloyd update: centroid l <- mean of its members, members in ascending id order, fp64 accumulation (deterministic).
// One block per list, threads over the dimension.
__global__ void __launch_bounds__(256) k_list_mean(const float* __restrict__ x, const int64_t* __restrict__ order,
                                                   const int64_t* __restrict__ off, int d, float* __restrict__ cent) {
    const int64_t l = blockIdx.x;
    const int64_t beg = off[l], end = off[l + 1];
    if (end == beg) return;  // empty list: the host re-seeds it
    for (int e = threadIdx.x; e < d; e += 256) {
        double acc = 0.0;
        for (int64_t i = beg; i < end; ++i) acc += (double)x[order[i] * d + e];
        cent[l * d + e] = (float)(acc / (double)(end - beg));
    }
}
// squared distance (fp64) of every point to its assigned centroid: the k-means objective, one wave per point
__global__ void __launch_bounds__(256) k_assigned_dist(const float* __restrict__ x, const float* __restrict__ cent,
                                                       const int64_t* __restrict__ assign, int64_t n, int d, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    const float* xp = x + i * d;
    const float* cp = cent + assign[i] * d;
    double acc = 0.0;
    for (int e = lane; e < d; e += 64) {
        const double t = (double)xp[e] - (double)cp[e];
        acc += t * t;
    }
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) out[i] = acc;
}
// vecs[p] = x[order[p]]: the list-major copy of the vectors
__global__ void __launch_bounds__(256) k_gather_rows(const float* __restrict__ x, const int64_t* __restrict__ order, int64_t n, int d,
                                                     float* __restrict__ out) {
    const int64_t p = blockIdx.x;
    const float4* src = (const float4*)(x + order[p] * d);
    float4* dst = (float4*)(out + p * d);
    for (int e = threadIdx.x; e < d / 4; e += 256) dst[e] = src[e];
}

}  // namespace rvcmi

using namespace rvcmi;

struct rvcmi_ivf {
    int device = 0;
    BlobHeader hdr;
    char* blob = nullptr;  // device
    bool owns_blob = false;
    // search workspace (grown on demand; see rvcmi_ivf_reserve)
    int64_t cap_nq = 0;
    int cap_nprobe = 0;
    DevBuf assign, P, Dtmp, Itmp, flag, cdist, cscore, qcnt, qperm;
    int64_t cap_chunk = 0;  // queries per coarse-score chunk (bounds the nq x nlist fp32 scratch)
    // list-major scan (ivf_lm_kernels.hpp): row norms + statistics (once per handle), score scratch and work items (per capacity)
    DevBuf lm_rn, lm_S, lm_items, lm_n, lm_qinfo;
    bool lm_ready = false;
    double lm_vmax = 0.0;      // max |v| over the stored rows (error bound of the fp32 prefilter)
    int64_t lm_maxlen = 0;     // longest list
    int64_t lm_cap_items = 0;  // capacity of lm_items
    int64_t lm_cap_nq = 0;     // queries a call may bring for the list-major kernels (0: not reserved -- they never run)
    int64_t lm_buf_nq = 0;     // queries lm_S / lm_qinfo / lm_items are sized for = one PASS: a larger call runs several (the score scratch,
                               // queries x longest list x 4 B, is kept at or below 1 GiB)
    bool lm_warned = false;    // the one-time "large call on the query-major scan" diagnostic was printed
    Profiler prof;
    // dev / test options (common.hpp Options): IVF_COARSE_F64 (brute-force fp64 coarse quantizer), IVF_GENERIC (any-d scan kernel),
    // IVF_STAMPS (prints; syncs), IVF_DBG, IVF_SORT (1 = scan the queries in list-sorted, XCD-contiguous order).  Read from RVCMI_<KEY> once at handle creation; later only rvcmi_ivf_set_option.
    rvcmi::Options opt;
    rvcmi_ivf() { opt.load_env({"IVF_COARSE_F64", "IVF_GENERIC", "IVF_STAMPS", "IVF_DBG", "IVF_SORT", "IVF_LM", "IVF_LM_MIN"}); }
    const float* centroids() const { return (const float*)(blob + hdr.off_centroids); }
    const float4* centroids_t() const { return (const float4*)(blob + hdr.off_centroids_t); }
    const float* cnorm() const { return (const float*)(blob + hdr.off_cnorm); }
    const int64_t* list_off() const { return (const int64_t*)(blob + hdr.off_list_offsets); }
    const int64_t* ids() const { return (const int64_t*)(blob + hdr.off_ids); }
    const float* vecs() const { return (const float*)(blob + hdr.off_vecs); }
    ~rvcmi_ivf() {
        if (owns_blob && blob) (void)hipFree(blob);
    }
};

namespace rvcmi {

static void validate(int d, int64_t n, int64_t nlist, int nprobe) {
    if (d < 4 || (d & 3)) RVCMI_FAIL(RVCMI_ERR_INVALID, "dimension %d must be a positive multiple of 4", d);
    if (n < 0 || nlist < 1) RVCMI_FAIL(RVCMI_ERR_INVALID, "bad sizes n=%lld nlist=%lld", (long long)n, (long long)nlist);
    if (nprobe < 1) RVCMI_FAIL(RVCMI_ERR_INVALID, "nprobe must be >= 1");
}

static std::vector<char> build_blob(int d, int64_t n, int64_t nlist, int nprobe, const float* centroids,
                                    const int64_t* list_offsets, const int64_t* ids, const float* vecs) {
    validate(d, n, nlist, nprobe);
    if (list_offsets[0] != 0 || list_offsets[nlist] != n) RVCMI_FAIL(RVCMI_ERR_INVALID, "list_offsets do not cover [0, n)");
    for (int64_t l = 0; l < nlist; ++l)
        if (list_offsets[l + 1] < list_offsets[l]) RVCMI_FAIL(RVCMI_ERR_INVALID, "list_offsets not monotone at %lld", (long long)l);
    BlobHeader h;
    memset(&h, 0, sizeof(h));
    h.magic = kMagic;
    h.version = 1;
    h.d = d;
    h.nprobe = nprobe;
    h.ntotal = n;
    h.nlist = nlist;
    h.pos_last = -1;
    for (int64_t i = 0; i < n; ++i)
        if (ids[i] == n - 1) h.pos_last = i;
    if (h.pos_last < 0) h.pos_last = n > 0 ? n - 1 : 0;
    uint64_t off = sizeof(BlobHeader);
    h.off_centroids = off;
    off = align_up(off + (uint64_t)nlist * d * 4, 256);
    h.off_list_offsets = off;
    off = align_up(off + (uint64_t)(nlist + 1) * 8, 256);
    h.off_ids = off;
    off = align_up(off + (uint64_t)std::max<int64_t>(n, 1) * 8, 256);
    h.off_vecs = off;
    off = align_up(off + (uint64_t)std::max<int64_t>(n, 1) * d * 4, 256);
    h.off_centroids_t = off;
    off = align_up(off + (uint64_t)nlist * d * 4, 256);
    h.off_cnorm = off;
    off = align_up(off + (uint64_t)nlist * 4, 256);
    h.total_bytes = off;
    std::vector<char> blob(off, 0);
    memcpy(blob.data(), &h, sizeof(h));
    memcpy(blob.data() + h.off_centroids, centroids, (size_t)nlist * d * 4);
    memcpy(blob.data() + h.off_list_offsets, list_offsets, (size_t)(nlist + 1) * 8);
    {
        float* cnp = (float*)(blob.data() + h.off_cnorm);
        double cmax2 = 0.0;
        for (int64_t c = 0; c < nlist; ++c) {
            double n2 = 0.0;
            for (int e = 0; e < d; ++e) n2 += (double)centroids[c * d + e] * (double)centroids[c * d + e];
            cnp[c] = (float)n2;
            cmax2 = std::max(cmax2, n2);
        }
        ((BlobHeader*)blob.data())->cmax = std::sqrt(cmax2);
    }
    {
        float* ct = (float*)(blob.data() + h.off_centroids_t);
        const int d4 = d / 4;
        for (int64_t c = 0; c < nlist; ++c)
            for (int e = 0; e < d4; ++e) memcpy(ct + ((size_t)e * nlist + c) * 4, centroids + c * d + e * 4, 16);
    }
    if (n) {
        memcpy(blob.data() + h.off_ids, ids, (size_t)n * 8);
        memcpy(blob.data() + h.off_vecs, vecs, (size_t)n * d * 4);
    }
    return blob;
}

static rvcmi_ivf* from_host_blob(const std::vector<char>& blob, int device) {
    DeviceGuard dg(device);
    std::unique_ptr<rvcmi_ivf> h(new rvcmi_ivf());
    h->device = device;
    memcpy(&h->hdr, blob.data(), sizeof(BlobHeader));
    HIP_CHECK(hipMalloc((void**)&h->blob, blob.size()));
    h->owns_blob = true;
    HIP_CHECK(hipMemcpy(h->blob, blob.data(), blob.size(), hipMemcpyHostToDevice));
    return h.release();
}

// ---- faiss on-disk format (impl/index_write.cpp / index_read.cpp of faiss, as recalled; see
//      oracle/ivf_oracle.py for the independent python twin used to cross-check this reader) ----
struct Reader {
    FILE* f;
    const char* path;
    void read(void* dst, size_t n) {
        if (n && fread(dst, 1, n, f) != n) RVCMI_FAIL(RVCMI_ERR_IO, "%s: truncated file", path);
    }
    template <typename T>
    T get() {
        T v;
        read(&v, sizeof(T));
        return v;
    }
    void fourcc(char out[5]) {
        read(out, 4);
        out[4] = 0;
    }
};
struct FileCloser {
    FILE* f;
    ~FileCloser() {
        if (f) fclose(f);
    }
};

static void read_index_header(Reader& r, int& d, int64_t& ntotal, int& metric) {
    d = r.get<int32_t>();
    ntotal = r.get<int64_t>();
    (void)r.get<int64_t>();
    (void)r.get<int64_t>();
    (void)r.get<uint8_t>();  // is_trained
    metric = r.get<int32_t>();
    if (metric > 1) (void)r.get<float>();
}

static rvcmi_ivf* read_faiss(const char* path, int device) {
    FILE* f = fopen(path, "rb");
    if (!f) RVCMI_FAIL(RVCMI_ERR_IO, "cannot open '%s'", path);
    FileCloser fc{f};
    Reader r{f, path};
    char cc[5];
    r.fourcc(cc);
    if (strcmp(cc, "IwFl")) RVCMI_FAIL(RVCMI_ERR_IO, "%s: fourcc '%s' is not an IndexIVFFlat (IwFl)", path, cc);
    int d, metric;
    int64_t ntotal;
    read_index_header(r, d, ntotal, metric);
    if (metric != 1) RVCMI_FAIL(RVCMI_ERR_IO, "%s: metric %d; only METRIC_L2 (web.py:547) is supported", path, metric);
    const uint64_t nlist = r.get<uint64_t>();
    const uint64_t nprobe = r.get<uint64_t>();
    r.fourcc(cc);
    if (strcmp(cc, "IxF2") && strcmp(cc, "IxFl")) RVCMI_FAIL(RVCMI_ERR_IO, "%s: quantizer '%s' is not a flat L2 index", path, cc);
    int qd, qmetric;
    int64_t qn;
    read_index_header(r, qd, qn, qmetric);
    if (qmetric != 1) RVCMI_FAIL(RVCMI_ERR_IO, "%s: the coarse quantizer '%s' uses metric %d; only a flat L2 quantizer is supported", path, cc, qmetric);
    const uint64_t nfl = r.get<uint64_t>();
    if (qd != d || (uint64_t)qn != nlist || nfl != nlist * (uint64_t)d) RVCMI_FAIL(RVCMI_ERR_IO, "%s: quantizer shape mismatch", path);
    std::vector<float> cent(nfl);
    r.read(cent.data(), nfl * 4);
    const int dm_type = r.get<int8_t>();  // DirectMap::Type: 0 NoMap, 1 Array (a vector<idx_t> follows), 2 Hashtable
    if (dm_type != 0 && dm_type != 1)
        RVCMI_FAIL(RVCMI_ERR_IO, "%s: direct map type %d (Hashtable) is not supported; re-write the index without a direct map "
                   "(RVC never builds one, web.py:547-571)", path, dm_type);
    const uint64_t dmn = r.get<uint64_t>();
    if (fseek(f, (long)(dmn * 8), SEEK_CUR)) RVCMI_FAIL(RVCMI_ERR_IO, "%s: truncated direct map", path);
    r.fourcc(cc);
    if (strcmp(cc, "ilar")) RVCMI_FAIL(RVCMI_ERR_IO, "%s: inverted lists '%s' are not ArrayInvertedLists", path, cc);
    const uint64_t nl2 = r.get<uint64_t>(), code_size = r.get<uint64_t>();
    if (nl2 != nlist || code_size != 4ull * d) RVCMI_FAIL(RVCMI_ERR_IO, "%s: inverted-list header mismatch", path);
    r.fourcc(cc);
    std::vector<int64_t> off(nlist + 1, 0);
    std::vector<uint64_t> sizes(nlist, 0);
    const uint64_t cnt = r.get<uint64_t>();
    if (!strcmp(cc, "full")) {
        if (cnt != nlist) RVCMI_FAIL(RVCMI_ERR_IO, "%s: 'full' size vector length", path);
        r.read(sizes.data(), cnt * 8);
    } else if (!strcmp(cc, "sprs")) {
        std::vector<uint64_t> pairs(cnt);
        r.read(pairs.data(), cnt * 8);
        for (uint64_t i = 0; i + 1 < cnt; i += 2) {
            if (pairs[i] >= nlist) RVCMI_FAIL(RVCMI_ERR_IO, "%s: sparse list id out of range", path);
            sizes[pairs[i]] = pairs[i + 1];
        }
    } else {
        RVCMI_FAIL(RVCMI_ERR_IO, "%s: unknown list size encoding '%s'", path, cc);
    }
    for (uint64_t l = 0; l < nlist; ++l) off[l + 1] = off[l] + (int64_t)sizes[l];
    const int64_t n = off[nlist];
    if (n != ntotal) RVCMI_FAIL(RVCMI_ERR_IO, "%s: lists hold %lld rows, header says %lld", path, (long long)n, (long long)ntotal);
    std::vector<float> vecs((size_t)std::max<int64_t>(n, 1) * d);
    std::vector<int64_t> ids(std::max<int64_t>(n, 1));
    for (uint64_t l = 0; l < nlist; ++l) {
        if (!sizes[l]) continue;
        r.read(vecs.data() + (size_t)off[l] * d, sizes[l] * d * 4);
        r.read(ids.data() + off[l], sizes[l] * 8);
    }
    auto blob = build_blob(d, n, (int64_t)nlist, (int)std::max<uint64_t>(1, nprobe), cent.data(), off.data(), ids.data(), vecs.data());
    return from_host_blob(blob, device);
}

static void write_faiss(const rvcmi_ivf* h, const char* path) {
    std::vector<char> blob(h->hdr.total_bytes);
    HIP_CHECK(hipMemcpy(blob.data(), h->blob, blob.size(), hipMemcpyDeviceToHost));
    const BlobHeader& b = h->hdr;
    FILE* f = fopen(path, "wb");
    if (!f) RVCMI_FAIL(RVCMI_ERR_IO, "cannot create '%s'", path);
    FileCloser fc{f};
    auto put = [&](const void* p, size_t n) {
        if (n && fwrite(p, 1, n, f) != n) RVCMI_FAIL(RVCMI_ERR_IO, "%s: short write", path);
    };
    auto header = [&](int32_t d, int64_t nt) {
        int64_t dummy = 1 << 20;
        uint8_t trained = 1;
        int32_t metric = 1;
        put(&d, 4); put(&nt, 8); put(&dummy, 8); put(&dummy, 8); put(&trained, 1); put(&metric, 4);
    };
    const uint64_t nlist = b.nlist, nprobe = b.nprobe;
    put("IwFl", 4);
    header(b.d, b.ntotal);
    put(&nlist, 8); put(&nprobe, 8);
    put("IxF2", 4);
    header(b.d, b.nlist);
    const uint64_t nfl = nlist * (uint64_t)b.d;
    put(&nfl, 8);
    put(blob.data() + b.off_centroids, nfl * 4);
    int8_t dm = 0;
    uint64_t zero = 0;
    put(&dm, 1); put(&zero, 8);
    put("ilar", 4);
    const uint64_t code_size = 4ull * b.d;
    put(&nlist, 8); put(&code_size, 8);
    const int64_t* off = (const int64_t*)(blob.data() + b.off_list_offsets);
    uint64_t nonzero = 0;
    for (uint64_t l = 0; l < nlist; ++l) nonzero += off[l + 1] > off[l];
    if (nonzero > nlist / 2) {
        put("full", 4);
        put(&nlist, 8);
        for (uint64_t l = 0; l < nlist; ++l) { uint64_t s = off[l + 1] - off[l]; put(&s, 8); }
    } else {
        put("sprs", 4);
        uint64_t cnt = nonzero * 2;
        put(&cnt, 8);
        for (uint64_t l = 0; l < nlist; ++l)
            if (off[l + 1] > off[l]) { uint64_t s = off[l + 1] - off[l]; put(&l, 8); put(&s, 8); }
    }
    for (uint64_t l = 0; l < nlist; ++l) {
        const uint64_t s = off[l + 1] - off[l];
        if (!s) continue;
        put(blob.data() + b.off_vecs + (size_t)off[l] * b.d * 4, s * b.d * 4);
        put(blob.data() + b.off_ids + (size_t)off[l] * 8, s * 8);
    }
}

static int num_cus_ivf() {
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    static std::atomic<int> cached[64];
    int v = cached[dev & 63].load();
    if (!v) {
        hipDeviceProp_t p;
        HIP_CHECK(hipGetDeviceProperties(&p, dev));
        v = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
        cached[dev & 63].store(v);
    }
    return v;
}

// List-major scan: usable for nprobe 1, d a multiple of the MFMA K chunk, a list count the planner holds in LDS, and a score scratch
// (nq x longest list, fp32) of at most 1 GiB; fewer than 16 queries keep the one-launch query-major kernel.  (A list longer than the
// 2048-entry LDS row of the selector is worked on in place in the global scratch, round 5.)  Option IVF_LM: 0 = never, default 1.
// Returns nullptr when the list-major kernels can take a call of nq queries, else the reason they cannot (printed once per handle
// for calls of 64 queries or more: such a call silently costs the 34x row re-reads of the query-major scan otherwise).
static const char* lm_unusable_reason(const rvcmi_ivf* h, int64_t nq) {
    const BlobHeader& b = h->hdr;
    // from 16 queries on (round 5; 64 before): a realtime chunk's 16 guarded rows take 8 + 11 + 13 us in plan + tiles + select against 51 us in
    // the query-major kernel, whose one block per query walks a ~300-row list alone (chunk p50 0.487 -> 0.465 ms, ABAB); option IVF_LM_MIN
    const int lm_min = h->opt.geti("IVF_LM_MIN", 16);
    if (nq < lm_min) {
        static thread_local char why[64];
        snprintf(why, sizeof(why), "fewer than %d queries%s", lm_min, lm_min == 16 ? "" : " (option IVF_LM_MIN)");
        return why;
    }
    if (std::min<int64_t>(b.nprobe, b.nlist) != 1) return "nprobe > 1";
    if ((b.d % CG_K) != 0) return "d is not a multiple of 32";
    if (b.nlist > LM_MAXL) return "more than 16384 lists (the one-block planner counts them in LDS)";
    if (b.ntotal < 1) return "empty index";
    if (!h->lm_ready || h->lm_maxlen < 1) return "list statistics not reserved";
    if (nq >= (1ll << 30)) return "2^30 queries or more";
    // a pass holds at least 64 queries (lm_pass_queries): with one list of more than 2^22 rows even that pass would exceed the 1 GiB bound of
    // the score scratch (the bound was documented but the 64-query floor could break it by a wide margin)
    if ((int64_t)align_up((uint64_t)h->lm_maxlen, 32) * 4 * 64 > ((int64_t)1 << 30))
        return "a list of more than 4 194 304 rows (64 queries x the longest list x 4 B would exceed the 1 GiB score scratch)";
    return nullptr;
}
// queries per pass of the list-major kernels: the fp32 score scratch (queries x longest list) stays at or below 1 GiB
static int64_t lm_pass_queries(const rvcmi_ivf* h) {
    const int64_t pitch = (int64_t)align_up((uint64_t)std::max<int64_t>(h->lm_maxlen, 1), 32);
    return std::max<int64_t>(64, ((int64_t)1 << 30) / (pitch * 4));
}
static bool lm_usable(const rvcmi_ivf* h, int64_t nq) { return lm_unusable_reason(h, nq) == nullptr; }
static void lm_reserve(rvcmi_ivf* h, int64_t nq) {
    const BlobHeader& b = h->hdr;
    if (std::min<int64_t>(b.nprobe, b.nlist) != 1 || (b.d % CG_K) != 0 || b.nlist > LM_MAXL || b.ntotal < 1) return;
    if (!h->lm_ready) {  // once per handle (never inside a stream capture: reserve() runs before the first search)
        h->lm_rn.alloc((size_t)b.ntotal * 4);
        h->lm_n.alloc(256);
        HIP_CHECK(hipMemset(h->lm_n.p, 0, 256));
        hipLaunchKernelGGL(k_lm_row_norms, dim3((unsigned)((b.ntotal + 3) / 4)), dim3(256), 0, nullptr, h->vecs(), b.ntotal, b.d, h->lm_rn.as<float>());
        const int64_t m = std::max<int64_t>(b.ntotal, b.nlist);
        hipLaunchKernelGGL(k_lm_stats, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, nullptr, h->lm_rn.as<float>(), b.ntotal, h->list_off(), b.nlist,
                           h->lm_n.as<unsigned>() + 16);
        HIP_CHECK(hipGetLastError());
        unsigned st2[2];
        HIP_CHECK(hipMemcpy(st2, h->lm_n.as<unsigned>() + 16, 8, hipMemcpyDeviceToHost));
        float vm2;
        memcpy(&vm2, &st2[0], 4);
        h->lm_vmax = std::sqrt((double)vm2) * (1.0 + 1e-6);  // (the norms are fp32-rounded: a hair of head room)
        h->lm_maxlen = (int64_t)st2[1];
        static std::atomic<unsigned long long> attr_done{0};
        const unsigned long long bit = 1ull << (h->device & 63);
        if (!(attr_done.load() & bit)) {
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lm_plan), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (LM_MAXL + 1) * 4));
            attr_done.fetch_or(bit);
        }
        h->lm_ready = true;
    }
    // (declining to allocate must also retire an earlier, smaller reservation: reserve() raises cap_nq to the new value, and a later
    //  search with nq <= cap_nq would otherwise run the list-major kernels on buffers sized for fewer queries)
    h->lm_cap_nq = h->lm_buf_nq = 0;
    if (!lm_usable(h, std::max<int64_t>(nq, 64))) return;  // (option IVF_LM_MIN lowers only the routing threshold)
    const int64_t pitch = (int64_t)align_up((uint64_t)h->lm_maxlen, 32);
    const int64_t nb = std::min<int64_t>(std::max<int64_t>(nq, 1), lm_pass_queries(h));  // queries of one pass
    h->lm_S.alloc((size_t)nb * pitch * 4);
    h->lm_qinfo.alloc((size_t)nb * sizeof(LmQuery));
    // worst case of the planner: every probed list adds at most one partial query tile on top of nb / 32 full ones
    h->lm_cap_items = (nb / 32 + std::min<int64_t>(nb, b.nlist) + 1) * (pitch / 32);
    h->lm_items.alloc((size_t)h->lm_cap_items * sizeof(LmItem));
    h->lm_buf_nq = nb;
    h->lm_cap_nq = std::max<int64_t>(nq, 1);
}

static void reserve(rvcmi_ivf* h, int64_t nq) {
    const int np = (int)std::min<int64_t>(h->hdr.nprobe, h->hdr.nlist);
    const bool have_scores = np > 1 || (h->cscore.p && h->cap_chunk > 0);
    if (nq <= h->cap_nq && np <= h->cap_nprobe && h->flag.p && have_scores) {
        // the list-major scratch has a capacity of its own: a reserve above its 1 GiB bound retires it (lm_reserve), a later call
        // that fits gets it back at its own size
        if (h->lm_ready && nq > h->lm_cap_nq && lm_usable(h, nq)) {
            DeviceGuard dg(h->device);
            lm_reserve(h, nq);
        }
        return;
    }
    DeviceGuard dg(h->device);
    nq = std::max<int64_t>(nq, h->cap_nq);
    h->assign.alloc((size_t)std::max<int64_t>(nq, 1) * np * 8);
    h->P.alloc((size_t)std::max<int64_t>(nq, 1) * KMAX * 8);
    h->Dtmp.alloc((size_t)std::max<int64_t>(nq, 1) * KMAX * 4);
    h->Itmp.alloc((size_t)std::max<int64_t>(nq, 1) * KMAX * 8);
    if (!h->flag.p) h->flag.alloc(256);
    h->qcnt.alloc((size_t)(h->hdr.nlist + 1) * 4);
    h->qperm.alloc((size_t)std::max<int64_t>(nq, 1) * 4);
    if (np > 1) h->cdist.alloc((size_t)std::max<int64_t>(nq, 1) * h->hdr.nlist * 8);
    else {
        const int64_t per = std::max<int64_t>(64, (int64_t)(512ll << 20) / (h->hdr.nlist * 4));  // <= 512 MiB of scores
        h->cap_chunk = std::min<int64_t>(std::max<int64_t>(nq, 1), per);
        h->cscore.alloc((size_t)h->cap_chunk * h->hdr.nlist * 4);
    }
    h->cap_nq = nq;
    h->cap_nprobe = np;
    lm_reserve(h, nq);
}

struct BlendFuse {
    float* feats;
    float rate, omr;
};
// returns true when the blend of `bf` was done inside the scan kernel
static bool search(rvcmi_ivf* h, int64_t nq, const float* q, int k, float* D, int64_t* I, hipStream_t st,
                   const BlendFuse* bf = nullptr) {
    if (!h || nq < 0 || (nq && (!q || !D || !I))) RVCMI_FAIL(RVCMI_ERR_INVALID, "null argument");
    if (k < 1 || k > KMAX)
        RVCMI_FAIL(RVCMI_ERR_INVALID, "k=%d outside [1,%d]: the kernels keep at most %d neighbours per query (RVC asks for 8, "
                   "pipeline.py:126; legacy top-1 tools/cmd/infer-pm-index256.py:161)", k, KMAX, KMAX);
    if (nq == 0) return false;
    DeviceGuard dg(h->device);  // a C caller's current device need not be the handle's (restored on return)
    reserve(h, nq);
    const BlobHeader& b = h->hdr;
    const int d = b.d;
    const int np = (int)std::min<int64_t>(b.nprobe, b.nlist);
    HIP_CHECK(hipMemsetAsync(h->flag.p, 0, 4, st));
    const double cflops = 3.0 * (double)nq * b.nlist * d;
    if (np == 1 && !h->opt.on("IVF_COARSE_F64")) {
        // fp32 MFMA prefilter + fp64 verification (exactly the fp64 argmin; see k_coarse_pick)
        h->prof.launch("ivf_coarse", 2.0 * (double)nq * b.nlist * d, (double)nq * d * 4 + (double)b.nlist * d * 4 + 2.0 * nq * b.nlist * 4, st, [&] {
            if (h->cap_chunk <= 0 || !h->cscore.p) RVCMI_FAIL(RVCMI_ERR_INVALID, "coarse score scratch not reserved");
            for (int64_t qs = 0; qs < nq; qs += h->cap_chunk) {
                const int64_t nqc = std::min<int64_t>(h->cap_chunk, nq - qs);
                const int64_t big_blocks = ((b.nlist + 127) / 128) * ((nqc + 63) / 64);
                if (big_blocks >= 512) {
                    dim3 grid((unsigned)((b.nlist + 127) / 128), (unsigned)((nqc + 63) / 64));
                    hipLaunchKernelGGL((k_coarse_gemm<2, 4>), grid, dim3(256), 0, st, q + qs * d, h->centroids(), h->cnorm(), nqc,
                                       b.nlist, d, h->cscore.as<float>());
                } else {
                    dim3 grid((unsigned)((b.nlist + 31) / 32), (unsigned)((nqc + 31) / 32));
                    hipLaunchKernelGGL(k_coarse_gemm_ks, grid, dim3(256), 0, st, q + qs * d, h->centroids(), h->cnorm(), nqc,
                                       b.nlist, d, h->cscore.as<float>());
                }
                hipLaunchKernelGGL(k_coarse_pick, dim3((unsigned)((nqc + 3) / 4)), dim3(256), 0, st, q + qs * d, h->centroids(),
                                   h->cscore.as<float>(), nqc, b.nlist, d, b.cmax, h->assign.as<int64_t>() + qs);
            }
        });
    } else if (np == 1) {
        const size_t smem = align_up((size_t)QT * d * 4, 16) + 4 * QT * 16;
        h->prof.launch("ivf_coarse", cflops, (double)nq * d * 4 + (double)b.nlist * d * 4, st, [&] {
            hipLaunchKernelGGL(k_coarse1, dim3((unsigned)((nq + QT - 1) / QT)), dim3(256), smem, st, q, h->centroids_t(), nq,
                               b.nlist, d, h->assign.as<int64_t>());
        });
    } else {
        h->prof.launch("ivf_coarse", cflops, (double)nq * b.nlist * 8, st, [&] {
            const int64_t tot = nq * b.nlist;
            hipLaunchKernelGGL(k_coarse_dist, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, q, h->centroids(), nq,
                               b.nlist, d, h->cdist.as<double>());
            hipLaunchKernelGGL(k_coarse_select, dim3((unsigned)nq), dim3(64), 0, st, h->cdist.as<double>(), b.nlist, np,
                               h->assign.as<int64_t>());
        });
    }
    // list-major scan: plan (sort + work items) -> fp32 MFMA score tiles -> per-query exact verification (+ fused blend)
    const bool lm_wanted = h->opt.geti("IVF_LM", 1) && !h->opt.on("IVF_GENERIC");
    if (lm_wanted && nq >= 64 && !h->lm_warned) {
        const char* why = lm_unusable_reason(h, nq);
        if (why) {
            h->lm_warned = true;
            fprintf(stderr, "[rvcmi] ivf search: %lld queries run on the query-major scan (every probed list is re-read per query) because the "
                    "list-major kernels do not take this call: %s.  Printed once per index.\n", (long long)nq, why);
        }
    }
    if (lm_wanted && lm_usable(h, nq) && h->lm_S.p && nq <= h->lm_cap_nq && h->lm_buf_nq > 0) {
        const int pitch = (int)align_up((uint64_t)h->lm_maxlen, 32);
        const double rows = (double)b.ntotal / (double)b.nlist;
        // passes of at most lm_buf_nq queries (one, unless queries x longest list x 4 B would exceed the 1 GiB score scratch)
        for (int64_t qs = 0; qs < nq; qs += h->lm_buf_nq) {
            const int64_t nqp = std::min<int64_t>(h->lm_buf_nq, nq - qs);
            const float* qp = q + qs * d;
            const int64_t* asg = h->assign.as<int64_t>() + qs;
            h->prof.launch("ivf_plan", 0.0, (double)nqp * 12 + (double)b.nlist * 8, st, [&] {
                hipLaunchKernelGGL(k_lm_plan, dim3(1), dim3(1024), (size_t)2 * (b.nlist + 1) * 4, st, asg, (int)nqp, (int)b.nlist, h->list_off(),
                                   h->lm_qinfo.as<LmQuery>(), h->lm_items.as<LmItem>(), h->lm_n.as<int>(), (int)std::min<int64_t>(h->lm_cap_items, 0x7fffffff));
            });
            // (flops / bytes: the SURVEY 8d model -- N / nlist rows per query -- as for the query-major kernel, so that the roofline
            //  lines of the two paths compare)
            h->prof.launch("ivf_scan", 2.0 * nqp * rows * d, (double)nqp * rows * (4.0 * d + 8) + (double)nqp * d * 4, st, [&] {
                const int grid = (int)std::min<int64_t>(h->lm_cap_items, (int64_t)num_cus_ivf() * 3);
                hipLaunchKernelGGL(k_lm_gemm, dim3((unsigned)std::max(grid, 1)), dim3(256), 0, st, qp, h->lm_qinfo.as<LmQuery>(), h->vecs(), h->lm_rn.as<float>(),
                                   h->list_off(), h->lm_items.as<LmItem>(), h->lm_n.as<int>(), d, pitch, h->lm_S.as<float>());
            });
            h->prof.launch("ivf_select", 3.0 * nqp * 12 * d, (double)nqp * (12.0 * d * 4 + pitch * 4.0 + d * 8.0), st, [&] {
                auto kern = d == 768 ? &k_lm_select<3> : (d == 256 ? &k_lm_select<1> : &k_lm_select<0>);
                hipLaunchKernelGGL(kern, dim3((unsigned)((nqp + 3) / 4)), dim3(256), (size_t)4 * ((size_t)std::min(pitch, LM_MAXPITCH) * 4 + LM_WAVE_EXTRA), st, qp,
                                   h->lm_qinfo.as<LmQuery>(), h->ids(), h->vecs(), h->lm_S.as<float>(), (int)nqp, (int)b.nlist, d, pitch, h->lm_vmax, k, D + qs * k,
                                   I + qs * k, h->P.as<int64_t>() + qs * k, h->flag.as<int>(), bf ? bf->feats + qs * d : nullptr, bf ? bf->rate : 0.f,
                                   bf ? bf->omr : 0.f, h->hdr.pos_last);
            });
        }
        HIP_CHECK(hipGetLastError());
        return bf != nullptr;
    }
    // list-sorted, XCD-contiguous query order for the specialised scans (see k_qsort_hist)
    const int* perm = nullptr;
    if (np == 1 && nq >= 64 && nq < (1ll << 31) && (d == 768 || d == 256) && !h->opt.on("IVF_GENERIC") && h->opt.geti("IVF_SORT", 0)) {
        h->prof.launch("ivf_sort", 0.0, (double)nq * 16 + (double)b.nlist * 8, st, [&] {
            HIP_CHECK(hipMemsetAsync(h->qcnt.p, 0, (size_t)(b.nlist + 1) * 4, st));
            hipLaunchKernelGGL(k_qsort_hist, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, h->assign.as<int64_t>(), nq, b.nlist, h->qcnt.as<int>());
            hipLaunchKernelGGL(k_qsort_scan, dim3(1), dim3(1024), 0, st, h->qcnt.as<int>(), b.nlist + 1);
            hipLaunchKernelGGL(k_qsort_scatter, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, h->assign.as<int64_t>(), nq, b.nlist, h->qcnt.as<int>(),
                               h->qperm.as<int>());
        });
        perm = h->qperm.as<int>();
    }
    bool fused = false;
    const double rows = b.nlist ? (double)b.ntotal / (double)b.nlist * np : 0;
    const size_t smem = align_up((size_t)d * 4, 16) + SCAN_GROUPS * sizeof(TopK);
    h->prof.launch("ivf_scan", 3.0 * nq * rows * d, (double)nq * rows * (4.0 * d + 8) + (double)nq * d * 4, st, [&] {
        const size_t sm2 = SCAN_GROUPS * sizeof(TopK);
        if (d == 768 && !h->opt.on("IVF_GENERIC")) {
            static unsigned long long* tsd = nullptr;
            const bool want_ts = h->opt.on("IVF_STAMPS");
            if (want_ts && !tsd) HIP_CHECK(hipMalloc(&tsd, 16 * 8));
            if (want_ts) HIP_CHECK(hipMemsetAsync(tsd, 0, 16 * 8, st));
            hipLaunchKernelGGL((k_scan_v<6, 32, 2>), dim3((unsigned)nq), dim3(256), sm2, st, q, h->assign.as<int64_t>(), np, h->list_off(),
                               h->ids(), h->vecs(), nq, k, D, I, h->P.as<int64_t>(), h->flag.as<int>(), bf ? bf->feats : nullptr,
                               bf ? bf->rate : 0.f, bf ? bf->omr : 0.f, h->hdr.pos_last, want_ts ? tsd : nullptr, perm);
            if (want_ts) {
                HIP_CHECK(hipStreamSynchronize(st));
                unsigned long long t[16];
                HIP_CHECK(hipMemcpy(t, tsd, sizeof(t), hipMemcpyDeviceToHost));
                fprintf(stderr, "[ivf stamps] (10 ns units, block 0) assign %llu list_off %llu it0 %llu it1 %llu it2 %llu loop_end %llu barrier %llu merge+blend %llu total %llu\n",
                        t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] ? t[4] - t[3] : 0, t[5] ? t[5] - t[4] : 0, t[6] - (t[5] ? t[5] : (t[4] ? t[4] : t[3])), t[7] - t[6], t[8] - t[7], t[8] - t[0]);
            }
            fused = bf != nullptr;
            return;
        }
        if (d == 256 && !h->opt.on("IVF_GENERIC")) {
            hipLaunchKernelGGL((k_scan_v<4, 16, 2>), dim3((unsigned)nq), dim3(256), sm2, st, q, h->assign.as<int64_t>(), np, h->list_off(),
                               h->ids(), h->vecs(), nq, k, D, I, h->P.as<int64_t>(), h->flag.as<int>(), bf ? bf->feats : nullptr,
                               bf ? bf->rate : 0.f, bf ? bf->omr : 0.f, h->hdr.pos_last, nullptr, perm);
            fused = bf != nullptr;
            return;
        }
        hipLaunchKernelGGL(k_scan, dim3((unsigned)nq), dim3(256), smem, st, q, h->assign.as<int64_t>(), np,
                           h->list_off(), h->ids(), h->vecs(), nq, d, k, D, I, h->P.as<int64_t>(), h->flag.as<int>(),
                           h->opt.geti("IVF_DBG", 0));
    });
    HIP_CHECK(hipGetLastError());
    return fused;
}

}  // namespace rvcmi

extern "C" {

int rvcmi_ivf_create_from_file(const char* path, int device, rvcmi_ivf** out) {
    return guarded([&] {
        if (!path || !out) RVCMI_FAIL(RVCMI_ERR_INVALID, "null argument");
        *out = read_faiss(path, device);
    });
}
int rvcmi_ivf_write_file(const rvcmi_ivf* h, const char* path) {
    return guarded([&] {
        if (!h || !path) RVCMI_FAIL(RVCMI_ERR_INVALID, "null argument");
        DeviceGuard dg(h->device);
        write_faiss(h, path);
    });
}
// shared by rvcmi_ivf_build (centroids_out == nullptr: k-means + add -> *out) and rvcmi_kmeans (centroids only: niter Lloyd updates, no
// final assignment pass, no list-major copy of the vectors, no index object)
static int ivf_build_impl(int d, int64_t n, const float* x_host, int64_t nlist, int niter, uint64_t seed, int device,
                          double* objective_out, float* centroids_out, rvcmi_ivf** out) {
    return guarded([&] {
        if (!x_host || (!out && !centroids_out)) RVCMI_FAIL(RVCMI_ERR_INVALID, "null argument");
        DeviceGuard dg(device);
        validate(d, n, nlist, 1);
        if (n < nlist) RVCMI_FAIL(RVCMI_ERR_INVALID, "need at least nlist=%lld training vectors, got %lld", (long long)nlist, (long long)n);
        if (niter < 0 || niter > 1000) RVCMI_FAIL(RVCMI_ERR_INVALID, "niter out of range");
        hipStream_t st = nullptr;
        DevBuf X, Cd, Cn, Sc, As, Ord, Off, Dist;
        X.alloc((size_t)n * d * 4);
        HIP_CHECK(hipMemcpy(X.p, x_host, (size_t)n * d * 4, hipMemcpyHostToDevice));
        Cd.alloc((size_t)nlist * d * 4);
        Cn.alloc((size_t)nlist * 4);
        As.alloc((size_t)n * 8);
        Ord.alloc((size_t)n * 8);
        Off.alloc((size_t)(nlist + 1) * 8);
        Dist.alloc((size_t)n * 8);
        const int64_t chunk = std::max<int64_t>(64, std::min<int64_t>(n, ((int64_t)512 << 20) / (4 * nlist)));
        Sc.alloc((size_t)chunk * nlist * 4);
        // init: nlist distinct training vectors (seeded partial Fisher-Yates), like faiss' random subset initialisation
        std::vector<float> cent((size_t)nlist * d);
        {
            std::vector<int64_t> perm(n);
            for (int64_t i = 0; i < n; ++i) perm[i] = i;
            uint64_t sd = seed * 6364136223846793005ULL + 1442695040888963407ULL;
            for (int64_t i = 0; i < nlist; ++i) {
                sd ^= sd >> 12; sd ^= sd << 25; sd ^= sd >> 27;  // xorshift64*
                const uint64_t r = (sd * 2685821657736338717ULL) % (uint64_t)(n - i);
                std::swap(perm[i], perm[i + (int64_t)r]);
                memcpy(&cent[(size_t)i * d], x_host + (size_t)perm[i] * d, (size_t)d * 4);
            }
        }
        std::vector<int64_t> assign(n), order(n), off(nlist + 1);
        std::vector<float> cn(nlist);
        std::vector<double> dist(n);
        for (int it = 0; it <= niter; ++it) {
            if (centroids_out && it == niter) break;  // centres only: the assignment after the last update is not needed
            double cmax2 = 0.0;
            for (int64_t c = 0; c < nlist; ++c) {
                double n2 = 0.0;
                for (int e = 0; e < d; ++e) n2 += (double)cent[c * d + e] * (double)cent[c * d + e];
                cn[c] = (float)n2;
                cmax2 = std::max(cmax2, n2);
            }
            HIP_CHECK(hipMemcpyAsync(Cd.p, cent.data(), cent.size() * 4, hipMemcpyHostToDevice, st));
            HIP_CHECK(hipMemcpyAsync(Cn.p, cn.data(), cn.size() * 4, hipMemcpyHostToDevice, st));
            // exact (fp64-verified) nearest centroid of every training vector: the search path's own coarse kernels
            for (int64_t qs = 0; qs < n; qs += chunk) {
                const int64_t nqc = std::min<int64_t>(chunk, n - qs);
                const float* q = X.as<float>() + qs * d;
                if (((nlist + 127) / 128) * ((nqc + 63) / 64) >= 512) {
                    dim3 grid((unsigned)((nlist + 127) / 128), (unsigned)((nqc + 63) / 64));
                    hipLaunchKernelGGL((k_coarse_gemm<2, 4>), grid, dim3(256), 0, st, q, Cd.as<float>(), Cn.as<float>(), nqc, nlist, d, Sc.as<float>());
                } else {
                    dim3 grid((unsigned)((nlist + 31) / 32), (unsigned)((nqc + 31) / 32));
                    hipLaunchKernelGGL((k_coarse_gemm<1, 1>), grid, dim3(64), 0, st, q, Cd.as<float>(), Cn.as<float>(), nqc, nlist, d, Sc.as<float>());
                }
                hipLaunchKernelGGL(k_coarse_pick, dim3((unsigned)((nqc + 3) / 4)), dim3(256), 0, st, q, Cd.as<float>(), Sc.as<float>(), nqc,
                                   nlist, d, std::sqrt(cmax2), As.as<int64_t>() + qs);
            }
            HIP_CHECK(hipGetLastError());
            if (objective_out) {
                hipLaunchKernelGGL(k_assigned_dist, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, X.as<float>(), Cd.as<float>(),
                                   As.as<int64_t>(), n, d, Dist.as<double>());
                HIP_CHECK(hipMemcpyAsync(dist.data(), Dist.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
            }
            HIP_CHECK(hipMemcpyAsync(assign.data(), As.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            if (objective_out) {
                double o = 0.0;
                for (int64_t i = 0; i < n; ++i) o += dist[i];
                objective_out[it] = o;
            }
            // lists: stable counting sort by centroid => ids ascending inside a list (what sequential index.add produces)
            std::fill(off.begin(), off.end(), 0);
            for (int64_t i = 0; i < n; ++i) off[assign[i] + 1]++;
            for (int64_t l = 0; l < nlist; ++l) off[l + 1] += off[l];
            {
                std::vector<int64_t> cur(off.begin(), off.end() - 1);
                for (int64_t i = 0; i < n; ++i) order[cur[assign[i]]++] = i;
            }
            HIP_CHECK(hipMemcpyAsync(Ord.p, order.data(), (size_t)n * 8, hipMemcpyHostToDevice, st));
            HIP_CHECK(hipMemcpyAsync(Off.p, off.data(), (size_t)(nlist + 1) * 8, hipMemcpyHostToDevice, st));
            if (it == niter) break;
            hipLaunchKernelGGL(k_list_mean, dim3((unsigned)nlist), dim3(256), 0, st, X.as<float>(), Ord.as<int64_t>(), Off.as<int64_t>(), d,
                               Cd.as<float>());
            HIP_CHECK(hipMemcpyAsync(cent.data(), Cd.p, cent.size() * 4, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            // empty lists: split the currently largest one (faiss' split_clusters idea, deterministic choice): copy its
            // centroid and nudge the two copies apart by 1/1024 in alternating coordinates
            std::vector<int64_t> sz(nlist);
            for (int64_t l = 0; l < nlist; ++l) sz[l] = off[l + 1] - off[l];
            for (int64_t l = 0; l < nlist; ++l) {
                if (sz[l]) continue;
                const int64_t big = std::max_element(sz.begin(), sz.end()) - sz.begin();
                for (int e = 0; e < d; ++e) {
                    const float v = cent[big * d + e];
                    const float eps = 1.f / 1024.f;
                    cent[l * d + e] = (e & 1) ? v * (1.f - eps) : v * (1.f + eps);
                    cent[big * d + e] = (e & 1) ? v * (1.f + eps) : v * (1.f - eps);
                }
                sz[l] = sz[big] / 2;
                sz[big] -= sz[l];
            }
            // Relocation (round 5).  Lloyd iterations from a random start never repair "two centres inside one natural cluster, none in
            // another": against the reference's MiniBatchKMeans call (web.py:522-536; it re-seeds low-count centres every batch) they ended
            // 1.43x above its objective on well-separated blobs.  After the update, while settling iterations remain: the centre whose
            // deletion costs least -- every point of cluster j re-assigned to j's nearest other centre i costs at most n_j |c_j - c_i|^2 --
            // moves to the farthest point p of the cluster with the largest distortion, if the EXACT gain of a centre at p for that
            // cluster's own points, sum max(0, |x - c_o|^2 - |x - p|^2), exceeds that cost: the objective of the next assignment cannot
            // go up (the build's objective stays non-increasing).  Deterministic; at most nlist / 20 moves per iteration; on rows without
            // cluster structure no move passes the test and the iterations are the plain ones.
            if (it + 3 < niter && nlist >= 2) {
                hipLaunchKernelGGL(k_assigned_dist, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, X.as<float>(), Cd.as<float>(), As.as<int64_t>(), n, d,
                                   Dist.as<double>());  // distance of every row to ITS new mean
                HIP_CHECK(hipMemcpyAsync(dist.data(), Dist.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
                // nearest OTHER centre of every centre: the two nearest centres of the new means (fp64, the nprobe > 1 kernels)
                const int64_t cch = std::max<int64_t>(1, std::min<int64_t>(nlist, ((int64_t)256 << 20) / (8 * nlist)));
                DevBuf CC, NN;
                CC.alloc((size_t)cch * nlist * 8);
                NN.alloc((size_t)nlist * 2 * 8);
                for (int64_t cs = 0; cs < nlist; cs += cch) {
                    const int64_t nc = std::min<int64_t>(cch, nlist - cs);
                    hipLaunchKernelGGL(k_coarse_dist, dim3((unsigned)((nc * nlist + 255) / 256)), dim3(256), 0, st, Cd.as<float>() + cs * d, Cd.as<float>(), nc,
                                       nlist, d, CC.as<double>());
                    hipLaunchKernelGGL(k_coarse_select, dim3((unsigned)nc), dim3(64), 0, st, CC.as<double>(), nlist, 2, NN.as<int64_t>() + cs * 2);
                }
                HIP_CHECK(hipGetLastError());
                std::vector<int64_t> nn2((size_t)nlist * 2);
                HIP_CHECK(hipMemcpyAsync(nn2.data(), NN.p, nn2.size() * 8, hipMemcpyDeviceToHost, st));
                HIP_CHECK(hipStreamSynchronize(st));
                std::vector<double> S(nlist, 0.0), cost(nlist, INFINITY);
                std::vector<int64_t> nn(nlist, -1);
                for (int64_t l = 0; l < nlist; ++l)
                    for (int64_t p = off[l]; p < off[l + 1]; ++p) S[l] += dist[order[p]];
                for (int64_t l = 0; l < nlist; ++l) {
                    const int64_t cnt = off[l + 1] - off[l];
                    if (!cnt) continue;  // (an empty list was just re-seeded above)
                    int64_t o = nn2[l * 2] == l ? nn2[l * 2 + 1] : nn2[l * 2];
                    if (o < 0 || o == l || off[o + 1] == off[o]) continue;
                    double dn = 0.0;
                    for (int e = 0; e < d; ++e) {
                        const double t = (double)cent[l * d + e] - (double)cent[o * d + e];
                        dn += t * t;
                    }
                    nn[l] = o;
                    cost[l] = (double)cnt * dn;
                }
                std::vector<int64_t> by_cost(nlist), by_S(nlist);
                for (int64_t l = 0; l < nlist; ++l) by_cost[l] = by_S[l] = l;
                std::sort(by_cost.begin(), by_cost.end(), [&](int64_t a1, int64_t b1) { return cost[a1] < cost[b1] || (cost[a1] == cost[b1] && a1 < b1); });
                std::sort(by_S.begin(), by_S.end(), [&](int64_t a1, int64_t b1) { return S[a1] > S[b1] || (S[a1] == S[b1] && a1 < b1); });
                // used[]: centres that took part in a move of this iteration (the moved centre, the split cluster, the receiving centre) -- none
                // of them is deleted or split again.  moved[]: centres that are GONE from their old place: a later candidate whose nearest centre
                // was moved has a stale cost (n_j |c_j - c_nn|^2 against a centre that is no longer there) and is skipped, so every accepted
                // move still satisfies gain > cost and the objective cannot go up.
                std::vector<char> used(nlist, 0), moved(nlist, 0);
                const int64_t maxmoves = std::max<int64_t>(1, nlist / 20);
                int64_t moves = 0, si = 0;
                for (int64_t ci = 0; ci < nlist && moves < maxmoves; ++ci) {
                    const int64_t j = by_cost[ci];
                    if (used[j] || nn[j] < 0 || !(cost[j] < INFINITY) || moved[nn[j]]) continue;
                    // the global cursor passes only clusters that are out for EVERY later candidate (used, or too small to split); the clusters
                    // excluded for this candidate alone (j itself, its receiver) are stepped over by the local cursor
                    while (si < nlist && (used[by_S[si]] || off[by_S[si] + 1] - off[by_S[si]] < 2)) ++si;
                    int64_t sj = si;
                    while (sj < nlist && (used[by_S[sj]] || by_S[sj] == j || by_S[sj] == nn[j] || off[by_S[sj] + 1] - off[by_S[sj]] < 2)) ++sj;
                    if (sj >= nlist) break;
                    const int64_t o = by_S[sj];
                    int64_t far = order[off[o]];
                    for (int64_t p = off[o]; p < off[o + 1]; ++p)
                        if (dist[order[p]] > dist[far]) far = order[p];
                    const float* xp = x_host + (size_t)far * d;
                    double gain = 0.0;
                    for (int64_t p = off[o]; p < off[o + 1]; ++p) {
                        const float* xi = x_host + (size_t)order[p] * d;
                        double dp = 0.0;
                        for (int e = 0; e < d; ++e) {
                            const double t = (double)xi[e] - (double)xp[e];
                            dp += t * t;
                        }
                        gain += std::max(0.0, dist[order[p]] - dp);
                    }
                    if (!(gain > cost[j])) break;  // the cheapest deletion no longer pays for the best split: done for this iteration
                    memcpy(&cent[(size_t)j * d], xp, (size_t)d * 4);
                    used[j] = used[o] = used[nn[j]] = 1;
                    moved[j] = 1;
                    ++moves;
                }
            }
        }
        if (centroids_out) {  // every centre is valid: a cluster that lost all its points was re-seeded by splitting the largest one
            memcpy(centroids_out, cent.data(), cent.size() * 4);
            return;
        }
        // index.add: list-major copy of the vectors, then the packed blob (same layout the reader produces)
        DevBuf V;
        V.alloc((size_t)std::max<int64_t>(n, 1) * d * 4);
        hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)n), dim3(256), 0, st, X.as<float>(), Ord.as<int64_t>(), n, d, V.as<float>());
        HIP_CHECK(hipGetLastError());
        std::vector<float> vecs((size_t)n * d);
        HIP_CHECK(hipMemcpy(vecs.data(), V.p, vecs.size() * 4, hipMemcpyDeviceToHost));
        auto blob = build_blob(d, n, nlist, 1, cent.data(), off.data(), order.data(), vecs.data());
        *out = from_host_blob(blob, device);
    });
}
int rvcmi_ivf_build(int d, int64_t n, const float* x_host, int64_t nlist, int niter, uint64_t seed, int device,
                    double* objective_out, rvcmi_ivf** out) {
    if (!out) return rvcmi::guarded([&] { RVCMI_FAIL(RVCMI_ERR_INVALID, "null argument"); });
    return ivf_build_impl(d, n, x_host, nlist, niter, seed, device, objective_out, nullptr, out);
}
int rvcmi_kmeans(int d, int64_t n, const float* x_host, int64_t k, int niter, uint64_t seed, int device, double* objective_out,
                 float* centroids_out_host) {
    if (!centroids_out_host) return rvcmi::guarded([&] { RVCMI_FAIL(RVCMI_ERR_INVALID, "null argument"); });
    return ivf_build_impl(d, n, x_host, k, niter, seed, device, objective_out, centroids_out_host, nullptr);
}
int rvcmi_ivf_create(int d, int64_t n, int64_t nlist, int nprobe, const float* centroids, const int64_t* list_offsets,
                     const int64_t* ids, const float* vecs, int device, rvcmi_ivf** out) {
    return guarded([&] {
        if (!centroids || !list_offsets || !out || (n && (!ids || !vecs))) RVCMI_FAIL(RVCMI_ERR_INVALID, "null argument");
        auto blob = build_blob(d, n, nlist, nprobe, centroids, list_offsets, ids, vecs);
        *out = from_host_blob(blob, device);
    });
}
int rvcmi_ivf_destroy(rvcmi_ivf* h) {
    return guarded([&] { delete h; });
}
int rvcmi_ivf_d(const rvcmi_ivf* h) { return h ? h->hdr.d : RVCMI_ERR_INVALID; }
int64_t rvcmi_ivf_ntotal(const rvcmi_ivf* h) { return h ? h->hdr.ntotal : RVCMI_ERR_INVALID; }
int64_t rvcmi_ivf_nlist(const rvcmi_ivf* h) { return h ? h->hdr.nlist : RVCMI_ERR_INVALID; }
int rvcmi_ivf_nprobe(const rvcmi_ivf* h) { return h ? h->hdr.nprobe : RVCMI_ERR_INVALID; }
int rvcmi_ivf_set_nprobe(rvcmi_ivf* h, int nprobe) {
    return guarded([&] {
        if (!h || nprobe < 1) RVCMI_FAIL(RVCMI_ERR_INVALID, "bad nprobe");
        h->hdr.nprobe = nprobe;
        DeviceGuard dg(h->device);
        HIP_CHECK(hipMemcpy(h->blob, &h->hdr, sizeof(BlobHeader), hipMemcpyHostToDevice));
    });
}
int rvcmi_ivf_reserve(rvcmi_ivf* h, int64_t max_nq) {
    return guarded([&] {
        if (!h || max_nq < 0) RVCMI_FAIL(RVCMI_ERR_INVALID, "bad argument");
        reserve(h, max_nq);
    });
}
int rvcmi_ivf_search(rvcmi_ivf* h, int64_t nq, const float* q, int k, float* D, int64_t* I, void* stream) {
    return guarded([&] { search(h, nq, q, k, D, I, (hipStream_t)stream); });
}
int rvcmi_ivf_search_blend(rvcmi_ivf* h, int64_t nq, float* feats, float index_rate, int k, int skip_if_short, void* stream) {
    return guarded([&] {
        if (!h || (nq && !feats)) RVCMI_FAIL(RVCMI_ERR_INVALID, "null argument");
        if (nq == 0) return;
        hipStream_t st = (hipStream_t)stream;
        reserve(h, nq);
        const int d = h->hdr.d;
        // torch evaluates `npy * index_rate + (1 - index_rate) * feats` with the python scalars cast to fp32
        const float rate = index_rate, omr = (float)(1.0 - (double)index_rate);
        const BlendFuse bf = {feats, rate, omr};
        // without the realtime guard the blend of a row needs nothing but that row's own results: done by the scan block
        if (search(h, nq, feats, k, h->Dtmp.as<float>(), h->Itmp.as<int64_t>(), st, skip_if_short ? nullptr : &bf)) return;
        h->prof.launch("ivf_blend", 2.0 * nq * k * d, (double)nq * d * 4 * (2 + k), st, [&] {
            hipLaunchKernelGGL(k_blend, dim3((unsigned)nq), dim3(256), 0, st, feats, h->Dtmp.as<float>(), h->P.as<int64_t>(),
                               h->vecs(), d, k, h->hdr.pos_last, rate, omr, h->flag.as<int>(), skip_if_short);
        });
        HIP_CHECK(hipGetLastError());
    });
}
int rvcmi_ivf_search_blend_expand(rvcmi_ivf* h, int64_t nq, const float* feats, float index_rate, int k, int skip_if_short,
                                  const float* pitchf, float protect, int64_t p_len, float* out, void* stream) {
    return guarded([&] {
        if (!h || (nq && (!feats || !out))) RVCMI_FAIL(RVCMI_ERR_INVALID, "null argument");
        if (p_len < 0 || p_len > 2 * nq) RVCMI_FAIL(RVCMI_ERR_INVALID, "p_len %lld outside [0, 2*nq]", (long long)p_len);
        if (nq == 0 || p_len == 0) return;
        hipStream_t st = (hipStream_t)stream;
        reserve(h, nq);
        search(h, nq, feats, k, h->Dtmp.as<float>(), h->Itmp.as<int64_t>(), st);
        const int d = h->hdr.d;
        const float rate = index_rate, omr = (float)(1.0 - (double)index_rate);
        h->prof.launch("ivf_blend_x2", 2.0 * nq * k * d, (double)nq * d * 4 * (1 + k) + (double)p_len * d * 4, st, [&] {
            hipLaunchKernelGGL(k_blend_expand, dim3((unsigned)nq), dim3(256), 0, st, feats, h->Dtmp.as<float>(), h->P.as<int64_t>(),
                               h->vecs(), d, k, h->hdr.pos_last, rate, omr, h->flag.as<int>(), skip_if_short, pitchf, protect,
                               p_len, 2, out);
        });
        HIP_CHECK(hipGetLastError());
    });
}
int rvcmi_ivf_reconstruct_n(const rvcmi_ivf* h, int64_t i0, int64_t n, float* out_host) {
    return guarded([&] {
        if (!h || !out_host) RVCMI_FAIL(RVCMI_ERR_INVALID, "null argument");
        const BlobHeader& b = h->hdr;
        if (i0 < 0 || n < 0 || i0 + n > b.ntotal) RVCMI_FAIL(RVCMI_ERR_INVALID, "range [%lld,%lld) outside ntotal %lld",
                                                                 (long long)i0, (long long)(i0 + n), (long long)b.ntotal);
        if (!n) return;
        DeviceGuard dg(h->device);
        std::vector<int64_t> ids(b.ntotal);
        std::vector<float> vecs((size_t)b.ntotal * b.d);
        HIP_CHECK(hipMemcpy(ids.data(), h->blob + b.off_ids, ids.size() * 8, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(vecs.data(), h->blob + b.off_vecs, vecs.size() * 4, hipMemcpyDeviceToHost));
        // faiss' reconstruct_n needs a direct map and raises for ids it cannot find; here rows the index does not hold (ids
        // that are not a permutation of 0..ntotal-1, e.g. after add_with_ids) are an error too, never uninitialised memory
        std::vector<char> seen((size_t)n, 0);
        for (int64_t p = 0; p < b.ntotal; ++p) {
            const int64_t id = ids[p];
            if (id >= i0 && id < i0 + n) {
                memcpy(out_host + (size_t)(id - i0) * b.d, vecs.data() + (size_t)p * b.d, (size_t)b.d * 4);
                seen[(size_t)(id - i0)] = 1;
            }
        }
        for (int64_t i = 0; i < n; ++i)
            if (!seen[(size_t)i]) RVCMI_FAIL(RVCMI_ERR_INVALID, "reconstruct_n: id %lld is not in the index (ids are not sequential)", (long long)(i0 + i));
    });
}
int rvcmi_ivf_blob(const rvcmi_ivf* h, void** dev_ptr, size_t* bytes) {
    return guarded([&] {
        if (!h || !dev_ptr || !bytes) RVCMI_FAIL(RVCMI_ERR_INVALID, "null argument");
        *dev_ptr = h->blob;
        *bytes = h->hdr.total_bytes;
    });
}
int rvcmi_ivf_centroids(const rvcmi_ivf* h, float* out_host) {
    return guarded([&] {
        if (!h || !out_host) RVCMI_FAIL(RVCMI_ERR_INVALID, "null argument");
        DeviceGuard dg(h->device);
        HIP_CHECK(hipMemcpy(out_host, h->centroids(), (size_t)h->hdr.nlist * h->hdr.d * sizeof(float), hipMemcpyDeviceToHost));
    });
}
int rvcmi_ivf_blob_copy(const rvcmi_ivf* h, void* dst_dev, size_t capacity, void* stream) {
    return guarded([&] {
        if (!h || !dst_dev) RVCMI_FAIL(RVCMI_ERR_INVALID, "null argument");
        if (capacity < h->hdr.total_bytes) RVCMI_FAIL(RVCMI_ERR_INVALID, "blob_copy: destination holds %zu bytes, the index needs %zu", capacity, (size_t)h->hdr.total_bytes);
        DeviceGuard dg(h->device);
        HIP_CHECK(hipMemcpyAsync(dst_dev, h->blob, h->hdr.total_bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    });
}
int rvcmi_ivf_create_from_blob(void* dev_ptr, size_t bytes, int device, int take_ownership, rvcmi_ivf** out) {
    return guarded([&] {
        if (!dev_ptr || !out || bytes < sizeof(BlobHeader)) RVCMI_FAIL(RVCMI_ERR_INVALID, "bad blob");
        DeviceGuard dg(device);
        std::unique_ptr<rvcmi_ivf> h(new rvcmi_ivf());
        h->device = device;
        HIP_CHECK(hipMemcpy(&h->hdr, dev_ptr, sizeof(BlobHeader), hipMemcpyDeviceToHost));
        if (h->hdr.magic != kMagic || h->hdr.version != 1 || h->hdr.total_bytes != bytes)
            RVCMI_FAIL(RVCMI_ERR_INVALID, "not an rvcmi IVF blob (magic/version/size mismatch)");
        validate(h->hdr.d, h->hdr.ntotal, h->hdr.nlist, h->hdr.nprobe);
        h->blob = (char*)dev_ptr;
        h->owns_blob = take_ownership != 0;
        *out = h.release();
    });
}
int rvcmi_ivf_set_option(rvcmi_ivf* h, const char* key, double value) {
    return guarded([&] {
        if (!h || !key) RVCMI_FAIL(RVCMI_ERR_INVALID, "null argument");
        if (!h->opt.set(key, value)) RVCMI_FAIL(RVCMI_ERR_INVALID, "unknown option '%s' for this handle", key);
    });
}
int rvcmi_ivf_profile_enable(rvcmi_ivf* h, int enable) {
    return guarded([&] {
        if (!h) RVCMI_FAIL(RVCMI_ERR_INVALID, "null handle");
        h->prof.enabled = enable != 0;
    });
}
int rvcmi_ivf_profile_read(rvcmi_ivf* h, rvcmi_kernel_stat* stats, int capacity, int* n, int reset) {
    return guarded([&] {
        if (!h) RVCMI_FAIL(RVCMI_ERR_INVALID, "null handle");
        h->prof.read(stats, capacity, n, reset);
    });
}

}  // extern "C"
