// List-major IVF-Flat scan for nprobe == 1 (index.search(npy, k=8), infer/modules/vc/pipeline.py:126): the rows of a probed
// list are read ONCE per tile of 32 queries instead of once per query.
//
// The query-major scan (k_scan_v: one block per query) walks 568 MB of list rows through the L2s for the 16.5 MB the benchmark
// clip's 599 queries actually probe (34x), and evaluates every (query, row) pair in fp64.  Here:
//
//   k_lm_plan    one block: counting sort of the queries by probed list (qinfo: query, list range per sorted position), then one
//                work item per (list, 32 sorted queries, 32 rows) tile;
//   k_lm_gemm    per item: S[q][r] = fl32(|v_r|^2) - 2 dot32(q, v_r) on the fp32 MFMA (v_mfma_f32_32x32x2_f32: an fmaf chain, so
//                the classic bound |S_hat - S| <= (2 d + 4) u (|q| vmax + vmax^2) holds -- the same prefilter as the coarse
//                quantiser, k_coarse_gemm_ks); K split over the block's four waves;
//   k_lm_select  one wave per query: t8 = 8th smallest S of its list, every row with S <= t8 + 2 E is re-evaluated as
//                sum((q - v)^2) in fp64 (exactly the arithmetic of the query-major kernels' definition: exact differences,
//                fp64 accumulation) and the exact (distance, id) top-k wins; the blend of pipeline.py:129-138 runs in the same wave.
//
// Why the result IS the exact top-k: at least k rows have S_hat <= t8, hence true distance (minus |q|^2) <= t8 + E, so the true
// k-th best is <= t8 + E; a true top-k row therefore has S_hat <= true + E <= t8 + 2 E and is among the verified candidates.
#pragma once

namespace rvcmi {

struct LmItem {  // 16 bytes: one store / one load
    int list;  // probed list
    int q0;    // first sorted query position of the tile
    int r0;    // first row of the tile inside the list
    int nqr;   // queries in the tile (1..32) | rows in the tile (1..32) << 8
    __host__ __device__ int nq() const { return nqr & 0xff; }
    __host__ __device__ int nr() const { return nqr >> 8; }
};
static_assert(sizeof(LmItem) == 16, "LmItem is stored and loaded as one 16-byte word");

constexpr int LM_MAXL = 16384;     // lists the one-block planner counts in LDS (2 x 4 B each)
constexpr int LM_MAXPITCH = 2048;  // longest score row the selector stages in LDS (4 waves x 8 KB); longer lists (round 5) are worked on in place
                                   // in the query's own row of the global score scratch (L2-hot: the tile kernel just wrote it)

// |v|^2 of every stored row, fp64 accumulation rounded to fp32 (as the centroid norms of the coarse prefilter); one wave per row
__global__ void __launch_bounds__(256) k_lm_row_norms(const float* __restrict__ vecs, int64_t n, int d, float* __restrict__ rn) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= n) return;
    const float* v = vecs + r * d;
    double acc = 0.0;
    for (int e = lane; e < d; e += 64) acc = fma((double)v[e], (double)v[e], acc);
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) rn[r] = (float)acc;
}
// out[0] = max row norm^2 (as float bits, non-negative => integer order), out[1] = longest list
__global__ void __launch_bounds__(256) k_lm_stats(const float* __restrict__ rn, int64_t n, const int64_t* __restrict__ list_off, int64_t nlist,
                                                  unsigned* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) atomicMax(&out[0], __float_as_uint(rn[i]));
    if (i < nlist) atomicMax(&out[1], (unsigned)min((int64_t)0x7fffffff, list_off[i + 1] - list_off[i]));
}

// One block of 1024 threads.  LDS (dynamic): cnt[nlist + 1] | off[nlist + 1].
struct LmQuery {  // per sorted position: everything k_lm_select needs to start (one 16-byte load instead of three dependent ones)
    int qi;   // query index
    int len;  // rows of its list (0: no list / empty list)
    long long beg;  // first row of its list
};
constexpr int LM_PER = (LM_MAXL + 1 + 1023) / 1024;  // lists per planner thread

__global__ void __launch_bounds__(1024) k_lm_plan(const int64_t* __restrict__ assign, int nq, int nlist, const int64_t* __restrict__ list_off,
                                                  LmQuery* __restrict__ qinfo, LmItem* __restrict__ items, int* __restrict__ nitems, int max_items) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ int part[1024];
    int* cnt = (int*)smem_raw;
    int* off = cnt + (nlist + 1);
    const int n1 = nlist + 1;  // bucket nlist: queries without a list (assign < 0)
#ifdef LM_PLAN_STAMPS
    unsigned long long pst[8]; int psn = 0;
#define LM_PST() do { __syncthreads(); if (psn < 8) pst[psn++] = __builtin_readcyclecounter(); } while (0)
#else
#define LM_PST() do {} while (0)
#endif
    LM_PST();
    const int per = (n1 + 1023) / 1024;
    const int b = min(n1, (int)threadIdx.x * per), e = min(n1, b + per);
    // the lengths of this thread's lists, requested before anything else (their latency runs under the counting pass)
    int lenr[LM_PER];
#pragma unroll
    for (int j = 0; j < LM_PER; ++j) {  // (unconditional, clamped loads: a `j < per ? load : 0` is a branch with its own wait per j -- 2 x 16 dependent
        const int l = min(b + j, nlist - 1);  //  round trips at nlist = 16000 -- and the select on the VALUE costs nothing)
        const int64_t o0 = list_off[l], o1 = list_off[l + 1];
        lenr[j] = j < per ? (int)(o1 - o0) : 0;
    }
    for (int i = threadIdx.x; i < n1; i += 1024) cnt[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < nq; i += 1024) {
        const int64_t l = assign[i];
        atomicAdd(&cnt[(l < 0 || l >= nlist) ? nlist : (int)l], 1);
    }
    __syncthreads();
    LM_PST();  // 1: counted
    // exclusive scan of cnt -> off (thread t owns the contiguous chunk [b, e))
    auto block_scan = [&](int s) {  // exclusive prefix of s over the 1024 threads: wave scans + the 16 wave totals (two barriers)
        const int ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
        int inc = s;
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(inc, o, 64);
            if (ln >= o) inc += v;
        }
        if (ln == 63) part[wv] = inc;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < 16; ++w) base += w < wv ? part[w] : 0;
        __syncthreads();  // part is reused by the next scan
        return base + inc - s;
    };
    {
        int s = 0;
        for (int i = b; i < e; ++i) s += cnt[i];
        int run = block_scan(s);
        for (int i = b; i < e; ++i) {
            off[i] = run;
            run += cnt[i];
        }
    }
    __syncthreads();
    LM_PST();  // 2: scanned
    for (int i = threadIdx.x; i < nq; i += 1024) {  // scatter (order inside a list: whatever the atomics give; results do not depend on it)
        const int64_t l = assign[i];
        const bool has = l >= 0 && l < nlist;
        const int64_t lb = has ? list_off[l] : 0, le = has ? list_off[l + 1] : 0;
        qinfo[atomicAdd(&off[has ? (int)l : nlist], 1)] = LmQuery{i, (int)(le - lb), (long long)lb};
    }
    __syncthreads();  // off[l] is now the END of list l's range in the sorted order
    LM_PST();  // 3: scattered
    // work items: list l with cnt queries and len rows -> ceil(cnt / 32) x ceil(len / 32) tiles.  A thread knows the item range of each
    // of its lists; the EMISSION is wave-cooperative -- for every list with items (ballot over the wave's lanes) its parameters are
    // broadcast from the owning lane and the 64 lanes write the items side by side.  (One thread writing its lists' items alone was 13k of
    // the planner's 20k cycles for one clip -- 39 serial single-lane stores for the longest list -- and 3.2M cycles for 64 clips.)
    int nt[LM_PER], st0[LM_PER];
    int s = 0;
#pragma unroll
    for (int j = 0; j < LM_PER; ++j) {
        nt[j] = (b + j < e && b + j < nlist && lenr[j] > 0) ? ((cnt[b + j] + 31) / 32) * ((lenr[j] + 31) / 32) : 0;
        s += nt[j];
    }
    int run = block_scan(s);
#pragma unroll
    for (int j = 0; j < LM_PER; ++j) {
        st0[j] = run;
        run += nt[j];
    }
    const int ln = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < LM_PER; ++j) {
        if (j >= per) break;  // (block-uniform)
        const int lj = min(b + j, nlist - 1);
        const int cj = cnt[lj], q0j = off[lj] - cj;
        unsigned long long mask = __ballot(nt[j] > 0);
        while (mask) {
            const int k = __builtin_ctzll(mask);
            mask &= mask - 1;
            const int l = __builtin_amdgcn_readlane(lj, k), c = __builtin_amdgcn_readlane(cj, k), q0 = __builtin_amdgcn_readlane(q0j, k);
            const int len = __builtin_amdgcn_readlane(lenr[j], k), n = __builtin_amdgcn_readlane(nt[j], k), base = __builtin_amdgcn_readlane(st0[j], k);
            const int nrt = (len + 31) / 32;
            for (int i = ln; i < n; i += 64) {
                const int qt = i / nrt, rt = i - qt * nrt;
                if (base + i < max_items) items[base + i] = LmItem{l, q0 + qt * 32, rt * 32, min(32, c - qt * 32) | (min(32, len - rt * 32) << 8)};
            }
        }
    }
    LM_PST();  // 4: items
#ifdef LM_PLAN_STAMPS
    if (threadIdx.x == 0) printf("[lm plan] nq %d nlist %d: count %llu scan %llu scatter %llu items %llu\n", nq, nlist, pst[1] - pst[0], pst[2] - pst[1], pst[3] - pst[2], pst[4] - pst[3]);
#endif
    if (threadIdx.x == 1023) *nitems = min(run, max_items);  // (the host sizes `items` for the worst case: never truncated)
}

// S tile of one work item: 32 sorted queries x 32 rows of one list, K split over the 4 waves (k_coarse_gemm_ks's schedule).
// Persistent: block i takes items i, i + gridDim.x, ...
__global__ void __launch_bounds__(256) k_lm_gemm(const float* q, const LmQuery* __restrict__ qinfo, const float* __restrict__ vecs,
                                                 const float* __restrict__ rn, const int64_t* __restrict__ list_off,
                                                 const LmItem* __restrict__ items, const int* __restrict__ nitems, int d, int pitch,
                                                 float* __restrict__ S) {
    __shared__ float St[4][64 * CG_S];  // per wave: 32 query rows then 32 list rows
    __shared__ float Red[3][64 * 16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = *nitems;
    for (int it = blockIdx.x; it < n; it += gridDim.x) {
        const LmItem I = items[it];
        const int64_t rbase = list_off[I.list] + I.r0;
        const int Inq = I.nq(), Inr = I.nr();
        float* st = St[wave];
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        // this lane's 8 staging rows (fixed over the K loop): rows 0..31 = queries (gathered through perm), 32..63 = list rows
        const float* src[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int row = (lane + s * 64) >> 3;
            src[s] = row < 32 ? q + (int64_t)qinfo[I.q0 + min(row, Inq - 1)].qi * d : vecs + (rbase + min(row - 32, Inr - 1)) * d;
        }
        if (d == 768) ks_wave_tile<6, 4>(acc, src, st, wave, lane);       // (ivf.hip: deep prefetch ring over this wave's K chunks)
        else if (d == 256) ks_wave_tile<2, 2>(acc, src, st, wave, lane);
        else {
            float4 pre[8];
            const int kstep = 4 * CG_K;
            for (int k0 = wave * CG_K; k0 < d; k0 += kstep) {
#pragma unroll
                for (int s2 = 0; s2 < 8; ++s2) pre[s2] = *(const float4*)(src[s2] + k0 + ((lane + s2 * 64) & 7) * 4);
#pragma unroll
                for (int s2 = 0; s2 < 8; ++s2) {
                    const int idx = lane + s2 * 64;
                    float* dst = st + (idx >> 3) * CG_S + (idx & 7) * 4;
                    dst[0] = pre[s2].x; dst[1] = pre[s2].y; dst[2] = pre[s2].z; dst[3] = pre[s2].w;
                }
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
            const int c = lane & 31;
            if (c < Inr) {
                const float rnc = rn[rbase + c];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float dot = ((acc[r] + Red[0][r * 64 + lane]) + Red[1][r * 64 + lane]) + Red[2][r * 64 + lane];
                    const int qr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (qr < Inq) S[(int64_t)(I.q0 + qr) * pitch + I.r0 + c] = rnc - 2.f * dot;
                }
            }
        }
        __syncthreads();  // Red is rewritten by the next item
    }
}

// One wave per query (sorted position p), 4 per block.  q / bfeats are not __restrict__: the fused blend runs in place.
// NV = float4 chunks of a row per lane (d = 256 NV: 3 for the 768-d v2 index, 1 for the 256-d v1 index): every loop over the
// dimension is unrolled, so the loads of a phase -- the query slice, eight candidate rows, the eight gathered rows of the blend --
// are all in flight together (with runtime trip counts each chunk was its own dependent round trip: 12 of them in the blend alone).
// NV = 0: any d (runtime loops).
// LDS per wave: the score row sr[pitch] (later the compacted candidate list), then LM_VCAP verified candidates (distance, id, row).
constexpr int LM_VCAP = 64;
constexpr int LM_WAVE_EXTRA = LM_VCAP * (8 + 8 + 4) + 2 * KMAX * 8;  // cd | ci | cr | bd8 (as double slots) | bp8
__device__ __forceinline__ unsigned lm_key(float v) {  // order-preserving map float -> unsigned (NaN sorts last)
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float lm_unkey(unsigned kx) { return __uint_as_float((kx & 0x80000000u) ? (kx & 0x7fffffffu) : ~kx); }

template <int NV>
__global__ void __launch_bounds__(256) k_lm_select(const float* q, const LmQuery* __restrict__ qinfo, const int64_t* __restrict__ ids,
                                                   const float* __restrict__ vecs, float* S, int nq, int nlist, int d,
                                                   int pitch, double vmax, int k, float* __restrict__ D, int64_t* __restrict__ I,
                                                   int64_t* __restrict__ P, int* __restrict__ any_short, float* bfeats, float rate, float omr,
                                                   int64_t pos_last) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + wave;
    if (p >= nq) return;  // (no block-wide barrier below: waves are independent; a wave's LDS operations execute in order)
    const int pl = min(pitch, LM_MAXPITCH);  // entries of the LDS score row
    char* wbase = smem_raw + (size_t)wave * ((size_t)pl * 4 + LM_WAVE_EXTRA);
    double* cd = (double*)(wbase + (size_t)pl * 4);
    long long* ci = (long long*)(cd + LM_VCAP);
    int* cr = (int*)(ci + LM_VCAP);
    float* bd8 = (float*)(cr + LM_VCAP);
    long long* bp8 = (long long*)(bd8 + 2 * KMAX);
#ifdef LM_STAMPS  // dev: cycle stamps of one wave's phases (RVCMI_DEFINES=LM_STAMPS), printed by block 40
    unsigned long long tst[10];
    int tsn = 0;
#define LM_STAMP() do { if (tsn < 10) tst[tsn++] = __builtin_readcyclecounter(); } while (0)
#else
#define LM_STAMP() do {} while (0)
#endif
    LM_STAMP();
    const LmQuery Q = qinfo[p];
    const int64_t qi = Q.qi, beg = Q.beg;
    const int len = Q.len;
    // this query's score row: staged in LDS, or -- a list longer than the LDS row -- used where the tile kernel wrote it (wave-uniform; the
    // candidate compaction below overwrites the row in place either way: it is this query's own)
    const bool in_lds = len <= pl;
    float* sr = in_lds ? (float*)wbase : S + (int64_t)p * pitch;
    const float* qp = q + qi * d;
    const int d4 = d >> 2;
    constexpr int NVR = NV ? NV : 1;
    float4 xq[NVR];  // NV > 0: this lane's slice of the query, chunks lane, lane + 64, ...
    if constexpr (NV > 0) {
#pragma unroll
        for (int j = 0; j < NV; ++j) xq[j] = ((const float4*)qp)[lane + 64 * j];
    }
    // this lane's scores: entries lane, lane + 64, ... -- the first 8 also stay in registers as sortable keys
    constexpr int KR = 8;
    unsigned kv[KR];
#pragma unroll
    for (int j = 0; j < KR; ++j) {
        const int c = lane + 64 * j;
        const float v = c < len ? S[(int64_t)p * pitch + c] : __uint_as_float(0x7fc00000u);  // beyond the row: NaN = last in key order
        kv[j] = c < len ? lm_key(v) : 0xffffffffu;
        if (c < len && in_lds) sr[c] = v;
    }
    if (in_lds)
        for (int c = lane + 64 * KR; c < len; c += 64) sr[c] = S[(int64_t)p * pitch + c];
    double qn2 = 0.0;
    if constexpr (NV > 0) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            qn2 = fma((double)xq[j].x, (double)xq[j].x, qn2); qn2 = fma((double)xq[j].y, (double)xq[j].y, qn2);
            qn2 = fma((double)xq[j].z, (double)xq[j].z, qn2); qn2 = fma((double)xq[j].w, (double)xq[j].w, qn2);
        }
    } else {
        for (int e = lane; e < d; e += 64) qn2 = fma((double)qp[e], (double)qp[e], qn2);
    }
    for (int off = 32; off >= 1; off >>= 1) qn2 += __shfl_xor(qn2, off, 64);
    LM_STAMP();  // 1: qinfo, q slice, S row, |q|^2
    float lim = INFINITY;
    if (len > k) {
        // An upper bound of the k-th smallest score is enough (a looser threshold only adds candidates): the k-th smallest of the 64
        // LANE MINIMA -- k distinct entries lie at or below it.  Found by binary search over the 32 key bits with ONE ballot per step
        // (no cross-lane data movement; k rounds of wave-wide argmin cost 96 dependent LDS-crossbar shuffles, the exact k-th
        // smallest by ballots over all strides 11.6k cycles).  The true k smallest mostly sit in different lanes: the bound is the
        // k-th to (k+2)-th smallest in practice, i.e. about one extra candidate per query.
        unsigned mk = kv[0];
#pragma unroll
        for (int jj = 1; jj < KR; ++jj) mk = min(mk, kv[jj]);
        for (int c = lane + 64 * KR; c < len; c += 64) mk = min(mk, lm_key(sr[c]));
        const int kk = min(k, min(len, 64));  // (at least kk lanes hold an entry)
        unsigned lo = 0u, hi = 0xffffffffu;
        while (lo < hi) {
            const unsigned mid = lo + ((hi - lo) >> 1);
            if (__builtin_popcountll(__ballot(mk <= mid)) >= kk) hi = mid;
            else lo = mid + 1u;
        }
        const float t8 = lm_unkey(hi);
        // |S_hat - S| <= (2 d + 4) u (|q| vmax + vmax^2), u = 2^-24 (k_coarse_pick); a NaN threshold (non-finite input) keeps every row
        const double E = (2.0 * d + 4.0) * 5.9604644775390625e-08 * (sqrt(qn2) * vmax + vmax * vmax);
        const float thr = (float)((double)t8 + 2.0 * E + 1e-30);
        const float thr_up = __uint_as_float(__float_as_uint(fabsf(thr)) + 2u);  // round the threshold outwards
        lim = thr >= 0.f ? thr_up : -__uint_as_float(__float_as_uint(fabsf(thr)) - 2u);
        if (!(lim == lim)) lim = INFINITY;
    }
    LM_STAMP();  // 2: threshold
    // candidates: their row numbers are compacted into this wave's (now dead) score row ...
    int ncand = 0;
    int* cl = (int*)sr;
    for (int c0 = 0; c0 < len; c0 += 64) {
        const int c = c0 + lane;
        const bool cand = c < len && !(sr[c] > lim);  // (NaN scores are candidates)
        const unsigned long long mask = __ballot(cand);
        // (entry c0 + lane is read by this lane before any lane can overwrite it: the slots written lie below c0 + 64)
        if (cand) cl[ncand + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = c;
        ncand += __builtin_popcountll(mask);
    }
    if (!in_lds) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // (the compacted list was written to global memory: lanes read each other's slots below)
    // ... and verified EIGHT at a time: every row load of a batch is in flight before the first reduction (a typical query has 8-9
    // candidates).  Verified candidates collect in cd / ci / cr; lane j ranks entry j against all others under (distance, id).
    LM_STAMP();  // 3: compaction
    int nm = 0;  // entries in the verified list
    auto rank_of = [&](int j) {  // rank of entry j (j < nm) under the strict order (distance, id, slot)
        const double md = cd[j];
        const long long mi = ci[j];
        int rank = 0;
        for (int o = 0; o < nm; ++o) {
            const double od = cd[o];
            const long long oi = ci[o];
            rank += (od < md || (od == md && (oi < mi || (oi == mi && o < j)))) ? 1 : 0;
        }
        return rank;
    };
    constexpr int CB = 12;  // (mean 8-9 candidates, one wave per SIMD: 512 registers -- one batch for almost every query)
    for (int c0 = 0; c0 < ncand; c0 += CB) {
        if (nm + CB > LM_VCAP) {  // (pathological margins only) keep the best KMAX of the list, in rank order
            const int r = lane < nm ? rank_of(lane) : KMAX;
            const double kd = lane < nm ? cd[lane] : 0.0;
            const long long ki = lane < nm ? ci[lane] : 0;
            const int kr = lane < nm ? cr[lane] : 0;
            if (r < KMAX) { cd[r] = kd; ci[r] = ki; cr[r] = kr; }  // (all reads above precede these writes in program order)
            nm = min(nm, KMAX);
        }
        const int nb = min(CB, ncand - c0);  // wave-uniform: the branches on it below are scalar
        int rr[CB];
        int64_t idv[CB];
        double a[CB];
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            rr[u] = cl[min(c0 + u, ncand - 1)];
            idv[u] = ids[beg + rr[u]];
            a[u] = 0.0;
        }
        if constexpr (NV > 0) {
            float4 y[CB][NV];
#pragma unroll
            for (int u = 0; u < CB; ++u)
#pragma unroll
                for (int j = 0; j < NV; ++j)
                    if (u < 8 || u < nb) y[u][j] = ((const float4*)(vecs + (beg + rr[u]) * d))[lane + 64 * j];
            __builtin_amdgcn_sched_barrier(0);  // every load of the batch is issued before the first use (one round trip, not NV)
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const double x0 = (double)xq[j].x, x1 = (double)xq[j].y, x2 = (double)xq[j].z, x3 = (double)xq[j].w;
#pragma unroll
                for (int u = 0; u < CB; ++u) {
                    if (u >= 8 && u >= nb) continue;  // (the first 8 unconditionally: the usual batch; 9..12 only when present)
                    const double s0 = x0 - (double)y[u][j].x, s1 = x1 - (double)y[u][j].y, s2 = x2 - (double)y[u][j].z, s3 = x3 - (double)y[u][j].w;
                    a[u] = fma(s0, s0, a[u]); a[u] = fma(s1, s1, a[u]); a[u] = fma(s2, s2, a[u]); a[u] = fma(s3, s3, a[u]);
                }
            }
        } else {
            for (int e = lane; e < d4; e += 64) {
                float4 y[CB];
#pragma unroll
                for (int u = 0; u < CB; ++u) y[u] = ((const float4*)(vecs + (beg + rr[u]) * d))[e];
                const float4 x = *(const float4*)(qp + e * 4);
                const double x0 = (double)x.x, x1 = (double)x.y, x2 = (double)x.z, x3 = (double)x.w;
#pragma unroll
                for (int u = 0; u < CB; ++u) {
                    const double s0 = x0 - (double)y[u].x, s1 = x1 - (double)y[u].y, s2 = x2 - (double)y[u].z, s3 = x3 - (double)y[u].w;
                    a[u] = fma(s0, s0, a[u]); a[u] = fma(s1, s1, a[u]); a[u] = fma(s2, s2, a[u]); a[u] = fma(s3, s3, a[u]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            if (u >= 8 && u >= nb) continue;
            for (int off = 32; off >= 1; off >>= 1) a[u] += __shfl_xor(a[u], off, 64);
        }
        if (ncand <= CB) {
            // the usual case -- one batch: lane u < nb takes candidate u and ranks it against the (wave-uniform) distances in
            // registers; no list in LDS, no second pass
            double ma = a[0];
            long long mi = idv[0];
            int mr = rr[0];
#pragma unroll
            for (int u = 1; u < CB; ++u)
                if (lane == u) { ma = a[u]; mi = idv[u]; mr = rr[u]; }
            int rank = 0;
#pragma unroll
            for (int u = 0; u < CB; ++u)
                rank += (u < nb && (a[u] < ma || (a[u] == ma && (idv[u] < mi || (idv[u] == mi && u < lane))))) ? 1 : 0;
            if (lane < nb && rank < KMAX) {
                const float dv = (float)ma;
                const long long pos = beg + mr;
                bd8[rank] = dv;
                bp8[rank] = pos;
                if (rank < k) {
                    D[qi * k + rank] = dv;
                    I[qi * k + rank] = mi;
                    P[qi * k + rank] = pos;
                }
            }
            nm = -nb - 1;  // marks "results written": nb entries
            break;
        }
        {  // lane u < nb appends candidate u
            double ma = a[0];
            long long mi = idv[0];
            int mr = rr[0];
#pragma unroll
            for (int u = 1; u < CB; ++u)
                if (lane == u) { ma = a[u]; mi = idv[u]; mr = rr[u]; }
            if (lane < nb) { cd[nm + lane] = ma; ci[nm + lane] = mi; cr[nm + lane] = mr; }
        }
        nm += nb;
    }
    LM_STAMP();  // 4: verification
    // final ranks; slot s of the result = the entry of rank s, padded with FLT_MAX / -1 like faiss when the list is shorter than k
    const bool direct = nm < 0;  // single-batch fast path: the ranked entries are already out
    if (direct) nm = -nm - 1;
    {
        const int r = (!direct && lane < nm) ? rank_of(lane) : KMAX;
        if (!direct && lane < nm && r < KMAX) {
            const float dv = (float)cd[lane];
            const long long pos = beg + cr[lane];
            bd8[r] = dv;
            bp8[r] = pos;
            if (r < k) {
                D[qi * k + r] = dv;
                I[qi * k + r] = ci[lane];
                P[qi * k + r] = pos;
            }
        }
        if (lane >= nm && lane < KMAX) {
            bd8[lane] = FLT_MAX;
            bp8[lane] = -1;
            if (lane < k) {
                D[qi * k + lane] = FLT_MAX;
                I[qi * k + lane] = -1;
                P[qi * k + lane] = -1;
                atomicOr(any_short, 1);
            }
        }
    }
    LM_STAMP();  // 5: ranks + outputs
    if (bfeats) {
        // pipeline.py:129-138 exactly as k_blend evaluates it (numpy's operation order); the rows were just read: L2 hits
#pragma clang fp contract(off)
        float w[KMAX];
        int64_t pp[KMAX];
#pragma unroll
        for (int s = 0; s < KMAX; ++s) {
            const float inv = div_rn(1.0f, bd8[s]);
            w[s] = mul_rn(inv, inv);
            const int64_t ps = s < k ? bp8[s] : 0;
            pp[s] = ps < 0 ? pos_last : ps;
        }
        float sum;
        if (k == 8) {
            sum = add_rn(add_rn(add_rn(w[0], w[1]), add_rn(w[2], w[3])), add_rn(add_rn(w[4], w[5]), add_rn(w[6], w[7])));
        } else {
            sum = w[0];
#pragma unroll
            for (int s = 1; s < KMAX; ++s)
                if (s < k) sum = add_rn(sum, w[s]);
        }
#pragma unroll
        for (int s = 0; s < KMAX; ++s) w[s] = div_rn(w[s], sum);
        if constexpr (NV > 0) {
            float4 gv[KMAX][NV], fv[NV];
#pragma unroll
            for (int s = 0; s < KMAX; ++s)
#pragma unroll
                for (int j = 0; j < NV; ++j) gv[s][j] = ((const float4*)(vecs + pp[s] * d))[lane + 64 * j];
#pragma unroll
            for (int j = 0; j < NV; ++j) fv[j] = ((const float4*)(bfeats + qi * d))[lane + 64 * j];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                float o[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float acc = 0.f;
#pragma unroll
                    for (int s = 0; s < KMAX; ++s) {
                        if (s < k) {
                            const float g = c == 0 ? gv[s][j].x : c == 1 ? gv[s][j].y : c == 2 ? gv[s][j].z : gv[s][j].w;
                            const float prod = mul_rn(g, w[s]);
                            acc = s == 0 ? prod : add_rn(acc, prod);
                        }
                    }
                    const float f = c == 0 ? fv[j].x : c == 1 ? fv[j].y : c == 2 ? fv[j].z : fv[j].w;
                    o[c] = add_rn(mul_rn(acc, rate), mul_rn(omr, f));
                }
                ((float4*)(bfeats + qi * d))[lane + 64 * j] = make_float4(o[0], o[1], o[2], o[3]);
            }
        } else {
            for (int e = lane; e < d; e += 64) {
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
    }
    LM_STAMP();  // 6: blend
#ifdef LM_STAMPS
    if (lane == 0 && (blockIdx.x % 37) == 3 && wave == 1)
        printf("[lm stamps] blk %d len %d ncand %d: load %llu thr %llu compact %llu verify %llu rank %llu blend %llu\n", (int)blockIdx.x, len, ncand,
               tst[1] - tst[0], tst[2] - tst[1], tst[3] - tst[2], tst[4] - tst[3], tst[5] - tst[4], tst[6] - tst[5]);
#endif
}

}  // namespace rvcmi
