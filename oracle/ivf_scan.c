/* TEST INFRASTRUCTURE ONLY -- plain-C restatement of faiss IndexIVFFlat::search (METRIC_L2) and of
 * the reference's inverse-square blend.  PARITY UNPINNED (faiss-cpu is un-vendored and absent offline;
 * see oracle/ivf_oracle.py for the full statement).  Call sites restated:
 *   index.search(npy, k=8)            infer/modules/vc/pipeline.py:126
 *   weight/gather/blend               infer/modules/vc/pipeline.py:129-138
 * Two flavours: *_f64 = the exact-arithmetic definition used as the parity checker (fp64 direct
 * differences, ties -> lowest id); *_f32 = fp32 arithmetic like faiss' own scanners, used only as
 * the multi-core CPU baseline that bench.py times next to the GPU.  OpenMP over queries.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int before(double da, int64_t ia, double db, int64_t ib) { return da < db || (da == db && ia < ib); }

static void topk_push(double* td, int64_t* ti, int64_t* tp, int k, double d, int64_t id, int64_t pos) {
    if (!before(d, id, td[k - 1], ti[k - 1])) return;
    int s = k - 1;
    while (s > 0 && before(d, id, td[s - 1], ti[s - 1])) {
        td[s] = td[s - 1]; ti[s] = ti[s - 1]; tp[s] = tp[s - 1];
        --s;
    }
    td[s] = d; ti[s] = id; tp[s] = pos;
}

/* coarse: nprobe nearest centroids per query, ascending (dist, id) */
static void coarse(const float* q, const float* cent, int64_t nlist, int d, int nprobe, int64_t* out, int use_f32) {
    double* td = (double*)malloc(sizeof(double) * nprobe);
    int64_t* ti = (int64_t*)malloc(sizeof(int64_t) * nprobe);
    int64_t* tp = (int64_t*)malloc(sizeof(int64_t) * nprobe);
    for (int p = 0; p < nprobe; ++p) { td[p] = INFINITY; ti[p] = INT64_MAX; tp[p] = -1; }
    for (int64_t c = 0; c < nlist; ++c) {
        const float* v = cent + c * d;
        double acc;
        if (use_f32) {
            float a = 0.f;
            for (int e = 0; e < d; ++e) { float t = q[e] - v[e]; a += t * t; }
            acc = a;
        } else {
            acc = 0.0;
            for (int e = 0; e < d; ++e) { double t = (double)q[e] - (double)v[e]; acc += t * t; }
        }
        topk_push(td, ti, tp, nprobe, acc, c, c);
    }
    for (int p = 0; p < nprobe; ++p) out[p] = ti[p] == INT64_MAX ? -1 : ti[p];
    free(td); free(ti); free(tp);
}

/* D [nq,k] float, I [nq,k] int64 (-1 / FLT_MAX padded), P [nq,k] list-major positions (or NULL) */
void ivf_search(const float* q, int64_t nq, int d, const float* cent, int64_t nlist, int nprobe,
                const int64_t* list_off, const int64_t* ids, const float* vecs, int k, float* D, int64_t* I, int64_t* P,
                int use_f32) {
    if (nprobe > nlist) nprobe = (int)nlist;
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t i = 0; i < nq; ++i) {
        const float* qi = q + i * d;
        int64_t lists[64];
        int np = nprobe > 64 ? 64 : nprobe;
        coarse(qi, cent, nlist, d, np, lists, use_f32);
        double td[16]; int64_t ti[16], tp[16];
        for (int s = 0; s < k; ++s) { td[s] = INFINITY; ti[s] = INT64_MAX; tp[s] = -1; }
        for (int p = 0; p < np; ++p) {
            if (lists[p] < 0) continue;
            for (int64_t r = list_off[lists[p]]; r < list_off[lists[p] + 1]; ++r) {
                const float* v = vecs + r * d;
                double acc;
                if (use_f32) {
                    float a = 0.f;
                    for (int e = 0; e < d; ++e) { float t = qi[e] - v[e]; a += t * t; }
                    acc = a;
                } else {
                    acc = 0.0;
                    for (int e = 0; e < d; ++e) { double t = (double)qi[e] - (double)v[e]; acc += t * t; }
                }
                topk_push(td, ti, tp, k, acc, ids[r], r);
            }
        }
        for (int s = 0; s < k; ++s) {
            if (ti[s] == INT64_MAX) { D[i * k + s] = FLT_MAX; I[i * k + s] = -1; if (P) P[i * k + s] = -1; }
            else { D[i * k + s] = (float)td[s]; I[i * k + s] = ti[s]; if (P) P[i * k + s] = tp[s]; }
        }
    }
}

/* pipeline.py:129-138 in numpy's fp32 operation order; feats updated in place.  big_npy rows are
 * addressed through list-major positions P (id -1 -> pos_last == numpy's big_npy[-1]). */
void ivf_blend(float* feats, int64_t nq, int d, const float* D, const int64_t* P, int k, const float* vecs,
               int64_t pos_last, float rate, float omr) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nq; ++i) {
        float w[16];
        for (int s = 0; s < k; ++s) { volatile float inv = 1.0f / D[i * k + s]; w[s] = inv * inv; }
        volatile float sum;
        if (k == 8) {
            volatile float a = w[0] + w[1], b = w[2] + w[3], c = w[4] + w[5], e = w[6] + w[7];
            volatile float ab = a + b, ce = c + e;
            sum = ab + ce;
        } else {
            sum = w[0];
            for (int s = 1; s < k; ++s) sum = sum + w[s];
        }
        for (int s = 0; s < k; ++s) w[s] = w[s] / sum;
        for (int e = 0; e < d; ++e) {
            volatile float acc = 0.f;
            for (int s = 0; s < k; ++s) {
                int64_t p = P[i * k + s];
                if (p < 0) p = pos_last;
                volatile float prod = vecs[p * d + e] * w[s];
                acc = s == 0 ? prod : acc + prod;
            }
            volatile float x = acc * rate, y = omr * feats[i * d + e];
            feats[i * d + e] = x + y;
        }
    }
}
