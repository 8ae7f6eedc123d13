// Launch side of k_rb_stream: strip planning (how many persistent blocks per resblock, how long their strips) and the
// per-channel-count instantiations.  See rb_stream_kernels.hpp for the kernel and tools/model_rb_stream.py for its model.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <array>
#include <cstdlib>
#include <mutex>
#include <utility>
#include <vector>

#include "common.hpp"
#include "rb_stream.hpp"
#include "rb_stream_kernels.hpp"
// (k_rb_stream2 / 2x / 3 -- two blocks per CU, anti-phased groups, half-step slots -- were parity-green and measured SLOWER than
//  k_rb_stream in round 3; removed in round 5, `git log -- csrc/rb_stream2_kernels.hpp` has them.)

namespace rvcmi {
int num_cus();

namespace {

#ifndef RS_KL_DEFAULT
#define RS_KL_DEFAULT 2  // measured (r3k): 0.641 -> 0.623 ms per clip at B = 1 with the planning constant below, 0.614 -> 0.605 at B = 16
#endif

struct Geo {
    int MI, NJ, NCO, bpc;  // bpc = blocks per CU (k_rb_stream: one wave per SIMD, NCO * bpc = 4)
    double c0;             // planning: a block's time is steps x (k + c0) units
};
bool geo_for(int C, int nd, Geo& g, const Options& opt) {
    const bool sm = opt.geti("RS_SMALL", 1) != 0;
    if (C == 256 && nd == 1) { g = {2, sm ? 3 : 4, 4, 1, 4.4}; return true; }
    // (the lean K loop shortens the k = 11 / 7 pair-steps more than the k = 3 ones: per pair-step 45.6k / 33.0k / 20.2k cycles => k + 3.8)
    if (C == 128 && nd == 3) { g = {1, sm ? 6 : 8, 4, 1, opt.get("RS_C0", (sm && opt.geti("RS_KL", RS_KL_DEFAULT) == 2) ? 3.8 : 4.4)}; return true; }
    // (C = 128 pair by pair with TWO blocks per CU -- NJ = 4, 225 registers, 0 spills -- was measured: both waves of a SIMD sit in
    //  their K loops at the same time (72 cycles per MFMA per wave), the phases overlap no better than in the one-wave design
    //  (MFMA pipe 66 % busy in both) and three launches move 3x the bytes: 0.82 vs 0.71 ms on the same box.  Not kept.)
    // C = 64 / 32 (two / four blocks per CU) were built and measured: a pair-step holds too little MFMA work for one wave
    // per SIMD (0.58 vs 0.40 ms and 0.49 vs 0.26 ms per clip against k_rb_full), so they are not instantiated.
    return false;
}


// nblocks < 0: only make sure the > 64 KB dynamic-LDS attribute is set on the current device (once per device and instantiation;
// done at handle creation so that a first forward inside a stream capture does not have to)
template <typename OpT, int C, int MI, int NJ, int NCO, int ND, int KG = 4, int NB = 2, int OCC = 1, int KL = 1, int XH = 0>
void launch_inst(const RbStreamArgs& a, int nblocks, int B, size_t smem, hipStream_t st) {
    static std::atomic<unsigned long long> attr_done{0};
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    auto kern = &k_rb_stream<OpT, C, MI, NJ, NCO, ND, KG, NB, OCC, KL, XH>;
    if (!(attr_done.load() & bit)) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done.fetch_or(bit);
    }
    if (nblocks < 0) return;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks * (unsigned)B), dim3(64 * NCO), smem, st, a);
}


// weight ring of the C = 128 kernel: NB groups of KG k-steps; a group is requested (NB-1)*KG k-steps ahead of its use.
// (4 groups of 2 k-steps -- the same 32 registers, 6 instead of 4 k-steps of L2 latency covered -- was measured: 47 instead
// of 37 cycles per MFMA, the group bookkeeping comes twice as often; 0.69 vs 0.62 ms.  RVCMI_DEFINES="RS_KG128=2 RS_NB128=4".)
#ifndef RS_KG128
#define RS_KG128 4
#endif
#ifndef RS_NB128
#define RS_NB128 2
#endif
template <typename OpT>
void launch_t(int C, int nd, int NJ, const RbStreamArgs& a, int nblocks, int B, size_t smem, hipStream_t st) {
    if (C == 256 && nd == 1 && NJ == 4) return launch_inst<OpT, 256, 2, 4, 4, 1>(a, nblocks, B, smem, st);
    if (C == 256 && nd == 1 && NJ == 3) return launch_inst<OpT, 256, 2, 3, 4, 1>(a, nblocks, B, smem, st);
    if (C == 128 && nd == 3 && NJ == 8) return launch_inst<OpT, 128, 1, 8, 4, 3>(a, nblocks, B, smem, st);
    // (weight ring 3 / 4 groups deep instead of 2: 65 / 129 spilled registers, 0.70 -> 0.77 / 0.89 ms -- measured, not kept)
    if (C == 128 && nd == 3 && NJ == 6 && (a.flags & 4) && (a.flags & 16))
        return launch_inst<OpT, 128, 1, 6, 4, 3, RS_KG128, RS_NB128, 1, 2, 1>(a, nblocks, B, smem, st);  // lean K loop, fp16 input rows
    if (C == 128 && nd == 3 && NJ == 6 && (a.flags & 4)) return launch_inst<OpT, 128, 1, 6, 4, 3, RS_KG128, RS_NB128, 1, 2>(a, nblocks, B, smem, st);  // lean K loop
    if (C == 128 && nd == 3 && NJ == 6) return launch_inst<OpT, 128, 1, 6, 4, 3, RS_KG128, RS_NB128>(a, nblocks, B, smem, st);
    RVCMI_FAIL(RVCMI_ERR_INVALID, "rb_stream: no instantiation for C=%d nd=%d", C, nd);
}

}  // namespace

int num_cus() {
    static std::atomic<int> cached[64];
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    int v = cached[dev & 63].load();
    if (!v) {
        hipDeviceProp_t p;
        HIP_CHECK(hipGetDeviceProperties(&p, dev));
        v = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
        cached[dev & 63].store(v);
    }
    return v;
}

void rb_stream_prepare() {
    RbStreamArgs a;
    memset(&a, 0, sizeof(a));
    for (int nj : {4, 3}) {
        launch_t<__bf16>(256, 1, nj, a, -1, 1, 0, nullptr);
        launch_t<_Float16>(256, 1, nj, a, -1, 1, 0, nullptr);
    }
    for (int nj : {8, 6}) {
        launch_t<__bf16>(128, 3, nj, a, -1, 1, 0, nullptr);
        launch_t<_Float16>(128, 3, nj, a, -1, 1, 0, nullptr);
    }
    for (int fl : {4, 4 | 16}) {  // the lean-K-loop instantiations of NJ = 6: fp32 / fp16 input rows
        a.flags = fl;
        launch_t<__bf16>(128, 3, 6, a, -1, 1, 0, nullptr);
        launch_t<_Float16>(128, 3, 6, a, -1, 1, 0, nullptr);
    }
    a.flags = 0;
}

bool rb_stream_supported(int operand, int C, int nd) {
    Geo g;
    return operand != RVCMI_OPERAND_F32 && geo_for(C, nd, g, Options());
}

static bool launch_geo(const Geo& g, int min_steps_required, int operand, int C, int nd, const RbStreamDesc* jobs, int njobs, int L, int B,
                       long bstride, hipStream_t st, const Options& opt, bool dry_run);

bool rb_stream_launch(int operand, int C, int nd, const RbStreamDesc* jobs, int njobs, int L, int B, long bstride, bool force,
                      hipStream_t st, const Options& opt, bool dry_run) {
    Geo g;
    if (operand == RVCMI_OPERAND_F32 || njobs < 1 || njobs > 3) return false;
    if (!geo_for(C, nd, g, opt)) return false;
    // auto mode: only where the persistent walk measured faster than the tile kernels -- C = 128 (whole resblocks) from 4 steps
    // per block; C = 256 (pair level) only with long strips (large batches).
    return launch_geo(g, force ? 0 : (C == 128 ? 4 : 8), operand, C, nd, jobs, njobs, L, B, bstride, st, opt, dry_run);
}

static bool launch_geo(const Geo& g, int min_steps_required, int operand, int C, int nd, const RbStreamDesc* jobs, int njobs, int L, int B,
                       long bstride, hipStream_t st, const Options& opt, bool dry_run) {
    const int R = 32 * g.NJ;
    RbStreamArgs a;
    memset(&a, 0, sizeof(a));
    a.njobs = njobs;
    a.B = B;
    a.L = L;
    a.bstride = bstride;
    // blocks available to one utterance, shared by the resblocks
    const int slots1 = std::max(njobs, num_cus() * g.bpc / std::max(1, B));
    int nblocks = 0, side_rows = 0, min_steps = 1 << 30;
    int warm[3] = {0, 0, 0};
    for (int j = 0; j < njobs; ++j) {
        const RbStreamDesc& d = jobs[j];
        RbStreamJob& J = a.job[j];
        J.src = d.src;
        J.dst = d.dst;
        J.ct1 = d.ct1;
        J.ct2 = d.ct2;
        J.k = d.k;
        J.k_p = d.k_p;
        const int p2 = (d.k - 1) / 2;
        int sx = 0;
        warm[j] = 32 * nd + nd * p2;
        for (int m = 0; m < nd; ++m) {
            J.w1[m] = d.w1[m];
            J.w2[m] = d.w2[m];
            J.b1[m] = d.b1[m];
            J.b2[m] = d.b2[m];
            J.dil[m] = d.dil[m];
            const int p1 = d.dil[m] * (d.k - 1) / 2;
            if (p1 + p2 > 32 || 2 * p2 > RS_HROW || p1 + p2 + d.dil[m] - 32 > RS_SLACK || 32 + p1 - p2 > RS_HEAD - RS_HROW)
                return false;  // halo larger than the 32-row lag / the head room of the tile: not this kernel
            warm[j] += p1;
            J.sx_off[m] = sx;
            sx += 32 + p1 - p2;
        }
        for (int m = 0; m < nd; ++m) {
            J.sh_off[m] = sx;
            sx += 2 * p2;
        }
        side_rows = std::max(side_rows, sx);
    }
    // Strips.  A block's time is steps x (per-step cost), steps = ceil((rows + warm-up) / R) and the per-step cost measured with
    // the phase stamps is ~ (k + 4.4) units (14.8k + 3.4k * k cycles per pair-step at C = 128): choose the number of strips of
    // every resblock so that the slowest block finishes earliest (brute force over the splits of the slots; cached).
    int nst[3] = {1, 1, 1};
    {
        struct Key { int C, nd, L, slots, R, k[3], n; };
        static std::mutex mu;
        static std::vector<std::pair<Key, std::array<int, 3>>> cache;
        Key key{C, nd, L, slots1, R, {jobs[0].k, njobs > 1 ? jobs[1].k : 0, njobs > 2 ? jobs[2].k : 0}, njobs};
        bool hit = false;
        {
            std::lock_guard<std::mutex> lk(mu);
            for (auto& e : cache)
                if (!memcmp(&e.first, &key, sizeof(Key))) { nst[0] = e.second[0]; nst[1] = e.second[1]; nst[2] = e.second[2]; hit = true; break; }
        }
        if (!hit) {
            const int maxn = std::max(1, L / R);
            auto tcost = [&](int j, int n) {
                const int rows = (L + n - 1) / n;
                const int steps = (rows + warm[j] + R - 1) / R;
                return steps * (jobs[j].k + g.c0);
            };
            // With a large batch an utterance gets only a handful of blocks, too few to split in proportion to the three
            // costs (B = 64: 4 blocks as 2+1+1 = 75 % balance).  Strips are cheap (156 warm-up rows each), so also try
            // 2x, 3x, 4x the blocks -- whole extra rounds of equally long blocks -- and keep the fastest estimate.
            double best_total = 1e300;
            for (int mult = 1; mult <= (slots1 >= 64 ? 1 : 4); ++mult) {
                const int slots = slots1 * mult;
                int cand[3] = {1, 1, 1};
                double best = 1e300;
                const int lim0 = std::min(maxn, slots - (njobs - 1));
                for (int n0 = 1; n0 <= lim0; ++n0) {
                    if (njobs == 1) {
                        const double t = tcost(0, n0);
                        if (t < best) { best = t; cand[0] = n0; }
                        continue;
                    }
                    const int lim1 = std::min(maxn, slots - n0 - (njobs - 2));
                    for (int n1 = 1; n1 <= lim1; ++n1) {
                        double t = std::max(tcost(0, n0), tcost(1, n1));
                        if (t >= best) continue;
                        if (njobs == 2) { best = t; cand[0] = n0; cand[1] = n1; continue; }
                        const int n2 = std::min(maxn, slots - n0 - n1);
                        if (n2 < 1) continue;
                        t = std::max(t, tcost(2, n2));
                        if (t < best) { best = t; cand[0] = n0; cand[1] = n1; cand[2] = n2; }
                    }
                }
                if (best * mult < best_total * 0.97) {  // (a larger grid has to pay for itself)
                    best_total = best * mult;
                    nst[0] = cand[0]; nst[1] = cand[1]; nst[2] = cand[2];
                }
            }
            std::lock_guard<std::mutex> lk(mu);
            if (cache.size() > 64) cache.clear();
            cache.push_back({key, {nst[0], nst[1], nst[2]}});
        }
    }
    for (int j = 0; j < njobs; ++j) {
        RbStreamJob& J = a.job[j];
        const int rows = (L + nst[j] - 1) / nst[j];
        const int steps = (rows + warm[j] + R - 1) / R;
        const int len = std::max(rows, steps * R - warm[j]);
        J.strip_len = len;
        J.nstrips = (L + len - 1) / len;
        J.blk0 = nblocks;
        nblocks += J.nstrips;
        min_steps = std::min(min_steps, steps);
    }
    a.side_rows = side_rows;
    // k_rb_stream with the lean K loop (kconv) and the coalesced step IO; its buffer descriptors address an utterance's rows with
    // 32-bit byte offsets, so utterances of 2 GiB or more per stream (349 s of audio at C = 128) keep the D-layout form
    if (opt.geti("RS_KL", RS_KL_DEFAULT) == 2 && (double)L * C * 4.0 < 2147483648.0) a.flags = 4;
    a.lens = jobs[0].lens;
    a.lmul = jobs[0].lmul;
    if (jobs[0].y_half) a.flags |= 8;
    if (jobs[0].x_half) {
        if (!(a.flags & 4) || !(C == 128 && nd == 3 && g.NJ == 6)) return false;  // fp16 input rows: only in the coalesced step IO of KL = 2
        a.flags |= 16;
    }
    if (min_steps < min_steps_required) return false;
    const size_t smem = (size_t)(RS_HEAD + R + RS_SLACK + side_rows + 1) * (2 * C + 16) + (size_t)nd * 2 * C * sizeof(float);
    // k_rb_stream with KL = 2 (lean K loop + coalesced step IO): [side | dump | biases | M ...] with the fp32 transposition tile
    // T (R rows x (4 C + 16) bytes) starting at M and running over its end (rb_stream_kernels.hpp)
    const bool kl2 = C == 128 && nd == 3 && g.NJ == 6 && (a.flags & 4);
    const size_t smem_kl2 = (size_t)(side_rows + 1) * (2 * C + 16) + (size_t)nd * 2 * C * sizeof(float) +
                            std::max((size_t)(RS_HEAD + R + RS_SLACK) * (2 * C + 16), (size_t)R * (4 * C + 16));
    if (kl2 && smem_kl2 > (size_t)160 * 1024) RVCMI_FAIL(RVCMI_ERR_INVALID, "rb_stream: LDS image %zu B too large (C=%d)", smem_kl2, C);
    if (smem > (size_t)160 * 1024 / g.bpc)
        RVCMI_FAIL(RVCMI_ERR_INVALID, "rb_stream: LDS image %zu B too large for %d block(s) per CU (C=%d)", smem, g.bpc, C);
    if (dry_run) return true;
    // dev only: option RS_STAMPS prints the per-phase cycle breakdown of every launch (synchronises; never set it in a timed run)
    const bool want_stamps = opt.on("RS_STAMPS");
    unsigned long long* ts = nullptr;
    const size_t nts = (size_t)nblocks * B * g.NCO * 16;
    if (want_stamps) {
        HIP_CHECK(hipMalloc(&ts, nts * 8));
        HIP_CHECK(hipMemsetAsync(ts, 0, nts * 8, st));
        a.ts = ts;
    }
    if (operand == RVCMI_OPERAND_BF16) launch_t<__bf16>(C, nd, g.NJ, a, nblocks, B, kl2 ? smem_kl2 : smem, st);
    else launch_t<_Float16>(C, nd, g.NJ, a, nblocks, B, kl2 ? smem_kl2 : smem, st);
    if (want_stamps) {
        HIP_CHECK(hipStreamSynchronize(st));
        std::vector<unsigned long long> h(nts);
        HIP_CHECK(hipMemcpy(h.data(), ts, nts * 8, hipMemcpyDeviceToHost));
        (void)hipFree(ts);
        static const char* names[10] = {"phaseA", "barA", "conv1", "bar1", "phaseB", "barB", "conv2", "bar2", "xload", "store"};
        for (int j = 0; j < njobs; ++j) {
            double sum[10] = {0}, steps = 0, tot_max = 0;
            long cnt = 0;
            for (size_t w = 0; w < (size_t)nblocks * B * g.NCO; ++w) {
                const unsigned long long* p = &h[w * 16];
                if ((int)p[11] != j || !p[10]) continue;
                double tot = 0;
                for (int i = 0; i < 10; ++i) { sum[i] += (double)p[i]; tot += (double)p[i]; }
                tot_max = std::max(tot_max, tot);
                steps += (double)p[10];
                ++cnt;
            }
            if (!cnt) continue;
            fprintf(stderr, "[rs stamps] C=%d nd=%d NJ=%d k=%d strips=%d len=%d steps/blk=%.1f  cycles per PAIR-step:", C, nd, g.NJ, a.job[j].k,
                    a.job[j].nstrips, a.job[j].strip_len, steps / cnt);
            for (int i = 0; i < 8; ++i) fprintf(stderr, " %s %.0f", names[i], sum[i] / (steps * nd));
            fprintf(stderr, " | per step: xload %.0f store %.0f | slowest wave total %.0f cycles\n", sum[8] / steps, sum[9] / steps, tot_max);
        }
    }
    return true;
}

}  // namespace rvcmi
