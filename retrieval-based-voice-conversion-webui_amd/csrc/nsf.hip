// Host side of the generator: weight preparation (stands in for rvc/synthesizer.py:10-28 for
// `net_g.dec`), workspace, and the launch sequence of NSFGenerator.forward (rvc/layers/nsf.py:145-191)
// / Generator.forward (rvc/layers/generators.py:70-98).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <functional>
#include <memory>
#include <string>
#include <unordered_map>

#include "common.hpp"
#include "nsf_kernels.hpp"
#include "conv_pack.hpp"
#include "rb_stream.hpp"

namespace rvcmi {

thread_local std::string g_last_error;
void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

struct Geom {
    int MI, NJ, WCO, TT;
};
// Block/wave tiling per output-channel count (see k_conv_mfma): waves split output channels first
// (each wave streams its OWN weight slice), time second.
static Geom conv_geom(int cout) {
    Geom g;
    g.NJ = 4;
    if (cout >= 256) { g.MI = 2; g.WCO = 4; }
    else if (cout >= 128) { g.MI = 2; g.WCO = 2; }
    else if (cout >= 64) { g.MI = 2; g.WCO = 1; }
    else { g.MI = 1; g.WCO = 1; }
    g.TT = (4 / g.WCO) * g.NJ * 32;
    return g;
}

struct Stage {
    int cin, cout, u, k, pad;
    ConvLayer up;
    // noise conv (nsf.py:103-115)
    int nk = 0, ns = 1, npad = 0;
    DevBuf noise_w, noise_b;
    ConvLayer nz;          // the same noise conv as a 2-tap MFMA conv over frames of `ns` samples (when ns % 8 == 0)
    bool nz_mfma = false;
    DevBuf nz_k1_w;        // ... or, for windows of <= 16 samples, as ONE extra k-step inside k_ups
    bool nz_k1 = false;
    // resblocks[j].pair[m] = {conv1, conv2}
    std::vector<std::vector<std::pair<ConvLayer, ConvLayer>>> rb;
};

}  // namespace rvcmi

using namespace rvcmi;

struct rvcmi_nsf {
    rvcmi_nsf_config cfg;
    int device = 0;
    int max_B = 0, max_T = 0;
    int upp = 1;
    int C0 = 0;
    ConvLayer pre;
    DevBuf cond_w, cond_b;
    float lin_w = 1.f, lin_b = 0.f;
    std::vector<Stage> stages;
    DevBuf post_w;
    int c_last = 0;
    // workspace
    size_t S = 0;  // elements per activation buffer
    // P: conv_pre output.  X0: ups(+noise) output = input of the stage's resblocks.  Ya[j]/Yb[j]: ping-pong
    // fp32 streams of resblock j (its final output is one of them; the consumer sums the nk of them).
    DevBuf P, X0, Ya[RVCMI_MAX_RB], Yb[RVCMI_MAX_RB], H, NZ, har, har2, x2, phase, condv, dbg;
    size_t ws_bytes = 0;
    Profiler prof;
    // dev / test options (common.hpp Options): RB_STREAM (absent = auto, 1 = streaming resblock kernels whenever supported, 0 = never),
    // NO_RBFULL, NO_RB_SPLIT, RB_ORDER (1 = resblock-major pair launches), UPS_NJ, NB (weight-ring depth of k_rb_full), DBG (timing-ablation bit
    // mask: results are WRONG when non-zero) and the rb_stream keys (rb_stream.hpp).  Read from RVCMI_<KEY> once, in rvcmi_nsf_create.
    rvcmi::Options opt;
};

namespace rvcmi {


static void set_lds_limits();
constexpr int RB_KG = 4;  // k-steps per weight-prefetch group inside the fused resblock kernel
#ifndef RB256_NJ
#define RB256_NJ 4  // (96-row tiles, 2 blocks per CU, measured the same 0.33 ms at B=1; 128 rows reuse weights better)
#endif
static int rb_rows(int C) { return C == 256 ? 32 * RB256_NJ : RB_ROWS; }
// Resblock stages of at most this many rows (B x L) at C = 256 run conv by conv, split over output channels (run_conv_jobs): a
// realtime chunk's first stage is 310 rows = 9 fused pair tiles, each pulling 2.9 MB of weights through ONE CU (3 x 70 us).
constexpr int RB_SPLIT_MAX_ROWS = 1024;      // C = 256
constexpr int RB_SPLIT_MAX_ROWS_128 = 4096;  // C = 128 (a chunk's second stage: 3100 rows = 75 pair tiles of 3 x 40 us)

static void nsf_create(const rvcmi_nsf_config* cfg, const rvcmi_tensor* weights, int n_weights, int device, int max_B,
                       int max_T, rvcmi_nsf** out) {
    if (!cfg || !out || (!weights && n_weights)) RVCMI_FAIL(RVCMI_ERR_INVALID, "null argument");
    if (cfg->n_ups < 1 || cfg->n_ups > RVCMI_MAX_UPS || cfg->n_resblock_kernels < 1 ||
        cfg->n_resblock_kernels > RVCMI_MAX_RB)
        RVCMI_FAIL(RVCMI_ERR_INVALID, "bad stage counts");
    if (cfg->operand < 0 || cfg->operand > 2) RVCMI_FAIL(RVCMI_ERR_INVALID, "bad operand type %d", cfg->operand);
    if (max_B < 1 || max_T < 1) RVCMI_FAIL(RVCMI_ERR_INVALID, "max_B/max_T must be positive");
    HIP_CHECK(hipSetDevice(device));
    set_lds_limits();
    rb_stream_prepare();

    std::unique_ptr<rvcmi_nsf> h(new rvcmi_nsf());
    h->opt.load_env({"RB_STREAM", "NO_RBFULL", "NO_RB_SPLIT", "NO_RB_SPLIT128", "RB_ORDER", "UPS_NJ", "UPS_TW", "PRE_OP", "POST_DMA", "POST_DMA_OCC", "POST_DBG", "NB", "DBG", "Y_F16", "X0_F16", "UPS_TR", "UPS_BL", "X0_F16_NOSTREAM", "CONV_KS", "RBF_SMALL"});
    rb_stream_load_env(h->opt);
    h->cfg = *cfg;
    h->device = device;
    h->max_B = max_B;
    h->max_T = max_T;
    const int op = cfg->operand;
    WeightMap wm;
    for (int i = 0; i < n_weights; ++i) wm.m[weights[i].name] = &weights[i];

    const int C0 = cfg->upsample_initial_channel, inter = cfg->inter_channels;
    h->C0 = C0;
    {  // conv_pre: Conv1d(inter, C0, 7, padding=3)   nsf.py:84-86
        const rvcmi_tensor& w = wm.get("conv_pre.weight", {C0, inter, 7});
        const rvcmi_tensor& b = wm.get("conv_pre.bias", {C0});
        const float* wd = w.data;
        int nt = 7, off = -3;
        build_conv(h->pre, inter, C0, 1, &nt, &off, 1,
                   [=](int co, int ci, int, int tap) { return wd[((size_t)co * inter + ci) * 7 + tap]; }, b.data, op);
    }
    if (cfg->gin_channels) {  // cond: Conv1d(gin, C0, 1)   nsf.py:129-130
        const rvcmi_tensor& w = wm.get("cond.weight", {C0, cfg->gin_channels, 1});
        const rvcmi_tensor& b = wm.get("cond.bias", {C0});
        upload(h->cond_w, std::vector<float>(w.data, w.data + (size_t)C0 * cfg->gin_channels));
        upload(h->cond_b, std::vector<float>(b.data, b.data + C0));
    }
    if (cfg->use_f0) {  // m_source.l_linear: Linear(1,1)   nsf.py:51
        h->lin_w = wm.get("m_source.l_linear.weight", {1, 1}).data[0];
        h->lin_b = wm.get("m_source.l_linear.bias", {1}).data[0];
    }
    int upp = 1;
    for (int i = 0; i < cfg->n_ups; ++i) upp *= cfg->upsample_rates[i];
    h->upp = upp;

    h->stages.resize(cfg->n_ups);
    size_t S = (size_t)max_T * C0;
    long L = max_T;
    for (int i = 0; i < cfg->n_ups; ++i) {
        Stage& s = h->stages[i];
        s.cin = C0 >> i;
        s.cout = C0 >> (i + 1);
        s.u = cfg->upsample_rates[i];
        s.k = cfg->upsample_kernel_sizes[i];
        s.pad = (s.k - s.u) / 2;
        if (s.u > 16) RVCMI_FAIL(RVCMI_ERR_INVALID, "upsample rate %d > 16 unsupported", s.u);
        if (s.cout < 4 || (s.cout & 3)) RVCMI_FAIL(RVCMI_ERR_INVALID, "stage %d has %d channels", i, s.cout);
        // The reference requires len(ups) == len(noise_convs): (L-1)*u - 2*pad + k == u*L  <=>  k - 2*pad == u
        if (s.k - 2 * s.pad != s.u) RVCMI_FAIL(RVCMI_ERR_INVALID, "stage %d: k-u must be even (k=%d,u=%d)", i, s.k, s.u);
        L *= s.u;
        S = std::max(S, (size_t)L * s.cout);
        {  // ups[i]: ConvTranspose1d(cin, cout, k, u, padding=(k-u)//2)   nsf.py:92-102
            // out[q*u + r] = sum_j x[q + off_r - j] * W[:, :, cm_r + j*u],  c = r + pad, off_r = c / u, cm_r = c % u
            const rvcmi_tensor& w = wm.get("ups." + std::to_string(i) + ".weight", {s.cin, s.cout, s.k});
            const rvcmi_tensor& b = wm.get("ups." + std::to_string(i) + ".bias", {s.cout});
            int nt[16], off[16], cm[16];
            for (int r = 0; r < s.u; ++r) {
                const int c = r + s.pad;
                off[r] = c / s.u;
                cm[r] = c % s.u;
                nt[r] = (s.k - cm[r] + s.u - 1) / s.u;
            }
            const float* wd = w.data;
            const int cout = s.cout, k = s.k, u = s.u;
            std::vector<int> cmv(cm, cm + s.u);
            build_conv(s.up, s.cin, s.cout, s.u, nt, off, -1,
                       [=](int co, int ci, int p, int tap) { return wd[((size_t)ci * cout + co) * k + cmv[p] + tap * u]; },
                       b.data, op);
        }
        if (cfg->use_f0) {  // noise_convs[i]   nsf.py:103-115
            int sf = 1;
            for (int j = i + 1; j < cfg->n_ups; ++j) sf *= cfg->upsample_rates[j];
            const bool last = i + 1 == cfg->n_ups;
            s.nk = last ? 1 : 2 * sf;
            s.ns = last ? 1 : sf;
            s.npad = last ? 0 : sf / 2;
            const rvcmi_tensor& w = wm.get("noise_convs." + std::to_string(i) + ".weight", {s.cout, 1, s.nk});
            const rvcmi_tensor& b = wm.get("noise_convs." + std::to_string(i) + ".bias", {s.cout});
            std::vector<float> wt((size_t)s.nk * s.cout);
            for (int co = 0; co < s.cout; ++co)
                for (int j = 0; j < s.nk; ++j) wt[(size_t)j * s.cout + co] = w.data[(size_t)co * s.nk + j];
            upload(s.noise_w, wt);
            upload(s.noise_b, std::vector<float>(b.data, b.data + s.cout));
            if (op != RVCMI_OPERAND_F32 && !last && s.ns % 8 == 0 && s.ns <= 64) {
                const int cinp = s.ns <= 16 ? 16 : (s.ns <= 32 ? 32 : 64);
                int nt = 2, off = 0;
                const float* wd = w.data;
                const int ns = s.ns, nkk = s.nk;
                build_conv(s.nz, cinp, s.cout, 1, &nt, &off, 1,
                           [=](int co, int ci, int, int tap) { return ci < ns ? wd[(size_t)co * nkk + tap * ns + ci] : 0.f; }, b.data, op);
                s.nz_mfma = true;
            } else if (op != RVCMI_OPERAND_F32 && s.nk >= 2 && s.nk <= 16 && s.ns % 2 == 0) {
                const int ctiles = (s.cout + 31) / 32;
                std::vector<uint16_t> pk((size_t)ctiles * 512, 0);
                for (int ct = 0; ct < ctiles; ++ct)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int co = ct * 32 + (lane & 31), kk = 8 * (lane >> 5) + e;
                            if (co < s.cout && kk < s.nk) {
                                const float v = w.data[(size_t)co * s.nk + kk];
                                pk[(size_t)ct * 512 + lane * 8 + e] = op == RVCMI_OPERAND_BF16 ? f32_to_bf16(v) : f32_to_f16(v);
                            }
                        }
                s.nz_k1_w.alloc(pk.size() * 2);
                HIP_CHECK(hipMemcpy(s.nz_k1_w.p, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
                s.nz_k1 = true;
            }
        }
        s.rb.resize(cfg->n_resblock_kernels);
        for (int j = 0; j < cfg->n_resblock_kernels; ++j) {  // ResBlock1   residuals.py:19-58
            const int n = i * cfg->n_resblock_kernels + j;
            const int k = cfg->resblock_kernel_sizes[j];
            const int nd = cfg->n_dilations[j];
            if (nd < 1 || nd > RVCMI_MAX_DIL || !(k & 1)) RVCMI_FAIL(RVCMI_ERR_INVALID, "bad resblock %d", j);
            s.rb[j].resize(nd);
            for (int m = 0; m < nd; ++m) {
                const int d = cfg->resblock_dilation_sizes[j][m];
                const std::string base = "resblocks." + std::to_string(n);
                const int C = s.cout;
                for (int which = 0; which < 2; ++which) {
                    const std::string nm = base + (which ? ".convs2." : ".convs1.") + std::to_string(m);
                    const rvcmi_tensor& w = wm.get(nm + ".weight", {C, C, k});
                    const rvcmi_tensor& b = wm.get(nm + ".bias", {C});
                    const int dil = which ? 1 : d;
                    int nt = k, off = -((k * dil - dil) / 2);  // get_padding, rvc/layers/utils.py:14
                    const float* wd = w.data;
                    ConvLayer& L2 = which ? s.rb[j][m].second : s.rb[j][m].first;
                    build_conv(L2, C, C, 1, &nt, &off, dil,
                               [=](int co, int ci, int, int tap) { return wd[((size_t)co * C + ci) * k + tap]; }, b.data, op);
                }
            }
        }
    }
    h->c_last = C0 >> cfg->n_ups;
    {  // conv_post: Conv1d(ch, 1, 7, padding=3, bias=False)   nsf.py:126
        const rvcmi_tensor& w = wm.get("conv_post.weight", {1, h->c_last, 7});
        std::vector<float> wt((size_t)7 * h->c_last);
        for (int c = 0; c < h->c_last; ++c)
            for (int j = 0; j < 7; ++j) wt[(size_t)j * h->c_last + c] = w.data[(size_t)c * 7 + j];
        upload(h->post_w, wt);
    }
    if (h->c_last & 3) RVCMI_FAIL(RVCMI_ERR_INVALID, "last stage channel count %d", h->c_last);

    // workspace
    if (cfg->n_resblock_kernels > 3) RVCMI_FAIL(RVCMI_ERR_INVALID, "more than 3 resblock kernels per stage is unsupported");
    h->S = S * (size_t)max_B;
    const size_t fb = h->S * sizeof(float);
    h->P.alloc((size_t)max_B * max_T * C0 * sizeof(float));
    h->X0.alloc(fb);
    for (int j = 0; j < cfg->n_resblock_kernels; ++j) {
        h->Ya[j].alloc(fb);
        h->Yb[j].alloc(fb);
    }
    if (op == RVCMI_OPERAND_F32) h->H.alloc(fb);
    else h->H.alloc((size_t)3 * std::max(RB_SPLIT_MAX_ROWS * 256, RB_SPLIT_MAX_ROWS_128 * 128) * 2);  // conv1 outputs of the output-channel-split resblock path (tiny launches)
    {
        size_t nz = 0;
        long Ls = max_T;
        for (int i = 0; i < cfg->n_ups; ++i) {
            Ls *= cfg->upsample_rates[i];
            if (h->stages[i].nz_mfma) nz = std::max(nz, (size_t)Ls * h->stages[i].cout);
        }
        if (nz) h->NZ.alloc(nz * max_B * sizeof(float));
    }
    const size_t hb = (size_t)max_B * max_T * upp * sizeof(float);
    h->har.alloc(hb);
    h->har2.alloc(hb);
    h->x2.alloc((size_t)max_B * inter * max_T * sizeof(float));
    h->phase.alloc((size_t)max_B * max_T * sizeof(float));
    h->condv.alloc((size_t)max_B * C0 * sizeof(float));
    HIP_CHECK(hipMemset(h->condv.p, 0, h->condv.bytes));
    h->ws_bytes = (size_t)(1 + 2 * cfg->n_resblock_kernels + (op == RVCMI_OPERAND_F32 ? 1 : 0)) * fb + h->P.bytes + 2 * hb + h->x2.bytes + h->phase.bytes + h->condv.bytes;
    *out = h.release();
}

// ---- launches -----------------------------------------------------------------------------------

template <typename OpT, int CIN, int MI, int NJ, int WCO>
static void launch_inst(const ConvArgs& a, int B, hipStream_t st) {
    using TL = Tile<CIN>;
    constexpr int TT = (4 / WCO) * NJ * 32;
    const int co_blocks = (a.cout + 32 * MI * WCO - 1) / (32 * MI * WCO);
    dim3 grid((a.Lq + TT - 1) / TT, co_blocks * a.nphase, B);
    const size_t smem = (size_t)a.tile_rows * TL::STRIDE;
    if (smem > 160 * 1024) RVCMI_FAIL(RVCMI_ERR_INVALID, "LDS tile too large: %zu bytes (cin %d)", smem, CIN);
    hipLaunchKernelGGL((k_conv_mfma<OpT, CIN, MI, NJ, WCO>), grid, dim3(256), smem, st, a);
}

template <typename OpT, int CIN, int MI, int NJ, int WCO>
static void set_lds_inst() {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_mfma<OpT, CIN, MI, NJ, WCO>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
}

#define RVCMI_FOR_EACH_CIN(X, OpT, MI, NJ, WCO) \
    X(OpT, 16, MI, NJ, WCO)                     \
    X(OpT, 32, MI, NJ, WCO)                     \
    X(OpT, 64, MI, NJ, WCO)                     \
    X(OpT, 128, MI, NJ, WCO)                    \
    X(OpT, 192, MI, NJ, WCO)                    \
    X(OpT, 256, MI, NJ, WCO)                    \
    X(OpT, 512, MI, NJ, WCO)

template <typename OpT, int MI, int NJ, int WCO>
static void launch_by_cin(const ConvArgs& a, int B, hipStream_t st) {
    switch (a.cin) {
#define X(OpT_, CIN_, MI_, NJ_, WCO_) \
    case CIN_:                        \
        return launch_inst<OpT_, CIN_, MI_, NJ_, WCO_>(a, B, st);
        RVCMI_FOR_EACH_CIN(X, OpT, MI, NJ, WCO)
#undef X
        default:
            RVCMI_FAIL(RVCMI_ERR_INVALID, "unsupported C_in %d for the MFMA path", a.cin);
    }
}

template <typename OpT>
static void launch_conv_mfma(const ConvArgs& a, int nj, int B, hipStream_t st) {
    const Geom g = conv_geom(a.cout);
    if (g.MI == 2 && g.WCO == 4 && nj == 1) return launch_by_cin<OpT, 2, 1, 4>(a, B, st);
    if (g.MI == 2 && g.WCO == 4) return launch_by_cin<OpT, 2, 4, 4>(a, B, st);
    if (g.MI == 2 && g.WCO == 2) return launch_by_cin<OpT, 2, 4, 2>(a, B, st);
    if (g.MI == 2 && g.WCO == 1) return launch_by_cin<OpT, 2, 4, 1>(a, B, st);
    return launch_by_cin<OpT, 1, 4, 1>(a, B, st);
}

template <typename OpT>
static void set_lds_all() {
#define X(OpT_, CIN_, MI_, NJ_, WCO_) set_lds_inst<OpT_, CIN_, MI_, NJ_, WCO_>();
    RVCMI_FOR_EACH_CIN(X, OpT, 2, 4, 4)
    RVCMI_FOR_EACH_CIN(X, OpT, 2, 1, 4)
    RVCMI_FOR_EACH_CIN(X, OpT, 2, 4, 2)
    RVCMI_FOR_EACH_CIN(X, OpT, 2, 4, 1)
    RVCMI_FOR_EACH_CIN(X, OpT, 1, 4, 1)
#undef X
}
template <typename OpT>
static void set_lds_rb();
template <typename OpT>
static void set_lds_ups();
template <typename OpT>
static void set_lds_rbf();
static void set_lds_limits() {
    set_lds_all<__bf16>();
    set_lds_all<_Float16>();
    set_lds_rb<__bf16>();
    set_lds_rb<_Float16>();
    set_lds_ups<__bf16>();
    set_lds_ups<_Float16>();
    set_lds_rbf<__bf16>();
    set_lds_rbf<_Float16>();
}


template <typename OpT, int C, int MI, int NW, int KG, int NJ = RB_ROWS / 32, int NWT = 1, int OCC = 2>
static void launch_rb_inst(const RbPairArgs& ra, int tiles, int nj, int B, int rows, hipStream_t st) {
    const size_t smem = (size_t)rows * Tile<C>::STRIDE + 2 * 32 * MI * NW * 4 + NW * NWT * 64;  // tile + the two bias vectors + dev stamps
    if (smem > 160 * 1024) RVCMI_FAIL(RVCMI_ERR_INVALID, "resblock LDS tile too large (%zu B)", smem);
    hipLaunchKernelGGL((k_rb_pair<OpT, C, MI, NW, KG, NJ, NWT, OCC>), dim3(tiles, nj, B), dim3(64 * NW * NWT), smem, st, ra);
}
template <typename OpT>
static void launch_rb_pair_t(int C, const RbPairArgs& ra, int tiles, int nj, int B, int rows, hipStream_t st) {
    switch (C) {
#ifdef RB_MI1  // experiment: 32-channel waves (64 accumulator registers) -> more waves per SIMD
        case 256: return launch_rb_inst<OpT, 256, 1, 8, RB_KG>(ra, tiles, nj, B, rows, st);
        case 128: return launch_rb_inst<OpT, 128, 1, 4, RB_KG>(ra, tiles, nj, B, rows, st);
#else
#ifdef RB_TSPLIT  // experiment: 128 rows split over 2 time slabs (NJ = 2, 64 accumulator registers) -> 2x the waves per tile
        case 256: return launch_rb_inst<OpT, 256, 2, 4, RB_KG, 2, 2, 2>(ra, tiles, nj, B, rows, st);
        case 128: return launch_rb_inst<OpT, 128, 2, 2, RB_KG, 2, 2, RB_TSPLIT>(ra, tiles, nj, B, rows, st);
#else
        case 256: return launch_rb_inst<OpT, 256, 2, 4, RB_KG, RB256_NJ>(ra, tiles, nj, B, rows, st);
        case 128: return launch_rb_inst<OpT, 128, 2, 2, RB_KG>(ra, tiles, nj, B, rows, st);
#endif
#endif
        case 64: return launch_rb_inst<OpT, 64, 2, 1, RB_KG>(ra, tiles, nj, B, rows, st);
        case 32: return launch_rb_inst<OpT, 32, 1, 1, RB_KG>(ra, tiles, nj, B, rows, st);
        case 16: return launch_rb_inst<OpT, 16, 1, 1, RB_KG>(ra, tiles, nj, B, rows, st);
        default: RVCMI_FAIL(RVCMI_ERR_INVALID, "unsupported resblock channel count %d", C);
    }
}
static void launch_rb_pair(int op, int C, const RbPairArgs& ra, int tiles, int nj, int B, int rows, hipStream_t st) {
    if (op == RVCMI_OPERAND_BF16) launch_rb_pair_t<__bf16>(C, ra, tiles, nj, B, rows, st);
    else launch_rb_pair_t<_Float16>(C, ra, tiles, nj, B, rows, st);
}
template <typename OpT>
static void set_lds_rb() {
#define RB_ATTR(C_, MI_, NW_)                                                                                   \
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rb_pair<OpT, C_, MI_, NW_, RB_KG>),           \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#ifdef RB_MI1
    RB_ATTR(256, 1, 8) RB_ATTR(128, 1, 4)
#else
#ifdef RB_TSPLIT
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rb_pair<OpT, 256, 2, 4, RB_KG, 2, 2, 2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rb_pair<OpT, 128, 2, 2, RB_KG, 2, 2, RB_TSPLIT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#else
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rb_pair<OpT, 256, 2, 4, RB_KG, RB256_NJ>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    RB_ATTR(128, 2, 2)
#endif
#endif
    RB_ATTR(64, 2, 1) RB_ATTR(32, 1, 1) RB_ATTR(16, 1, 1)
#undef RB_ATTR
}

template <typename OpT, int CIN, int MI, int WV>
static void launch_ups_inst(UpsArgs a, int nj, int B, hipStream_t st) {
    const int TQ = 32 * nj * (4 / WV);
    size_t smem = (size_t)a.tile_rows * Tile<CIN>::STRIDE + (a.nz_k1 ? (size_t)(TQ * a.u * a.ns + 16) * 2 : 0);
    // option UPS_BL: [2][cout] fp32 LDS copy of bias / bn.  Reserved FIRST so that both bounds below see it (appended after
    // the checks, a 52 KB block became 54 KB -- two resident blocks instead of three -- and a tile near 160 KB would have failed at launch);
    // dropped where it would break the 160 KB limit
    if (a.bias_off && a.cout % 4 == 0 && (smem + 15) / 16 * 16 + (size_t)2 * a.cout * 4 <= 160 * 1024) {
        smem = (smem + 15) / 16 * 16;
        a.bias_off = (int)smem;
        smem += (size_t)2 * a.cout * 4;
    } else {
        a.bias_off = 0;
    }
    if (smem > 160 * 1024) RVCMI_FAIL(RVCMI_ERR_INVALID, "upsampler LDS tile too large (%zu B)", smem);
    const int per_block = WV * a.vpw;
    dim3 grid((a.Lin + TQ - 1) / TQ, (a.nvt + per_block - 1) / per_block, B);
    // row-wise output through an LDS tile when the block owns whole rows (all phases and channels) and the tile keeps the block small
    // enough for three per CU (everything counted: operand tile, har copy, bias copy, output tile); (cout * esz) must be whole 16-byte chunks
    {
        const int esz = a.out_half ? 2 : 4;
        const size_t ot = (size_t)TQ * a.u * ((size_t)a.cout * esz + 16);
        smem = (smem + 15) / 16 * 16;
        if (a.out_tr && grid.y == 1 && (a.cout * esz) % 16 == 0 && smem + ot <= 52 * 1024) {
            a.out_tr_off = (int)smem;
            smem += ot;
        } else {
            a.out_tr = 0;
        }
    }
    // ... else, for one-tile waves writing fp16 (stage 1 of v2/48k: 10 phases x 128 channels, the block's output tile would be 87 KB): wave-private
    // transposition tiles, so that the stores are whole 128-byte lines (option UPS_TW)
    if (a.out_tw_off && !a.out_tr && a.out_half && nj == 1 && MI == 2 && a.cout % 64 == 0) {
        smem = (smem + 15) / 16 * 16;
        a.out_tw_off = (int)smem;
        smem += (size_t)4 * 32 * (MI * 64 + 16);
    } else {
        a.out_tw_off = 0;
    }
    if (nj == 4) hipLaunchKernelGGL((k_ups<OpT, CIN, MI, WV, 4>), grid, dim3(256), smem, st, a);
    else if (nj == 2) hipLaunchKernelGGL((k_ups<OpT, CIN, MI, WV, 2>), grid, dim3(256), smem, st, a);
    else hipLaunchKernelGGL((k_ups<OpT, CIN, MI, WV, 1>), grid, dim3(256), smem, st, a);
}
#define RVCMI_UPS_CASES(X, OpT) \
    X(OpT, 512, 2) X(OpT, 256, 2) X(OpT, 128, 2) X(OpT, 64, 1) X(OpT, 32, 1)
template <typename OpT>
static void launch_ups_t(const UpsArgs& a, int wv, int nj, int B, hipStream_t st) {
#define X(OpT_, CIN_, MI_)                                                        \
    if (a.cin == CIN_) {                                                          \
        if (wv == 4) return launch_ups_inst<OpT_, CIN_, MI_, 4>(a, nj, B, st);    \
        if (wv == 2) return launch_ups_inst<OpT_, CIN_, MI_, 2>(a, nj, B, st);    \
        return launch_ups_inst<OpT_, CIN_, MI_, 1>(a, nj, B, st);                 \
    }
    RVCMI_UPS_CASES(X, OpT)
#undef X
    RVCMI_FAIL(RVCMI_ERR_INVALID, "unsupported upsampler C_in %d", a.cin);
}
template <typename OpT>
static void set_lds_ups() {
#define Y(OpT_, CIN_, MI_, WV_, NJ_)                                                                                    \
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ups<OpT_, CIN_, MI_, WV_, NJ_>),                      \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#define X(OpT_, CIN_, MI_)                                                                                              \
    Y(OpT_, CIN_, MI_, 4, 4) Y(OpT_, CIN_, MI_, 4, 2) Y(OpT_, CIN_, MI_, 4, 1) Y(OpT_, CIN_, MI_, 2, 4) Y(OpT_, CIN_, MI_, 2, 2)   \
    Y(OpT_, CIN_, MI_, 2, 1) Y(OpT_, CIN_, MI_, 1, 4) Y(OpT_, CIN_, MI_, 1, 2) Y(OpT_, CIN_, MI_, 1, 1)
    RVCMI_UPS_CASES(X, OpT)
#undef Y
#undef X
}

// ---- fully fused resblock (C <= 64) ----------------------------------------------------------------
#ifndef RBF32_NJ
#define RBF32_NJ 3
#define RBF32_OCC 2
#endif
template <int C> struct RbFullGeom;
// C=64 uses 3 column tiles per wave (R = 384): x(96) + h(96) accumulators + operands stay under 512 registers without spills
// (NJ = 4 spilled ~100 registers to scratch)
#ifndef RBF64_NWV
#define RBF64_NWV 8
#endif
#if RBF64_NWV == 8
// C=64: 8 waves (2 per SIMD) x 64 rows = R 512: the co-resident wave hides the K loop's issue overhead, and the
// larger tile wastes less on overlap-save than R = 384 (x+h accumulators: 64+64 registers per wave)
template <> struct RbFullGeom<64> { static constexpr int MI = 2, NJ = 2, KG = 4, OCC = 2, NWV = 8; };
#else
template <> struct RbFullGeom<64> { static constexpr int MI = 2, NJ = 3, KG = 4, OCC = 1, NWV = 4; };
#endif
// C <= 32: R = 384 keeps the two operand tiles at 68 KB and the kernel under 256 registers => 2 blocks per CU, so one
// block's load / publish / store phases overlap the other's MFMA phases (worth more than the extra overlap-save waste)
#ifndef RBF32_NWV
#define RBF32_NWV 4
#endif
template <> struct RbFullGeom<32> { static constexpr int MI = 1, NJ = RBF32_NJ, KG = 4, OCC = RBF32_OCC, NWV = RBF32_NWV; };
template <> struct RbFullGeom<16> { static constexpr int MI = 1, NJ = RBF32_NJ, KG = 4, OCC = RBF32_OCC, NWV = RBF32_NWV; };
static int rbf_rows(int C) { return C == 64 ? RbFullGeom<64>::NWV * 32 * RbFullGeom<64>::NJ : RBF32_NWV * 32 * RBF32_NJ; }
static int rbf_nb(const Options& opt, int dflt) {  // weight-prefetch depth (register buffers); option NB overrides for A/B experiments
    const int v = opt.geti("NB", dflt);
    return v < 2 ? 2 : (v > 4 ? 4 : v);
}
template <typename OpT, int C>
static void launch_rbf_inst(const RbFullArgs& ra, int tiles, int nj, int B, hipStream_t st, const Options& opt) {
    using G = RbFullGeom<C>;
    constexpr int R = G::NWV * 32 * G::NJ;
    const size_t smem = (size_t)(R + 2 * RBF_G + R + 2 * RBF_G2) * Tile<C>::STRIDE + 3 * 2 * 32 * G::MI * 4 + 512;  // + bias vectors + dev phase stamps
    const int nb = rbf_nb(opt, G::NWV == 8 ? 2 : 3);  // the 8-wave geometry has 256 registers per wave: a 2-deep ring fits without spills
    if (nb == 2) hipLaunchKernelGGL((k_rb_full<OpT, C, G::MI, G::NJ, G::KG, 2, G::OCC, G::NWV>), dim3(tiles, nj, B), dim3(64 * G::NWV), smem, st, ra);
    else if (nb == 3) hipLaunchKernelGGL((k_rb_full<OpT, C, G::MI, G::NJ, G::KG, 3, G::OCC, G::NWV>), dim3(tiles, nj, B), dim3(64 * G::NWV), smem, st, ra);
    else hipLaunchKernelGGL((k_rb_full<OpT, C, G::MI, G::NJ, G::KG, 4, G::OCC, G::NWV>), dim3(tiles, nj, B), dim3(64 * G::NWV), smem, st, ra);
}
// C = 64 on a launch of a few tiles (a realtime chunk's 6200 rows = 44 tiles of 512 rows on 256 CUs, the launch as long as ONE k = 11 tile:
// 50 us): 256-row tiles of four waves -- 107 blocks, each half as long.  Two accumulator sets of 64 registers + a 3-deep ring: no spills.
constexpr int RBF64S_NJ = 2, RBF64S_NWV = 4, RBF64S_ROWS = RBF64S_NWV * 32 * RBF64S_NJ;
// (the same 256-row geometry for C <= 32, two blocks per CU: a chunk's 12 400 rows are 122 tiles of 384 rows, 214 of 256)
template <typename OpT, int C>
static void launch_rbf_small(const RbFullArgs& ra, int tiles, int nj, int B, hipStream_t st) {
    constexpr int R = RBF64S_ROWS, MI = C == 64 ? 2 : 1, OCCS = C == 64 ? 1 : 2;
    const size_t smem = (size_t)(R + 2 * RBF_G + R + 2 * RBF_G2) * Tile<C>::STRIDE + 3 * 2 * 32 * MI * 4 + 512;
    hipLaunchKernelGGL((k_rb_full<OpT, C, MI, RBF64S_NJ, 4, 3, OCCS, RBF64S_NWV>), dim3(tiles, nj, B), dim3(64 * RBF64S_NWV), smem, st, ra);
}
template <typename OpT>
static void launch_rbf_t(int C, const RbFullArgs& ra, int tiles, int nj, int B, hipStream_t st, const Options& opt, bool small64 = false) {
    if (small64 && C == 64) return launch_rbf_small<OpT, 64>(ra, tiles, nj, B, st);
    if (small64 && C == 32) return launch_rbf_small<OpT, 32>(ra, tiles, nj, B, st);
    switch (C) {
        case 64: return launch_rbf_inst<OpT, 64>(ra, tiles, nj, B, st, opt);
        case 32: return launch_rbf_inst<OpT, 32>(ra, tiles, nj, B, st, opt);
        case 16: return launch_rbf_inst<OpT, 16>(ra, tiles, nj, B, st, opt);
        default: RVCMI_FAIL(RVCMI_ERR_INVALID, "fused resblock: unsupported channel count %d", C);
    }
}
template <typename OpT>
static void set_lds_rbf() {
#define RBF_ATTR1(C_, NB_)                                                                                                  \
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rb_full<OpT, C_, RbFullGeom<C_>::MI, RbFullGeom<C_>::NJ, RbFullGeom<C_>::KG, NB_, RbFullGeom<C_>::OCC, RbFullGeom<C_>::NWV>), \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#define RBF_ATTR(C_) RBF_ATTR1(C_, 2) RBF_ATTR1(C_, 3) RBF_ATTR1(C_, 4)
    RBF_ATTR(64) RBF_ATTR(32) RBF_ATTR(16)
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rb_full<OpT, 64, 2, RBF64S_NJ, 4, 3, 1, RBF64S_NWV>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rb_full<OpT, 32, 1, RBF64S_NJ, 4, 3, 2, RBF64S_NWV>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#undef RBF_ATTR1
#undef RBF_ATTR
}

// Fill the common part of ConvArgs for `L` and launch it in the handle's operand mode.
// option RB_STREAM: absent = auto (streaming kernel when the strips are long enough), 1 = always when supported, 0 = never.
static int rb_stream_mode(const rvcmi_nsf* h) {
    if (!h->opt.has("RB_STREAM")) return 2;
    const int v = h->opt.geti("RB_STREAM", 2);
    return v == 0 ? 0 : (v == 1 ? 1 : 2);
}

// Option Y_F16 (default 1): the whole-ResBlock kernels (k_rb_stream ND = 3, k_rb_full) write their output streams Ya[j] as fp16
// and the next stage's k_ups / k_post read them as such (nsf_kernels.hpp pack4_h).  0 = fp32 streams (round 3).
static bool y_f16(const rvcmi_nsf* h) {
    return h->cfg.operand != RVCMI_OPERAND_F32 && h->opt.geti("Y_F16", 1) != 0;
}

// Option X0_F16 (default 1, only together with Y_F16): at the stages whose ResBlocks run on k_rb_full (C <= 64: the two HBM-heaviest
// stages of every shipped config) the ups output X0 -- read three times, once per ResBlock -- is stored as fp16 too.  Unlike the Y
// streams this rounds the START of the fp32 residual stream (relative 2^-11 once per stage; the stream itself stays fp32 inside the
// kernels): +1.1e-4 RMS on the full-clip golden, measured on the oracle and gated by the same 5e-4 test.  0 = fp32 X0.
static bool x0_f16(const rvcmi_nsf* h, int C, size_t maxnd) {
    return y_f16(h) && h->opt.geti("X0_F16", 1) != 0 && C <= 64 && maxnd <= 3 && !h->opt.on("NO_RBFULL");
}

// Whole resblocks of a stage on the streaming kernel (ND = 3).  Fills src[j] with the output streams on success.
// `xhalf`: X0 is an fp16 stream (round 5).  `dry`: plan only -- would this stage stream?  (decides BEFORE the upsampler runs whether it
// may write X0 as fp16: the tile kernels a short clip falls back to read fp32 rows)
static bool try_rb_stream_full(rvcmi_nsf* h, const Stage& s, int op, int C, int L, int B, int nk, const float** src, hipStream_t st,
                               const int* lens, int lm, bool xhalf = false, bool dry = false) {
    const bool yh = y_f16(h);
    const int mode = rb_stream_mode(h);
    if (mode == 0 || op == RVCMI_OPERAND_F32 || nk > 3) return false;
    for (int j = 0; j < nk; ++j)
        if (s.rb[j].size() != 3) return false;
    if (!rb_stream_supported(op, C, 3)) return false;
    RbStreamDesc sd[3];
    int order[3];
    for (int j = 0; j < nk; ++j) order[j] = j;
    std::sort(order, order + nk, [&](int a1, int b1) { return s.rb[a1][0].first.ntaps[0] > s.rb[b1][0].first.ntaps[0]; });
    double flops = 0, bytes = 0;
    for (int oj = 0; oj < nk; ++oj) {
        const int j = order[oj];
        RbStreamDesc& d = sd[oj];
        memset(&d, 0, sizeof(d));
        d.src = h->X0.as<float>();
        d.dst = h->Ya[j].as<float>();
        d.y_half = yh ? 1 : 0;
        d.x_half = xhalf ? 1 : 0;
        d.lens = lens;
        d.lmul = lm;
        d.k = s.rb[j][0].first.ntaps[0];
        d.k_p = s.rb[j][0].first.ntaps_p;
        d.ct1 = s.rb[j][0].first.ct_stride;
        d.ct2 = s.rb[j][0].second.ct_stride;
        for (int m = 0; m < 3; ++m) {
            const ConvLayer& c1 = s.rb[j][m].first;
            const ConvLayer& c2 = s.rb[j][m].second;
            d.w1[m] = c1.w_pack.p;
            d.w2[m] = c2.w_pack.p;
            d.b1[m] = c1.bias.as<float>();
            d.b2[m] = c2.bias.as<float>();
            d.dil[m] = c1.dstep;
            flops += (c1.flops_per_pos + c2.flops_per_pos) * (double)L * B;
            bytes += 2.0 * d.k * C * C * 2;
        }
        bytes += (double)B * L * C * ((yh ? 2 : 4) + (xhalf ? 2 : 4));
    }
    char nm[48];
    snprintf(nm, sizeof(nm), "rb_stream_c%d", C);
    if (!rb_stream_launch(op, C, 3, sd, nk, L, B, (long)L * C, mode == 1, st, h->opt, true)) return false;
    if (dry) return true;
    h->prof.launch(nm, flops, bytes, st, [&] { rb_stream_launch(op, C, 3, sd, nk, L, B, (long)L * C, mode == 1, st, h->opt); });
    HIP_CHECK(hipGetLastError());
    for (int j = 0; j < nk; ++j) src[j] = h->Ya[j].as<float>();
    return true;
}

static void run_conv(rvcmi_nsf* h, const ConvLayer& L, ConvArgs a, int B, const char* name, hipStream_t st) {
    if (!a.cf_stride) a.cf_stride = a.Lin;
    a.cin = L.cin;
    a.cout = L.cout;
    a.bias = L.bias.as<float>();
    a.dstep = L.dstep;
    a.nphase = L.nphase;
    const int op = h->cfg.operand;
    int jsum = 0;
    for (int p = 0; p < L.nphase; ++p) {
        a.ph_in_off[p] = L.in_off[p];
        a.ph_ntaps[p] = L.ntaps[p];
        a.ph_w_off[p] = op == RVCMI_OPERAND_F32 ? L.f32_off[p] : L.pack_off[p];
        jsum += L.ntaps[p];
    }
    a.in_off = L.in_off[0];
    const double flops = L.flops_per_pos * (double)a.Lq * B;
    const size_t esz = op == RVCMI_OPERAND_F32 ? 4 : 2;
    double bytes = (double)B * a.Lin * L.cin * (a.in_mode == IN_OP_RAW ? esz : 4) +
                   (double)B * a.Lq * L.nphase * L.cout * (a.out_mode == OUT_ACT ? esz : 4) * (a.accumulate ? 2 : 1) +
                   (a.res ? (double)B * a.Lq * L.nphase * L.cout * 4 : 0) + (double)jsum * L.cin * L.cout * esz;
    if (op == RVCMI_OPERAND_F32) {
        a.w = L.w_f32.p;
        a.ntaps = L.ntaps[0];
        a.roff = 0;
        a.tile_rows = 0;
        const size_t n = (size_t)a.Lq * a.cout;
        dim3 grid((unsigned)((n + 255) / 256), L.nphase, B);
        h->prof.launch(name, flops, bytes, st, [&] { hipLaunchKernelGGL(k_conv_f32, grid, dim3(256), 0, st, a); });
    } else {
        Geom g = conv_geom(L.cout);
        // a launch of a few dozen 128-row tiles (conv_pre of one clip: 20 blocks) is latency-bound on one tile's K loop:
        // 32-row tiles give 4x the blocks and a quarter of the per-block work
        int nj = 4;
        if (g.MI == 2 && g.WCO == 4 && (long)B * ((a.Lq + g.TT - 1) / g.TT) * ((L.cout + 255) / 256) * L.nphase < 128) {
            nj = 1;
            g.TT = 32;
        }
        a.w = L.w_pack.p;
        a.w_ct_stride = L.ct_stride;
        a.ntaps = L.ntaps_p;
        const int span = (L.ntaps_p - 1) * std::abs(L.dstep);
        a.roff = L.dstep < 0 ? span : 0;
        // (conv_pre of a realtime chunk through the K-split blocks was measured: 18 -> 14 us -- and removed: a second summation order for
        //  calls of at most 128 frames breaks "a ragged batch item is bit-equal to its separate call" for no pinned option)
        a.tile_rows = g.TT + span;
        h->prof.launch(name, flops, bytes, st, [&] {
            if (op == RVCMI_OPERAND_BF16) launch_conv_mfma<__bf16>(a, nj, B, st);
            else launch_conv_mfma<_Float16>(a, nj, B, st);
        });
    }
    HIP_CHECK(hipGetLastError());
}

// Up to three independent convs of the same shape class (C_in = C_out = 256) in ONE launch of k_conv_mfma_jobs, blocks of
// 32 output channels x 128 rows (MI = 1, NJ = 1, waves split time): grid = row tiles x 8 channel tiles x (jobs x batch).
template <typename OpT, int C>
static void launch_conv_jobs(const ConvJobs& js, int Lq, int B, size_t smem, hipStream_t st) {
    static std::atomic<unsigned long long> attr_done{0};
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    auto kern = &k_conv_mfma_jobs<OpT, C, 1, 1, 1>;
    if (!(attr_done.load() & (1ull << (dev & 63)))) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done.fetch_or(1ull << (dev & 63));
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)((Lq + 127) / 128), C / 32, (unsigned)(B * js.njobs)), dim3(256), smem, st, js);
}
// K-split form (nsf_kernels.hpp conv_ks_body): 32 rows x 32 channels per block, the four waves split the taps.
constexpr int conv_ks_nj(int C) { return C == 256 ? 1 : 3; }  // rows per block / 32 (C = 128: 96-row tiles = 396 blocks for a chunk's 3100 rows)
template <typename OpT, int C>
static void launch_conv_ks_jobs(const ConvJobs& js, int Lq, int B, size_t smem, hipStream_t st) {
    static std::atomic<unsigned long long> attr_done{0};
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    constexpr int NJ = conv_ks_nj(C);
    auto kern = &k_conv_ks_jobs<OpT, C, 1, NJ>;
    if (!(attr_done.load() & (1ull << (dev & 63)))) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done.fetch_or(1ull << (dev & 63));
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)((Lq + 32 * NJ - 1) / (32 * NJ)), C / 32, (unsigned)(B * js.njobs)), dim3(256), smem, st, js);
}
static void run_conv_jobs(rvcmi_nsf* h, const ConvLayer* const* Ls, const ConvArgs* as, int nj, int B, const char* name, hipStream_t st) {
    ConvJobs js;
    memset(&js, 0, sizeof(js));
    js.njobs = nj;
    double flops = 0, bytes = 0;
    size_t smem = 0;
    // K-split blocks (taps over the four waves, option CONV_KS, default on).  Realtime chunk, ABAB: C = 256 (310 rows, 32-row tiles = 240
    // blocks) six launches 134 -> 107 us; C = 128 (3100 rows) 113 -> 107 us with 96-row tiles (396 blocks) -- with 32-row tiles it LOST
    // (155 us: 1164 blocks that each stage 82 rows for 32 outputs).
    const bool ks = h->opt.geti("CONV_KS", 1) != 0;
    for (int j = 0; j < nj; ++j) {
        const ConvLayer& L = *Ls[j];
        ConvArgs a = as[j];
        if ((L.cin != 256 && L.cin != 128) || L.cout != L.cin || L.cin != Ls[0]->cin || L.nphase != 1 || L.dstep < 0)
            RVCMI_FAIL(RVCMI_ERR_INVALID, "run_conv_jobs: unsupported layer");
        a.cin = L.cin;
        a.cout = L.cout;
        a.bias = L.bias.as<float>();
        a.dstep = L.dstep;
        a.nphase = 1;
        a.in_off = L.in_off[0];
        a.w = L.w_pack.p;
        a.w_ct_stride = L.ct_stride;
        a.ntaps = L.ntaps_p;
        a.roff = 0;
        const int ksr = 32 * conv_ks_nj(L.cin);
        a.tile_rows = (ks ? ksr : 128) + (L.ntaps_p - 1) * L.dstep;
        smem = std::max(smem, (size_t)a.tile_rows * (2 * L.cin + 16) + (ks ? (size_t)3 * (ksr / 32) * 16 * 64 * 4 : 0));  // (+ the partial sums of 3 waves)
        flops += L.flops_per_pos * (double)a.Lq * B;
        bytes += (double)B * a.Lq * L.cin * (a.in_mode == IN_OP_RAW ? 2 : 4) + (double)B * a.Lq * L.cin * (a.out_mode == OUT_ACT ? 2 : 4) +
                 (a.res ? (double)B * a.Lq * L.cin * 4 : 0) + (double)L.ntaps[0] * L.cin * L.cin * 2;
        js.job[j] = a;
    }
    const bool c256 = Ls[0]->cin == 256;
    h->prof.launch(name, flops, bytes, st, [&] {
        if (ks) {
            if (h->cfg.operand == RVCMI_OPERAND_BF16) {
                if (c256) launch_conv_ks_jobs<__bf16, 256>(js, as[0].Lq, B, smem, st);
                else launch_conv_ks_jobs<__bf16, 128>(js, as[0].Lq, B, smem, st);
            } else {
                if (c256) launch_conv_ks_jobs<_Float16, 256>(js, as[0].Lq, B, smem, st);
                else launch_conv_ks_jobs<_Float16, 128>(js, as[0].Lq, B, smem, st);
            }
        } else if (h->cfg.operand == RVCMI_OPERAND_BF16) {
            if (c256) launch_conv_jobs<__bf16, 256>(js, as[0].Lq, B, smem, st);
            else launch_conv_jobs<__bf16, 128>(js, as[0].Lq, B, smem, st);
        } else {
            if (c256) launch_conv_jobs<_Float16, 256>(js, as[0].Lq, B, smem, st);
            else launch_conv_jobs<_Float16, 128>(js, as[0].Lq, B, smem, st);
        }
    });
    HIP_CHECK(hipGetLastError());
}

// option DBG: timing-ablation bit mask forwarded to the kernels (results are WRONG when non-zero; bench/dev only).
static int dbg_flags(const rvcmi_nsf* h) { return h->opt.geti("DBG", 0); }

static ConvArgs base_args() {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.slope_in = 0.1f;
    a.div_in = 1.f;
    a.slope_out = 0.1f;
    a.out_mul = 1;
    a.nphase = 1;
    return a;
}

struct TapRequest {
    const char* what = nullptr;
    float* out_host = nullptr;
    size_t capacity = 0;
    int64_t* shape = nullptr;
    bool done = false;
};

static void copy_tap_cl(rvcmi_nsf* h, const float* src_cl, int B, int L, int C, TapRequest* tr, hipStream_t st) {
    const size_t n = (size_t)B * L * C;
    if (n > tr->capacity) RVCMI_FAIL(RVCMI_ERR_NOMEM, "tap needs %zu floats, capacity %zu", n, tr->capacity);
    if (h->dbg.bytes < n * 4) h->dbg.alloc(n * 4);
    dim3 grid((unsigned)(((size_t)L * C + 255) / 256), B);
    hipLaunchKernelGGL(k_cl_to_cf, grid, dim3(256), 0, st, src_cl, h->dbg.as<float>(), L, C);
    HIP_CHECK(hipMemcpyAsync(tr->out_host, h->dbg.p, n * 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    tr->shape[0] = B;
    tr->shape[1] = C;
    tr->shape[2] = L;
    tr->done = true;
}

// `lens`: optional device array [B] of valid frames per item (1 <= lens[b] <= T): a ragged batch, every item computed exactly as
// a separate call of its own length (nsf_kernels.hpp item_rows); the output rows behind an item's end are zero.
static void nsf_forward(rvcmi_nsf* h, int B, int T, const int* lens, const float* x, const float* f0, const float* g,
                        const float* noise, int n_res, float* out, hipStream_t st, TapRequest* tr) {
    if (!h || !x || (!out && !tr)) RVCMI_FAIL(RVCMI_ERR_INVALID, "null argument");
    if (lens && n_res >= 0) RVCMI_FAIL(RVCMI_ERR_INVALID, "lengths and n_res (the realtime interpolation) cannot be combined");
    const rvcmi_nsf_config& c = h->cfg;
    if (c.use_f0 && !f0) RVCMI_FAIL(RVCMI_ERR_INVALID, "f0 is required for an NSF (use_f0) generator");
    if (B < 1 || T < 1) RVCMI_FAIL(RVCMI_ERR_INVALID, "B and T must be positive");
    const int Te = n_res >= 0 ? n_res : T;
    if (Te < 1) RVCMI_FAIL(RVCMI_ERR_INVALID, "n_res must be positive");
    if (B > h->max_B || T > h->max_T || Te > h->max_T)
        RVCMI_FAIL(RVCMI_ERR_NOMEM, "shape B=%d T=%d n_res=%d exceeds handle limits (max_B=%d max_T=%d)", B, T, n_res,
                   h->max_B, h->max_T);
    HIP_CHECK(hipSetDevice(h->device));
    const int upp = h->upp, inter = c.inter_channels, C0 = h->C0;
    const int op = c.operand;
    auto want = [&](const char* w) { return tr && strcmp(tr->what, w) == 0; };

    // ---- excitation: har[B][Te*upp]                                           nsf.py:152-153
    const float* har = nullptr;
    if (c.use_f0) {
        h->prof.launch("phase_scan", 0, (double)B * T * 8, st, [&] {
            hipLaunchKernelGGL(k_phase_scan, dim3(B), dim3(256), 0, st, f0, h->phase.as<float>(), T, (float)c.sr, (float)upp, lens);
        });
        const size_t total = (size_t)B * T * upp;
        h->prof.launch("sine_source", 0, (double)total * (noise ? 8 : 4), st, [&] {
            hipLaunchKernelGGL(k_sine_source, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, f0,
                               h->phase.as<float>(), noise, h->har.as<float>(), T, upp, (float)c.sr, h->lin_w, h->lin_b, total);
        });
        har = h->har.as<float>();
        if (n_res >= 0 && Te != T) {  // nsf.py:155-159
            const size_t tot2 = (size_t)B * Te * upp;
            hipLaunchKernelGGL(k_interp_linear, dim3((unsigned)((tot2 + 255) / 256)), dim3(256), 0, st, har,
                               h->har2.as<float>(), T * upp, Te * upp, tot2);
            har = h->har2.as<float>();
        }
        HIP_CHECK(hipGetLastError());
        if (want("har")) {
            const size_t n = (size_t)B * Te * upp;
            if (n > tr->capacity) RVCMI_FAIL(RVCMI_ERR_NOMEM, "tap capacity");
            HIP_CHECK(hipMemcpyAsync(tr->out_host, har, n * 4, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            tr->shape[0] = B; tr->shape[1] = 1; tr->shape[2] = (int64_t)Te * upp;
            tr->done = true;
            return;
        }
    }
    if (n_res >= 0 && Te != T) {  // nsf.py:160-162 / generators.py:76-79
        const size_t tot = (size_t)B * inter * Te;
        hipLaunchKernelGGL(k_interp_linear, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, x, h->x2.as<float>(), T, Te, tot);
        x = h->x2.as<float>();
    }
    // ---- conv_pre + cond                                                       nsf.py:164-166
    if (g && c.gin_channels) {
        hipLaunchKernelGGL(k_cond, dim3((C0 + 3) / 4, B), dim3(256), 0, st, g, h->cond_w.as<float>(),
                           h->cond_b.as<float>(), h->condv.as<float>(), c.gin_channels, C0);
    }
    float* P = h->P.as<float>();
    // round 6: on the MFMA path conv_pre writes the ACTIVATED OPERANDS to_op(lrelu(conv + bias + cond, 0.1)) -- the only thing stage 0's upsampler does
    // with P is to compute exactly that while staging (12 blocks per time tile each re-read and re-converted the fp32 rows) -- unless the fp32 "pre" tap is asked for
    const bool pre_op = op != RVCMI_OPERAND_F32 && !want("pre") && h->opt.geti("PRE_OP", 1) != 0;
    {
        ConvArgs a = base_args();
        a.in = x;
        a.in_bstride = (long)inter * Te;
        a.Lin = Te;
        a.in_mode = IN_F32_CF;
        a.Lq = Te;
        a.out_mode = pre_op ? OUT_ACT : OUT_F32;
        a.out = P;
        a.out_bstride = (long)Te * C0;
        a.out_C = C0;
        a.cb = (g && c.gin_channels) ? h->condv.as<float>() : nullptr;
        a.lens = lens;
        a.lmul_in = a.lmul_q = 1;
        run_conv(h, h->pre, a, B, "conv_pre", st);
    }
    if (want("pre")) return copy_tap_cl(h, P, B, Te, C0, tr, st);

    // The running activation is ((y[0] + y[1]) + y[2]) / div  (nsf.py:177-186); stage 0 reads P alone.
    const float* y[3] = {P, nullptr, nullptr};
    bool yhalf = false;  // y[] are fp16 streams (written by k_rb_stream ND = 3 / k_rb_full under option Y_F16)
    float div = 1.f;
    long L = Te;
    int Cprev = C0;
    int lm = 1;  // rows per frame at the current stage (ragged batches: item rows = lens[b] * lm)
    for (int i = 0; i < c.n_ups; ++i) {
        Stage& s = h->stages[i];
        const long Lin = L;
        const int lm_prev = lm;
        lm *= s.u;
        L = Lin * s.u;
        const int C = s.cout;
        const int nk = (int)s.rb.size();
        size_t maxnd0 = 0;
        for (int j = 0; j < nk; ++j) maxnd0 = std::max(maxnd0, s.rb[j].size());
        // X0 of this stage is fp16: where the consumer is k_rb_full (C <= 64), and -- round 5 -- where the whole resblocks WILL run on the
        // streaming kernel with its coalesced step IO (C = 128 on clips long enough for strips; decided by a dry run of the planner)
        // ... decided HERE, once, together with the branch the ResBlocks will take below: the conv-by-conv path for a handful of rows reads X0
        // as fp32 (IN_F32_ACT) and comes BEFORE the streaming kernel in that chain, so it must veto the fp16 X0 explicitly (ADVICE round 5:
        // the two could not coincide only because the planner happens to refuse such short launches)
        const bool rb_split = op != RVCMI_OPERAND_F32 &&
                              ((C == 256 && (long)B * L <= RB_SPLIT_MAX_ROWS) || (C == 128 && (long)B * L <= RB_SPLIT_MAX_ROWS_128 && !h->opt.on("NO_RB_SPLIT128"))) &&
                              nk <= 3 && !h->opt.on("NO_RB_SPLIT") && rb_stream_mode(h) != 1;
        // (round 6, review item 4: the same conv-by-conv launches for a FULL-length stage 0 with 64-channel x 256-row blocks -- 684 blocks per launch, no halo
        //  recompute -- were built as a dev option, parity-green, and measured 0.61 ms against k_rb_pair<256>'s 0.284 ms; removed)
        const bool x0h_stream = op != RVCMI_OPERAND_F32 && !rb_split && C > 64 && y_f16(h) && h->opt.geti("X0_F16", 1) != 0 && !h->opt.on("X0_F16_NOSTREAM") &&
                                try_rb_stream_full(h, s, op, C, (int)L, B, nk, nullptr, st, lens, lm, true, true);
        const bool x0h = op != RVCMI_OPERAND_F32 && (x0_f16(h, C, maxnd0) || x0h_stream);
        char nm[48];
        if (op == RVCMI_OPERAND_F32) {
            {  // x = ups[i](leaky_relu(x, 0.1))                                nsf.py:171-172
                ConvArgs a = base_args();
                a.in = y[0];
                a.in_b = y[1];
                a.in_c = y[2];
                a.in_bstride = Lin * Cprev;
                a.Lin = (int)Lin;
                a.in_mode = IN_F32_ACT;
                a.div_in = div;
                a.Lq = (int)Lin;
                a.out_mode = OUT_F32;
                a.out = h->X0.p;
                a.out_bstride = L * C;
                a.out_C = C;
                a.out_mul = s.u;
                a.lens = lens;
                a.lmul_in = a.lmul_q = lm_prev;
                snprintf(nm, sizeof(nm), "ups_c%d", s.cin);
                run_conv(h, s.up, a, B, nm, st);
            }
            if (c.use_f0) {  // x = x + noise_convs[i](har)                       nsf.py:173-174
                const size_t n = (size_t)L * C;
                snprintf(nm, sizeof(nm), "noise_conv_c%d", C);
                h->prof.launch(nm, 2.0 * s.nk * n * B, (double)B * n * 8, st, [&] {
                    hipLaunchKernelGGL(k_noise_add, dim3((unsigned)((n + 255) / 256), 1, B), dim3(256), 0, st, h->X0.as<float>(),
                                       har, s.noise_w.as<float>(), s.noise_b.as<float>(), (int)L, C, Te * upp, s.nk, s.ns, s.npad, lens, lm, upp);
                });
            }
        } else {  // ups + bias + noise conv in one MFMA kernel, X0 written once      nsf.py:171-174
            const ConvLayer& U = s.up;
            UpsArgs ua;
            memset(&ua, 0, sizeof(ua));
            ua.dbg = dbg_flags(h);
            ua.bias_off = h->opt.geti("UPS_BL", 1) != 0 ? 1 : 0;  // (the launcher turns it into the LDS offset)
            ua.in_a = y[0];
            ua.in_b = y[1];
            ua.in_c = y[2];
            ua.in_half = yhalf ? 1 : 0;
            ua.in_raw = (i == 0 && pre_op) ? 1 : 0;  // (stage 0: y[0] = P holds operands already)
            ua.lens = lens;
            ua.lmul = lm_prev;
            ua.lhmul = upp;
            ua.Lh_stride = Te * upp;
            ua.div = div;
            ua.Lin = (int)Lin;
            ua.cin = s.cin;
            ua.in_bstride = Lin * Cprev;
            ua.w = U.w_pack.p;
            ua.ct_stride = U.ct_stride;
            int lo = 1 << 30, hi = -(1 << 30);
            for (int r = 0; r < s.u; ++r) {
                ua.ph_w_off[r] = U.pack_off[r];
                ua.ph_in_off[r] = U.in_off[r];
                lo = std::min(lo, U.in_off[r] - (U.ntaps_p - 1));
                hi = std::max(hi, U.in_off[r]);
            }
            ua.ntaps_p = U.ntaps_p;
            ua.u = s.u;
            ua.cout = C;
            ua.lo = lo;
            ua.bias = U.bias.as<float>();
            ua.out = h->X0.as<float>();
            ua.out_half = x0h ? 1 : 0;
            ua.out_tr = h->opt.geti("UPS_TR", 1) != 0 ? 1 : 0;  // (the launcher clears it where the block does not own whole rows)
            ua.out_tw_off = h->opt.geti("UPS_TW", 1) != 0 ? 1 : 0;  // (the launcher turns it into the LDS offset where it applies)
            ua.out_bstride = L * C;
            if (c.use_f0 && s.nz_mfma) {  // noise_convs[i](har) as a 2-tap MFMA conv over frames -> NZ, added in k_ups' epilogue
                ConvArgs na = base_args();
                na.in = har;
                na.in_bstride = (long)Te * upp;
                na.Lin = Te * upp;
                na.in_mode = IN_HAR;
                na.hs = s.ns;
                na.hpad = s.npad;
                na.Lq = (int)L;
                na.out_mode = OUT_F32;
                na.out = h->NZ.p;
                na.out_bstride = L * C;
                na.out_C = C;
                na.lens = lens;
                na.lmul_in = upp;
                na.lmul_q = lm;
                snprintf(nm, sizeof(nm), "noise_mfma_c%d", C);
                run_conv(h, s.nz, na, B, nm, st);
                ua.addend = h->NZ.as<float>();
            } else if (c.use_f0) {
                ua.nz_k1 = s.nz_k1 ? 1 : 0;
                ua.wnz = s.nz_k1_w.p;
                ua.har = har;
                ua.Lh = Te * upp;
                ua.Wn = s.noise_w.as<float>();
                ua.bn = s.noise_b.as<float>();
                ua.nk = s.nk;
                ua.ns = s.ns;
                ua.npad = s.npad;
            }
            const int MIu = C >= 64 ? 2 : 1;
            ua.cog = (C + 32 * MIu - 1) / (32 * MIu);
            ua.nvt = s.u * ua.cog;
            const int wv = ua.nvt >= 4 ? 4 : (ua.nvt >= 2 ? 2 : 1);
            // time-tile height: blocks have equal duration, so a launch of ~1.1 x (resident blocks) runs two rounds for one
            // round of work; pick the largest tile that still gives >= 4 rounds of blocks (2 resident blocks per CU)
            // (measured at B = 1, same box, us: C_in 512: 52 / 36 / 37 for NJ 4 / 2 / 1; 256: 94 / 95 / 85; 128: 131 / 118 / 110;
            //  64 -> 32 channels: 98 / 102 / 150 -- with one 32-channel tile per wave, short tiles starve the MFMA of columns)
            int nj = 4;
            const long blocks4 = (long)((Lin + 128 * (4 / wv) - 1) / (128 * (4 / wv))) * B;
            if (blocks4 < 2048 && s.cin >= 128) nj = s.cin >= 512 ? 2 : 1;
            else if (blocks4 < 2048 && s.cin == 64) nj = 2;  // (us at B = 1 after the staging fix: 84 / 75 / 89 for NJ 4 / 2 / 1)
            // large grids (round 4, fp16 inputs + row-wise stores; us per clip at B = 16 for NJ 4 / 1 / 2, ABAB: C_in 128: 77 / 67 / 83,
            // 64: 70 / 73 / 55): the short tiles win there too -- they also keep the output tile inside the row-wise store's LDS budget
            else if (s.cin >= 512) nj = 2;   // (C_in 512: 22 vs 29 us for NJ 2 / 4; C_in 256 keeps NJ = 4: 52 vs 57)
            else if (s.cin == 128) nj = 1;
            else if (s.cin == 64) nj = 2;
            if (h->opt.has("UPS_NJ")) {
                const int v = h->opt.geti("UPS_NJ", nj);
                if (v == 1 || v == 2 || v == 4) nj = v;
            }
            // (round 6: NJ 2 / 4 and vpw 1 / 2 for the C_in = 256 stage alone, 8 variants x 2, ABAB: 72-110 us against 60 -- every one loses)
            const int TQ = 32 * nj * (4 / wv);
            ua.tile_rows = TQ + (hi - lo);
            const long qtiles = (Lin + TQ - 1) / TQ;
            int vpw = (ua.nvt + wv - 1) / wv;  // everything in one block ...
            while (vpw > 1 && qtiles * B * ((ua.nvt + wv * vpw - 1) / (wv * vpw)) < 300) --vpw;  // ... unless the grid would starve
            ua.vpw = vpw;
            snprintf(nm, sizeof(nm), "ups_c%d", s.cin);
            const double flops = U.flops_per_pos * (double)Lin * B + 2.0 * s.nk * (double)L * C * B;
            const double bytes = (double)B * Lin * Cprev * (yhalf ? 2 : 4) * (y[1] ? (y[2] ? 3 : 2) : 1) + (double)B * L * C * (x0h ? 2 : 4);
            h->prof.launch(nm, flops, bytes, st, [&] {
                if (op == RVCMI_OPERAND_BF16) launch_ups_t<__bf16>(ua, wv, nj, B, st);
                else launch_ups_t<_Float16>(ua, wv, nj, B, st);
            });
            HIP_CHECK(hipGetLastError());
        }
        snprintf(nm, sizeof(nm), "up%d", i);
        if (want(nm)) {
            if (x0h) {  // widen the fp16 X0 for the tap
                const size_t n = (size_t)B * L * C;
                hipLaunchKernelGGL(k_h2f, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const _Float16*)h->X0.p, h->Ya[0].as<float>(), n);
                return copy_tap_cl(h, h->Ya[0].as<float>(), B, (int)L, C, tr, st);
            }
            return copy_tap_cl(h, h->X0.as<float>(), B, (int)L, C, tr, st);
        }

        // resblocks[i*nk + j](x), j < nk                                         nsf.py:175-185
        size_t maxnd = 0;
        for (int j = 0; j < nk; ++j) maxnd = std::max(maxnd, s.rb[j].size());
        const float* src[RVCMI_MAX_RB];
        for (int j = 0; j < nk; ++j) src[j] = h->X0.as<float>();
        bool stage_half = false;  // this stage's resblock outputs are fp16 streams
        if (op == RVCMI_OPERAND_F32) {
            snprintf(nm, sizeof(nm), "rb_c%d", C);
            for (int j = 0; j < nk; ++j)
                for (size_t m = 0; m < s.rb[j].size(); ++m) {  // residuals.py:73-82
                    float* dst = (m & 1) ? h->Yb[j].as<float>() : h->Ya[j].as<float>();
                    ConvArgs a = base_args();
                    a.in = src[j];
                    a.in_bstride = L * C;
                    a.Lin = (int)L;
                    a.in_mode = IN_F32_ACT;
                    a.Lq = (int)L;
                    a.lens = lens;
                    a.lmul_in = a.lmul_q = lm;
                    a.out_mode = OUT_ACT;
                    a.out = h->H.p;
                    a.out_bstride = L * C;
                    a.out_C = C;
                    run_conv(h, s.rb[j][m].first, a, B, nm, st);
                    a = base_args();
                    a.in = h->H.p;
                    a.in_bstride = L * C;
                    a.Lin = (int)L;
                    a.in_mode = IN_OP_RAW;
                    a.Lq = (int)L;
                    a.lens = lens;
                    a.lmul_in = a.lmul_q = lm;
                    a.out_mode = OUT_F32;
                    a.out = dst;
                    a.out_bstride = L * C;
                    a.out_C = C;
                    a.res = src[j];
                    a.res_bstride = L * C;
                    run_conv(h, s.rb[j][m].second, a, B, nm, st);
                    src[j] = dst;
                }
        } else if (rb_split) {
            // a handful of rows (realtime chunk): conv1 / conv2 of all resblocks as two output-channel-split launches per pair level
            snprintf(nm, sizeof(nm), "rb_split_c%d", C);
            for (size_t m = 0; m < maxnd; ++m) {
                const ConvLayer* L1[3];
                const ConvLayer* L2[3];
                ConvArgs a1[3], a2[3];
                int nj = 0;
                for (int j = 0; j < nk; ++j) {
                    if (m >= s.rb[j].size()) continue;
                    float* dst = (m & 1) ? h->Yb[j].as<float>() : h->Ya[j].as<float>();
                    void* hbuf = (char*)h->H.p + (size_t)nj * (h->H.bytes / 3);
                    ConvArgs a = base_args();
                    a.in = src[j];
                    a.in_bstride = L * C;
                    a.Lin = (int)L;
                    a.in_mode = IN_F32_ACT;
                    a.Lq = (int)L;
                    a.lens = lens;
                    a.lmul_in = a.lmul_q = lm;
                    a.out_mode = OUT_ACT;
                    a.out = hbuf;
                    a.out_bstride = L * C;
                    a.out_C = C;
                    a1[nj] = a;
                    L1[nj] = &s.rb[j][m].first;
                    a = base_args();
                    a.in = hbuf;
                    a.in_bstride = L * C;
                    a.Lin = (int)L;
                    a.in_mode = IN_OP_RAW;
                    a.Lq = (int)L;
                    a.lens = lens;
                    a.lmul_in = a.lmul_q = lm;
                    a.out_mode = OUT_F32;
                    a.out = dst;
                    a.out_bstride = L * C;
                    a.out_C = C;
                    a.res = src[j];
                    a.res_bstride = L * C;
                    a2[nj] = a;
                    L2[nj] = &s.rb[j][m].second;
                    src[j] = dst;
                    ++nj;
                }
                if (!nj) continue;
                run_conv_jobs(h, L1, a1, nj, B, nm, st);
                run_conv_jobs(h, L2, a2, nj, B, nm, st);
            }
        } else if (try_rb_stream_full(h, s, op, C, (int)L, B, nk, src, st, lens, lm, x0h_stream)) {
            // streaming fused resblocks (rb_stream_kernels.hpp): persistent blocks walk strips of the time axis
            stage_half = y_f16(h);
        } else if (C <= 64 && maxnd <= 3 && !h->opt.on("NO_RBFULL")) {
            // whole resblocks fused (x resident in registers): ONE launch for the stage   residuals.py:68-85
            snprintf(nm, sizeof(nm), "rb_full_c%d", C);
            RbFullArgs ra;
            memset(&ra, 0, sizeof(ra));
            ra.L = (int)L;
            ra.lens = lens;
            ra.lmul = lm;
            ra.bstride = L * C;
            ra.dbg = dbg_flags(h);
            stage_half = y_f16(h);
            ra.yh = stage_half ? 1 : 0;
            ra.xh = x0h ? 1 : 0;
            // (C = 64) short launches take 256-row tiles: at 512 rows they would be fewer blocks than half the CUs (halo of the k = 11
            // resblock: 120 rows either way -- the small tiles keep 136 of 256 rows; option RBF_SMALL 0 / 1 pins the choice)
            bool small64 = false;
            if (C == 64 || C == 32) {
                long blocks512 = 0;
                for (int j = 0; j < nk; ++j) {
                    int dsum = 0;
                    for (size_t m = 0; m < s.rb[j].size(); ++m) dsum += s.rb[j][m].first.dstep;
                    const int k = s.rb[j][0].first.ntaps[0];
                    const int HL = (k - 1) / 2 * (dsum + (int)s.rb[j].size());
                    blocks512 += (L + (rbf_rows(C) - 2 * HL) - 1) / (rbf_rows(C) - 2 * HL) * B;
                    if (RBF64S_ROWS - 2 * HL < RBF64S_ROWS / 4) blocks512 = 1 << 30;  // (halo too large for the small tile)
                }
                small64 = h->opt.has("RBF_SMALL") ? h->opt.geti("RBF_SMALL", 0) != 0 && blocks512 < (1 << 30) : blocks512 < num_cus() / 2 * (C == 64 ? 1 : 2);
            }
            const int R = small64 ? RBF64S_ROWS : rbf_rows(C);
            int order[RVCMI_MAX_RB];
            for (int j = 0; j < nk; ++j) order[j] = j;
            std::sort(order, order + nk, [&](int a1, int b1) { return s.rb[a1][0].first.ntaps[0] > s.rb[b1][0].first.ntaps[0]; });
            int max_tiles = 0;
            double flops = 0, bytes = 0;
            for (int oj = 0; oj < nk; ++oj) {
                const int j = order[oj];
                RbFullJob& J = ra.job[oj];
                J.src = h->X0.as<float>();
                J.dst = h->Ya[j].as<float>();
                J.nd = (int)s.rb[j].size();
                J.k = s.rb[j][0].first.ntaps[0];
                J.k_p = s.rb[j][0].first.ntaps_p;
                J.ct1 = s.rb[j][0].first.ct_stride;
                J.ct2 = s.rb[j][0].second.ct_stride;
                int dsum = 0;
                for (int m = 0; m < J.nd; ++m) {
                    const ConvLayer& c1 = s.rb[j][m].first;
                    const ConvLayer& c2 = s.rb[j][m].second;
                    J.w1[m] = c1.w_pack.p;
                    J.w2[m] = c2.w_pack.p;
                    J.b1[m] = c1.bias.as<float>();
                    J.b2[m] = c2.bias.as<float>();
                    J.dil[m] = c1.dstep;
                    dsum += c1.dstep;
                    flops += (c1.flops_per_pos + c2.flops_per_pos) * (double)L * B;
                    bytes += 2.0 * J.k * C * C * 2;
                    if (c1.dstep * (J.k - 1) / 2 + c1.dstep > RBF_G) RVCMI_FAIL(RVCMI_ERR_INVALID, "dilation too large for the fused resblock");
                }
                J.HL = (J.k - 1) / 2 * (dsum + J.nd);
                J.tvalid = R - 2 * J.HL;
                if (J.tvalid < R / 4) RVCMI_FAIL(RVCMI_ERR_INVALID, "resblock halo too large for the fused kernel");
                J.ntiles = (int)((L + J.tvalid - 1) / J.tvalid);
                max_tiles = std::max(max_tiles, J.ntiles);
                bytes += (double)B * L * C * ((stage_half ? 2 : 4) + (x0h ? 2 : 4));
                src[j] = J.dst;
            }
            const size_t nblk = (size_t)max_tiles * nk * B;
            if (ra.dbg & 32) {
                if (h->dbg.bytes < nblk * 8 * 16 * 8) h->dbg.alloc(nblk * 8 * 16 * 8);
                HIP_CHECK(hipMemsetAsync(h->dbg.p, 0, nblk * 8 * 16 * 8, st));
                ra.ts = h->dbg.as<unsigned long long>();
            }
            h->prof.launch(nm, flops, bytes, st, [&] {
                if (op == RVCMI_OPERAND_BF16) launch_rbf_t<__bf16>(C, ra, max_tiles, nk, B, st, h->opt, small64);
                else launch_rbf_t<_Float16>(C, ra, max_tiles, nk, B, st, h->opt, small64);
            });
            HIP_CHECK(hipGetLastError());
            if (ra.dbg & 32) {  // dev only: per-phase cycle breakdown, averaged per resblock kernel size
                HIP_CHECK(hipStreamSynchronize(st));
                const int nwv = small64 ? RBF64S_NWV : (C == 64 ? RbFullGeom<64>::NWV : RBF32_NWV);  // waves per block: the stamp rows of a tile
                std::vector<unsigned long long> ts(nblk * nwv * 16);
                HIP_CHECK(hipMemcpy(ts.data(), h->dbg.p, ts.size() * 8, hipMemcpyDeviceToHost));
                for (int oj = 0; oj < nk; ++oj) {
                    double sum[16] = {0};
                    long cnt = 0;
                    for (int t = 0; t < ra.job[oj].ntiles; ++t)
                        for (int w = 0; w < nwv; ++w) {
                            const unsigned long long* p = &ts[(((size_t)(0 * nk + oj) * max_tiles + t) * nwv + w) * 16];
                            if (!p[0]) continue;
                            for (int i = 1; i < 16; ++i) sum[i] += p[i] ? (double)(p[i] - p[i - 1]) : 0.0;
                            ++cnt;
                        }
                    fprintf(stderr, "[rvcmi ts] %s k=%d tiles=%d:", nm, ra.job[oj].k, ra.job[oj].ntiles);
                    for (int i = 1; i < 16; ++i) fprintf(stderr, " %.0f", cnt ? sum[i] / cnt : 0.0);
                    fprintf(stderr, "\n");
                }
            }
        } else {
            snprintf(nm, sizeof(nm), "rb_pair_c%d", C);
            // Launch order.  Level-major (one launch = pair level m of all resblocks) maximises the grid; resblock-major
            // (one launch = one pair of ONE resblock, chains run back to back) keeps a chain's src+dst (2 x L*C*4 B) inside
            // the 256 MB Infinity Cache when L*C*4*6 does not fit -- chosen per stage by working-set size.
            const double ws_level = 6.0 * (double)B * L * C * 4;
            const bool jmajor = h->opt.on("RB_ORDER");
            (void)ws_level;
            const size_t npass = jmajor ? (size_t)nk * maxnd : maxnd;
            for (size_t pass = 0; pass < npass; ++pass) {
                const size_t m = jmajor ? pass % maxnd : pass;
                const int only_j = jmajor ? (int)(pass / maxnd) : -1;
                RbPairArgs ra;
                memset(&ra, 0, sizeof(ra));
                ra.lens = lens;
                ra.lmul = lm;
                ra.L = (int)L;
                ra.bstride = L * C;
                ra.dbg = dbg_flags(h);
                int nj = 0, max_tiles = 0, max_rows = 0;
                double flops = 0, bytes = 0;
                // heaviest kernel size first so that the long blocks are dispatched first
                int order[RVCMI_MAX_RB];
                for (int j = 0; j < nk; ++j) order[j] = j;
                std::sort(order, order + nk, [&](int a1, int b1) {
                    return s.rb[a1][0].first.ntaps[0] > s.rb[b1][0].first.ntaps[0];
                });
                float* dsts[RVCMI_MAX_RB] = {nullptr};
                for (int oj = 0; oj < nk; ++oj) {
                    const int j = order[oj];
                    if (m >= s.rb[j].size()) continue;
                    if (only_j >= 0 && j != only_j) continue;
                    const ConvLayer& c1 = s.rb[j][m].first;
                    const ConvLayer& c2 = s.rb[j][m].second;
                    RbJob& J = ra.job[nj++];
                    float* dst = (m & 1) ? h->Yb[j].as<float>() : h->Ya[j].as<float>();
                    dsts[j] = dst;
                    J.src = src[j];
                    J.dst = dst;
                    J.w1 = c1.w_pack.p;
                    J.w2 = c2.w_pack.p;
                    J.b1 = c1.bias.as<float>();
                    J.b2 = c2.bias.as<float>();
                    J.ct1 = c1.ct_stride;
                    J.ct2 = c2.ct_stride;
                    J.k = c1.ntaps[0];
                    J.k_p = c1.ntaps_p;
                    J.dil = c1.dstep;
                    J.tt2 = rb_rows(C) - (J.k - 1);
                    J.ntiles = (int)((L + J.tt2 - 1) / J.tt2);
                    max_tiles = std::max(max_tiles, J.ntiles);
                    max_rows = std::max(max_rows, rb_rows(C) + (J.k_p - 1) * J.dil);
                    flops += (c1.flops_per_pos + c2.flops_per_pos) * (double)L * B;
                    bytes += (double)B * L * C * 8 + 2.0 * J.k * C * C * 2;
                }
                if (nj == 0) continue;
                const int NWp = 8;  // upper bound of waves per block across the pair-kernel geometries
                const size_t nblk = (size_t)max_tiles * nj * B;
                if (ra.dbg & 32) {
                    if (h->dbg.bytes < nblk * NWp * 64) h->dbg.alloc(nblk * NWp * 64);
                    HIP_CHECK(hipMemsetAsync(h->dbg.p, 0, nblk * NWp * 64, st));
                    ra.ts = h->dbg.as<unsigned long long>();
                }
                bool streamed = false;
                if (rb_stream_mode(h) != 0 && rb_stream_supported(op, C, 1)) {
                    RbStreamDesc sd[RVCMI_MAX_RB];
                    for (int q = 0; q < nj; ++q) {
                        const RbJob& J = ra.job[q];
                        memset(&sd[q], 0, sizeof(sd[q]));
                        sd[q].src = J.src; sd[q].dst = J.dst; sd[q].w1[0] = J.w1; sd[q].w2[0] = J.w2; sd[q].b1[0] = J.b1; sd[q].b2[0] = J.b2;
                        sd[q].ct1 = J.ct1; sd[q].ct2 = J.ct2; sd[q].k = J.k; sd[q].k_p = J.k_p; sd[q].dil[0] = J.dil;
                        sd[q].lens = lens; sd[q].lmul = lm;
                    }
                    char nms[48];
                    snprintf(nms, sizeof(nms), "rb_stream1_c%d", C);
                    const bool force = rb_stream_mode(h) == 1;
                    if (rb_stream_launch(op, C, 1, sd, nj, (int)L, B, L * C, force, st, h->opt, true))
                        h->prof.launch(nms, flops, bytes, st, [&] { streamed = rb_stream_launch(op, C, 1, sd, nj, (int)L, B, L * C, force, st, h->opt); });
                }
                if (!streamed) h->prof.launch(nm, flops, bytes, st, [&] { launch_rb_pair(op, C, ra, max_tiles, nj, B, max_rows, st); });
                if ((ra.dbg & 32) && m == 0) {  // dev only: per-phase cycles of pair level 0, per kernel size
                    HIP_CHECK(hipStreamSynchronize(st));
                    std::vector<unsigned long long> ts(nblk * NWp * 8);
                    HIP_CHECK(hipMemcpy(ts.data(), h->dbg.p, ts.size() * 8, hipMemcpyDeviceToHost));
                    for (int oj = 0; oj < nj; ++oj) {
                        double sum[8] = {0};
                        long cnt = 0;
                        for (int t = 0; t < ra.job[oj].ntiles; ++t)
                            for (int w = 0; w < NWp; ++w) {
                                const unsigned long long* p = &ts[(((size_t)oj * max_tiles + t) * NWp + w) * 8];
                                if (!p[0]) continue;
                                for (int i = 1; i < 8; ++i) sum[i] += (double)(p[i] - p[i - 1]);
                                ++cnt;
                            }
                        fprintf(stderr, "[rvcmi ts] %s k=%d tiles=%d:", nm, ra.job[oj].k, ra.job[oj].ntiles);
                        for (int i = 1; i < 8; ++i) fprintf(stderr, " %.0f", cnt ? sum[i] / cnt : 0.0);
                        fprintf(stderr, "\n");
                    }
                }
                for (int j = 0; j < nk; ++j)
                    if (dsts[j]) src[j] = dsts[j];
            }
            HIP_CHECK(hipGetLastError());
        }
        for (int j = 0; j < 3; ++j) y[j] = j < nk ? src[j] : nullptr;
        yhalf = stage_half;
        div = (float)nk;  // x = xs / num_kernels (nsf.py:186) is applied by the consumer
        Cprev = C;
        snprintf(nm, sizeof(nm), "stage%d", i);
        if (want(nm)) {  // the un-divided sum
            const size_t n = (size_t)B * L * C;
            if (yhalf)
                hipLaunchKernelGGL(k_sum3h, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const _Float16*)y[0], (const _Float16*)y[1],
                                   (const _Float16*)y[2], h->X0.as<float>(), n);
            else
            hipLaunchKernelGGL(k_sum3, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, y[0], y[1], y[2], h->X0.as<float>(), n);
            return copy_tap_cl(h, h->X0.as<float>(), B, (int)L, C, tr, st);
        }
    }
    if (tr) RVCMI_FAIL(RVCMI_ERR_INVALID, "unknown tap '%s'", tr->what);
    // ---- x = tanh(conv_post(leaky_relu(x)))                                    nsf.py:187-189
    const int nkk = c.n_resblock_kernels;
    h->prof.launch("conv_post", 2.0 * 7 * Cprev * (double)L * B, (double)B * L * (Cprev * nkk * (yhalf ? 2 : 4) + 4), st, [&] {
        // round 6: three fp16 streams (every shipped config) through the LDS-DMA kernel: persistent blocks, next tile in flight (option POST_DMA = 0: k_post)
        if (yhalf && y[1] && y[2] && div == 3.f && (Cprev == 32 || Cprev == 16) && h->opt.geti("POST_DMA", 1) != 0) {
            const int ntiles = (int)((L + POSTD_TT - 1) / POSTD_TT);
            const int slots = std::max(1, num_cus() * h->opt.geti("POST_DMA_OCC", 3) / B);
            const dim3 grid((unsigned)std::min(ntiles, slots), B);
            const int dbg = h->opt.geti("POST_DBG", 0);
            if (Cprev == 32)
                hipLaunchKernelGGL(k_post_dma<32>, grid, dim3(POSTD_TT), post_dma_smem<32>(), st, (const _Float16*)y[0], (const _Float16*)y[1],
                                   (const _Float16*)y[2], h->post_w.as<float>(), out, (int)L, lens, lm, dbg);
            else
                hipLaunchKernelGGL(k_post_dma<16>, grid, dim3(POSTD_TT), post_dma_smem<16>(), st, (const _Float16*)y[0], (const _Float16*)y[1],
                                   (const _Float16*)y[2], h->post_w.as<float>(), out, (int)L, lens, lm, dbg);
            return;
        }
        const size_t smem = (size_t)(7 * Cprev + (POST_TT + 6) * (Cprev + 4)) * 4;
        hipLaunchKernelGGL(k_post, dim3((unsigned)((L + POST_TT - 1) / POST_TT), B), dim3(256), smem, st, y[0], y[1], y[2],
                           h->post_w.as<float>(), out, (int)L, Cprev, div, yhalf ? 1 : 0, lens, lm, h->opt.geti("POST_DBG", 0));
    });
    HIP_CHECK(hipGetLastError());
}

}  // namespace rvcmi

// ---- C ABI -------------------------------------------------------------------------------------
extern "C" {

const char* rvcmi_last_error(void) { return g_last_error.c_str(); }
int rvcmi_version(void) { return RVCMI_VERSION; }

int rvcmi_nsf_create(const rvcmi_nsf_config* cfg, const rvcmi_tensor* weights, int n_weights, int device, int max_B,
                     int max_T, rvcmi_nsf** out) {
    return guarded([&] { nsf_create(cfg, weights, n_weights, device, max_B, max_T, out); });
}
int rvcmi_nsf_destroy(rvcmi_nsf* h) {
    return guarded([&] { delete h; });
}
int rvcmi_nsf_forward(rvcmi_nsf* h, int B, int T, const int* lengths, const float* x, const float* f0, const float* g,
                      const float* noise, int n_res, float* out, void* stream) {
    return guarded([&] { nsf_forward(h, B, T, lengths, x, f0, g, noise, n_res, out, (hipStream_t)stream, nullptr); });
}
int rvcmi_nsf_upp(const rvcmi_nsf* h) { return h ? h->upp : RVCMI_ERR_INVALID; }
size_t rvcmi_nsf_workspace_bytes(const rvcmi_nsf* h) { return h ? h->ws_bytes : 0; }

int rvcmi_nsf_debug_forward(rvcmi_nsf* h, int B, int T, const float* x, const float* f0, const float* g,
                            const float* noise, int n_res, const char* what, float* out_host, size_t capacity_floats,
                            int64_t shape_out[3], void* stream) {
    return guarded([&] {
        if (!what || !out_host || !shape_out) RVCMI_FAIL(RVCMI_ERR_INVALID, "null argument");
        TapRequest tr;
        tr.what = what;
        tr.out_host = out_host;
        tr.capacity = capacity_floats;
        tr.shape = shape_out;
        nsf_forward(h, B, T, nullptr, x, f0, g, noise, n_res, nullptr, (hipStream_t)stream, &tr);
        if (!tr.done) RVCMI_FAIL(RVCMI_ERR_INVALID, "tap '%s' was not produced", what);
    });
}

int rvcmi_nsf_set_option(rvcmi_nsf* h, const char* key, double value) {
    return guarded([&] {
        if (!h || !key) RVCMI_FAIL(RVCMI_ERR_INVALID, "null argument");
        if (!h->opt.set(key, value)) RVCMI_FAIL(RVCMI_ERR_INVALID, "unknown option '%s' for this handle", key);
    });
}
int rvcmi_nsf_profile_enable(rvcmi_nsf* h, int enable) {
    return guarded([&] {
        if (!h) RVCMI_FAIL(RVCMI_ERR_INVALID, "null handle");
        h->prof.enabled = enable != 0;
    });
}
int rvcmi_nsf_profile_read(rvcmi_nsf* h, rvcmi_kernel_stat* stats, int capacity, int* n, int reset) {
    return guarded([&] {
        if (!h) RVCMI_FAIL(RVCMI_ERR_INVALID, "null handle");
        h->prof.read(stats, capacity, n, reset);
    });
}

}  // extern "C"
