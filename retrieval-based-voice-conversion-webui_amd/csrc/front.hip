// Host side of the synthesizer front: weight preparation for enc_p / flow
// (stands in for rvc/synthesizer.py:10-28 for those sub-modules), workspace, and the launch sequence of
// SynthesizerTrnMsNSFsid.infer up to the decoder call (rvc/layers/synthesizers.py:171-192):
//     m_p, logs_p, x_mask = enc_p(phone, pitch, lengths, flow_head)
//     z_p = (m_p + exp(logs_p) * noise * 0.66666) * x_mask
//     z   = flow(z_p, x_mask, g, reverse=True);  out = z * x_mask   (channel-first, what dec.forward takes)
#include <algorithm>
#include <atomic>
#include <cmath>
#include <memory>
#include <string>

#include "common.hpp"
#include "conv_pack.hpp"
#include "front_kernels.hpp"

using namespace rvcmi;

namespace {

struct AttnLayer {
    ConvLayer qkv, o, f1, f2;
    ConvLayer f2s[4];  // conv_2 re-packed per quarter of the hidden channels (split FFN, k_fr_ffn_part)
    DevBuf relk, relv, g1, b1, g2, b2;
};
struct FlowLayer {
    ConvLayer pre, post;
    ConvLayer in[8], rs[8];
    int phys_base = 0;
};

}  // namespace

struct rvcmi_front {
    rvcmi_front_config cfg;
    int device = 0, max_B = 0, max_T = 0, Tp = 0;
    ConvLayer emb, proj;
    DevBuf emb_pitch;
    std::vector<AttnLayer> layers;
    std::vector<FlowLayer> flows;
    DevBuf cond_w, cond_b;  // all flows' cond_layer concatenated: [n_flows * 2H * n_layers][gin]
    // workspace
    DevBuf X, X2, QK, KF, VT, A, F, ZP, Ha, Hb, SK, GC;  // QK: q [B][T][H]; KF / VT: k / v tiles in fragment order
    DevBuf FP;           // split FFN: fp32 partial sums [4][rows <= FFN_SPLIT_ROWS][H]
    size_t ws_bytes = 0;
    Profiler prof;
    // dev / test options (common.hpp Options): FR_NJ (1 / 2: time-tile height), FR_NO_FFN_FUSION, FR_FFN_SPLIT (0 / 1: the FFN with its
    // hidden channels split over 4x the blocks; default by grid size), FR_STAMPS (prints; syncs).
    // Read from RVCMI_<KEY> once, in rvcmi_front_create; changed afterwards only through rvcmi_front_set_option.
    rvcmi::Options opt;
};

namespace {

constexpr int FFN_SPLIT = 4;            // slices of the hidden channels in the split FFN
constexpr size_t FFN_SPLIT_ROWS = 8192;  // B * T rows up to which the split form may be chosen (sizes its scratch)

struct TapReq {
    std::string what;
    float* out_host;
    size_t capacity;
    int64_t* shape;
    bool done = false;
};

const float* wdata(const WeightMap& wm, const std::string& name, std::initializer_list<int64_t> shape) {
    return (const float*)wm.get(name, shape).data;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute: set it once per (kernel instantiation, device).
static void ensure_dyn_lds(const void* kern, std::atomic<unsigned long long>& done) {
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    done.fetch_or(bit, std::memory_order_release);
}

// Time-tile height: 64 rows (NJ = 2) when that still gives every CU a block, else 32 rows -- at B = 1 a 10 s clip is
// only 38 tiles of 32 frames, and one tile's MFMA work on one CU is the latency floor of a launch.
static int pick_nj(const rvcmi_front* h, int B, int T) {
    if (h->opt.has("FR_NJ")) {  // tests force the large-batch tile height on small inputs
        const int v = h->opt.geti("FR_NJ", 0);
        if (v == 1 || v == 2) return v;
    }
    return (long)B * ((T + 63) / 64) >= 192 ? 2 : 1;
}

template <typename OpT, int CIN, int MI, int NW, int EPI, int NJ>
void launch_conv_nj(rvcmi_front* h, const char* name, FrConvArgs a, const ConvLayer& L, int B, hipStream_t st) {
    a.w = L.w_pack.p;
    a.ct_stride = L.ct_stride;
    a.ntaps = L.ntaps[0];
    a.bias = L.bias.as<float>();
    a.cout = L.cout;
    constexpr int TT = NJ * 32;
    const int rows = TT + a.ntaps - 1 + 2;
    const size_t smem = std::max<size_t>((size_t)rows * Tile<CIN>::STRIDE, 2 * NW * NJ * 32 * sizeof(float));
    const int ctiles = (L.cout + 31) / 32;
    const int gy = (ctiles + NW * MI - 1) / (NW * MI);
    auto kern = k_fr_conv<OpT, CIN, MI, NJ, NW, EPI>;
    static std::atomic<unsigned long long> attr_done{0};  // one instantiation = one static; one bit per device
    ensure_dyn_lds(reinterpret_cast<const void*>(kern), attr_done);
    const double flops = L.flops_per_pos * (double)a.T * B;
    h->prof.launch(name, flops, 0.0, st, [&] {
        hipLaunchKernelGGL(kern, dim3((a.T + TT - 1) / TT, gy, B), dim3(64 * NW), smem, st, a);
    });
    HIP_CHECK(hipGetLastError());
}
template <typename OpT, int CIN, int MI, int NW, int EPI>
void launch_conv(rvcmi_front* h, const char* name, const FrConvArgs& a, const ConvLayer& L, int B, hipStream_t st) {
    if (pick_nj(h, B, a.T) == 1) launch_conv_nj<OpT, CIN, MI, NW, EPI, 1>(h, name, a, L, B, st);
    else launch_conv_nj<OpT, CIN, MI, NW, EPI, 2>(h, name, a, L, B, st);
}

template <typename OpT, bool LAST, int NJ>
void launch_wn_nj(rvcmi_front* h, FrWnArgs a, const ConvLayer& Lin, const ConvLayer& Lrs, int B, hipStream_t st) {
    constexpr int H = 192;
    a.w_in = Lin.w_pack.p;
    a.ct_in = Lin.ct_stride;
    a.ntaps = Lin.ntaps[0];
    a.pad = (a.ntaps - 1) / 2;
    a.b_in = Lin.bias.as<float>();
    a.w_rs = Lrs.w_pack.p;
    a.ct_rs = Lrs.ct_stride;
    a.b_rs = Lrs.bias.as<float>();
    constexpr int TT = NJ * 32;
    const size_t smem = (size_t)(TT + a.ntaps - 1 + 2 + TT + 2) * Tile<H>::STRIDE + 2 * H * sizeof(float);
    auto kern = k_fr_wn<OpT, H, NJ, LAST>;
    static std::atomic<unsigned long long> attr_done{0};  // one instantiation = one static; one bit per device
    ensure_dyn_lds(reinterpret_cast<const void*>(kern), attr_done);
    const double flops = (Lin.flops_per_pos + Lrs.flops_per_pos) * (double)a.T * B;
    static unsigned long long* stamps = nullptr;  // dev only
    const bool want_stamps = h->opt.on("FR_STAMPS");
    if (want_stamps && !stamps) HIP_CHECK(hipMalloc((void**)&stamps, 64 * 8));
    a.stamps = want_stamps ? stamps : nullptr;
    h->prof.launch(LAST ? "flow_wn_last" : "flow_wn", flops, 0.0, st, [&] {
        hipLaunchKernelGGL(kern, dim3((a.T + TT - 1) / TT, B), dim3(64 * (H / 32)), smem, st, a);
    });
    HIP_CHECK(hipGetLastError());
    if (a.stamps) {
        HIP_CHECK(hipStreamSynchronize(st));
        unsigned long long t[14];
        HIP_CHECK(hipMemcpy(t, stamps, sizeof(t), hipMemcpyDeviceToHost));
        fprintf(stderr, "[fr_wn%s] wall(10ns)/cycles:", LAST ? "_last" : "");
        for (int i = 1; i < 7; ++i) fprintf(stderr, "  p%d %llu/%llu", i, t[2 * i] - t[2 * i - 2], t[2 * i + 1] - t[2 * i - 1]);
        fprintf(stderr, "\n");
    }
}
// Split form of a WN layer for small grids (round 6; option FR_WN_SPLIT, default by grid size like the split FFN): the in_layer + gate as
// one launch whose blocks take a THIRD of the (tanh, sigmoid) channel pairs each (grid.y = 3: 114 blocks of two waves for a single clip
// instead of 38 blocks of six, each streaming 246 KB of weights instead of 737 KB), the gate activations through an OpT buffer (h->A,
// free during the flow), then the 1x1 res_skip with the residual / skip update as a second launch split the same way.  Same K loops, same
// epilogue arithmetic in the same order as k_fr_wn: bit-identical results.
template <typename OpT, bool LAST>
void launch_wn_split(rvcmi_front* h, const FrWnArgs& w, const ConvLayer& Lin, const ConvLayer& Lrs, int B, hipStream_t st) {
    constexpr int H = 192;
    {
        FrConvArgs a = {};
        a.in = w.x; a.in_op = 0; a.in_bstride = w.bstride; a.T = w.T; a.t_off = w.t_off; a.len = w.len;
        a.pad = (Lin.ntaps[0] - 1) / 2; a.H = H;
        a.out_op = h->A.p; a.out_op_bstride = (long)w.T * H;
        a.gc = w.gc; a.gc_bstride = w.gc_bstride;
        const int mode = h->opt.geti("FR_WN_SPLIT", 2);  // 1 = channel split only (bit-identical to k_fr_wn), 2 = taps over the waves (default)
        if (mode == 2 && Lin.ntaps[0] >= 2 && Lin.ntaps[0] <= 8) {
            a.w = Lin.w_pack.p; a.ct_stride = Lin.ct_stride; a.ntaps = Lin.ntaps[0]; a.bias = Lin.bias.as<float>(); a.cout = Lin.cout;
            const int rows = 32 + a.ntaps - 1 + 2;
            const size_t smem = (size_t)rows * Tile<H>::STRIDE + (size_t)(a.ntaps - 1) * 2 * 16 * 64 * sizeof(float);
            auto kern = k_fr_gate_ks<OpT, H, 1>;
            static std::atomic<unsigned long long> attr_done{0};
            ensure_dyn_lds(reinterpret_cast<const void*>(kern), attr_done);
            h->prof.launch("flow_wn_gate", Lin.flops_per_pos * (double)a.T * B, 0.0, st, [&] {
                hipLaunchKernelGGL(kern, dim3((a.T + 31) / 32, H / 32, B), dim3(64 * a.ntaps), smem, st, a);
            });
            HIP_CHECK(hipGetLastError());
        } else {
            launch_conv_nj<OpT, H, 2, 2, FR_GATE, 1>(h, "flow_wn_gate", a, Lin, B, st);
        }
    }
    {
        FrConvArgs a = {};
        a.in = h->A.p; a.in_op = 1; a.in_bstride = (long)w.T * H; a.T = w.T; a.t_off = w.t_off; a.len = w.len; a.H = H;
        a.res = w.x; a.out = w.x_out; a.out_bstride = w.bstride; a.out_C = H; a.skip = w.skip; a.first = w.first;
        if constexpr (LAST) {
            a.out = w.skip;  // (strides only; the last layer writes skip alone)
            launch_conv_nj<OpT, H, 1, 2, FR_WN_RS_LAST, 1>(h, "flow_wn_rs", a, Lrs, B, st);
        } else {
            launch_conv_nj<OpT, H, 2, 2, FR_WN_RS, 1>(h, "flow_wn_rs", a, Lrs, B, st);
        }
    }
}
template <typename OpT, bool LAST>
void launch_wn(rvcmi_front* h, const FrWnArgs& a, const ConvLayer& Lin, const ConvLayer& Lrs, int B, hipStream_t st) {
    const int nj = pick_nj(h, B, a.T);
    const long tiles = (long)((a.T + 32 * nj - 1) / (32 * nj)) * B;
    if (nj == 1 && h->opt.geti("FR_WN_SPLIT", tiles <= 96 ? 1 : 0) != 0) return launch_wn_split<OpT, LAST>(h, a, Lin, Lrs, B, st);
    if (nj == 1) launch_wn_nj<OpT, LAST, 1>(h, a, Lin, Lrs, B, st);
    else launch_wn_nj<OpT, LAST, 2>(h, a, Lin, Lrs, B, st);
}

template <typename OpT, int NJ1>
void launch_ffn_nj(rvcmi_front* h, FrFfnArgs a, const ConvLayer& L1, const ConvLayer& L2, int B, hipStream_t st) {
    constexpr int H = 192, F = 768;
    a.w1 = L1.w_pack.p;
    a.ct1 = L1.ct_stride;
    a.b1 = L1.bias.as<float>();
    a.w2 = L2.w_pack.p;
    a.ct2 = L2.ct_stride;
    a.b2 = L2.bias.as<float>();
    a.ntaps = L1.ntaps[0];
    constexpr int HR = 32 * NJ1;
    const int TV = HR - (a.ntaps - 1);
    const size_t smem = (size_t)(HR + a.ntaps - 1 + 2) * Tile<H>::STRIDE + (size_t)(HR + a.ntaps - 1 + 2) * Tile<F>::STRIDE;
    auto kern = k_fr_ffn<OpT, H, F, NJ1>;
    static std::atomic<unsigned long long> attr_done{0};  // one instantiation = one static; one bit per device
    ensure_dyn_lds(reinterpret_cast<const void*>(kern), attr_done);
    const double flops = (L1.flops_per_pos + L2.flops_per_pos) * (double)a.T * B;
    h->prof.launch("enc_ffn_ln", flops, 0.0, st, [&] {
        hipLaunchKernelGGL(kern, dim3((a.T + TV - 1) / TV, B), dim3(64 * (H / 32)), smem, st, a);
    });
    HIP_CHECK(hipGetLastError());
}
// split form for small grids (see k_fr_ffn_part): S x the blocks, each with a quarter of the weight stream, + a LayerNorm pass
template <typename OpT, int NJ1>
void launch_ffn_split_nj(rvcmi_front* h, FrFfnArgs a, const ConvLayer& L1, const ConvLayer (&L2s)[4], const ConvLayer& L2, int B, hipStream_t st) {
    constexpr int H = 192, F = 768, FS = F / FFN_SPLIT;
    a.w1 = L1.w_pack.p;
    a.ct1 = L1.ct_stride;
    a.b1 = L1.bias.as<float>();
    for (int s = 0; s < FFN_SPLIT; ++s) a.w2s[s] = L2s[s].w_pack.p;
    a.ct2s = L2s[0].ct_stride;
    a.b2 = L2.bias.as<float>();
    a.ntaps = L1.ntaps[0];
    a.part = h->FP.as<float>();
    a.nsplit = FFN_SPLIT;
    constexpr int HR = 32 * NJ1;
    const int TV = HR - (a.ntaps - 1);
    const size_t smem = (size_t)(HR + a.ntaps - 1 + 2) * Tile<H>::STRIDE + (size_t)(HR + a.ntaps - 1 + 2) * Tile<FS>::STRIDE;
    auto kern = k_fr_ffn_part<OpT, H, FS, NJ1>;
    static std::atomic<unsigned long long> attr_done{0};
    ensure_dyn_lds(reinterpret_cast<const void*>(kern), attr_done);
    const double flops = (L1.flops_per_pos + L2.flops_per_pos) * (double)a.T * B;
    h->prof.launch("enc_ffn_part", flops, 0.0, st, [&] {
        hipLaunchKernelGGL(kern, dim3((a.T + TV - 1) / TV, B, FFN_SPLIT), dim3(64 * (H / 32)), smem, st, a);
    });
    h->prof.launch("enc_ffn_ln", 0.0, 0.0, st, [&] {
        hipLaunchKernelGGL(k_fr_ffn_ln<H>, dim3((unsigned)(((size_t)B * a.T + 3) / 4)), dim3(256), 0, st, (const float*)a.part, FFN_SPLIT, a.x, a.xo, a.b2,
                           a.gamma, a.beta, a.len, a.T, B);
    });
    HIP_CHECK(hipGetLastError());
}
template <typename OpT>
void launch_ffn(rvcmi_front* h, const FrFfnArgs& a, const ConvLayer& L1, const ConvLayer& L2, int B, hipStream_t st) {
    if (pick_nj(h, B, a.T) == 1) launch_ffn_nj<OpT, 1>(h, a, L1, L2, B, st);
    else launch_ffn_nj<OpT, 2>(h, a, L1, L2, B, st);
}

void tap_copy(TapReq* tr, const char* what, const float* dev, int B, int T, int C, hipStream_t st) {
    if (!tr || tr->done || tr->what != what) return;
    const size_t n = (size_t)B * T * C;
    if (n > tr->capacity) RVCMI_FAIL(RVCMI_ERR_INVALID, "tap buffer too small: need %zu floats", n);
    HIP_CHECK(hipStreamSynchronize(st));
    HIP_CHECK(hipMemcpy(tr->out_host, dev, n * sizeof(float), hipMemcpyDeviceToHost));
    tr->shape[0] = B;
    tr->shape[1] = T;
    tr->shape[2] = C;
    tr->done = true;
}

template <typename OpT, int CIN>
void front_forward_t(rvcmi_front* h, int B, int T, const float* phone, const long long* pitch, const long long* lengths,
                     const float* g, const float* noise, int fh, float* z_out, hipStream_t st, TapReq* tr) {
    const rvcmi_front_config& c = h->cfg;
    constexpr int H = 192;
    const int T2 = T - fh;
    const int Tp = h->Tp;
    float* X = h->X.as<float>();
    float* X2 = h->X2.as<float>();  // the fused FFN writes the other buffer (its tiles read a halo of X)
    // ---- TextEncoder (encoders.py:134-159) ----
    {
        FrConvArgs a = {};
        a.in = phone; a.in_op = 0; a.in_bstride = (long)T * CIN; a.T = T; a.t_off = 0; a.len = lengths;
        a.pad = 0; a.out = X; a.out_bstride = (long)T * H; a.out_C = H;
        a.pitch = c.use_f0 ? pitch : nullptr; a.emb_pitch = h->emb_pitch.as<float>(); a.scale = sqrtf((float)H);
        launch_conv<OpT, CIN, 1, 6, FR_EMB>(h, "enc_emb", a, h->emb, B, st);
        tap_copy(tr, "emb", X, B, T, H, st);
        if (tr && tr->done) return;
    }
    for (int i = 0; i < c.n_layers; ++i) {
        AttnLayer& L = h->layers[i];
        {
            FrConvArgs a = {};
            a.in = X; a.in_bstride = (long)T * H; a.T = T; a.len = lengths;
            a.out_op = h->QK.p; a.out_op_bstride = (long)T * H;
            a.vt = h->VT.p; a.kf = h->KF.p; a.Tp = Tp; a.qdiv = sqrtf((float)(H / c.n_heads)); a.H = H;
            launch_conv<OpT, H, 1, 6, FR_QKV>(h, "enc_qkv", a, L.qkv, B, st);
        }
        {
            FrAttnArgs a = {};
            a.q = h->QK.p; a.kf = h->KF.p; a.vf = h->VT.p; a.out = h->A.p; a.relk = L.relk.p; a.relv = L.relv.as<float>(); a.len = lengths;
            a.T = T; a.Tp = Tp; a.H = H; a.ws = c.window_size;
            const double flops = 4.0 * (double)T * T * H * B;
            static unsigned long long* stamps = nullptr;  // dev only
            const bool want_stamps = h->opt.on("FR_STAMPS");
            if (want_stamps && !stamps) HIP_CHECK(hipMalloc((void**)&stamps, 256 * 8));
            a.stamps = want_stamps ? stamps : nullptr;
            h->prof.launch("enc_attn", flops, 0.0, st, [&] {
                if (c.window_size <= 10) hipLaunchKernelGGL((k_fr_attn<OpT, 96, 21>), dim3((T + 31) / 32, c.n_heads, B), dim3(256), 0, st, a);
                else hipLaunchKernelGGL((k_fr_attn<OpT, 96, 31>), dim3((T + 31) / 32, c.n_heads, B), dim3(256), 0, st, a);
            });
            HIP_CHECK(hipGetLastError());
            if (a.stamps) {
                HIP_CHECK(hipStreamSynchronize(st));
                unsigned long long t[12];
                HIP_CHECK(hipMemcpy(t, stamps, sizeof(t), hipMemcpyDeviceToHost));
                fprintf(stderr, "[fr_attn] wall(10ns)/cycles:");
                for (int k = 1; k < 5; ++k) fprintf(stderr, "  p%d %llu/%llu", k, t[2 * k] - t[2 * k - 2], t[2 * k + 1] - t[2 * k - 1]);
                fprintf(stderr, "\n");
            }
        }
        {
            FrConvArgs a = {};
            a.in = h->A.p; a.in_op = 1; a.in_bstride = (long)T * H; a.T = T; a.len = lengths;
            a.out = X; a.out_bstride = (long)T * H; a.out_C = H; a.res = X;
            a.gamma = L.g1.as<float>(); a.beta = L.b1.as<float>();
            launch_conv<OpT, H, 1, 6, FR_RES_LN>(h, "enc_o_ln", a, L.o, B, st);
            if (i == 0) tap_copy(tr, "attn0", X, B, T, H, st);
            if (tr && tr->done) return;
        }
        if (c.kernel_size <= 5 && !h->opt.on("FR_NO_FFN_FUSION")) {
            // FFN + residual + LayerNorm in one launch; reads X (with a halo) and writes the other stream buffer
            FrFfnArgs a = {};
            a.x = X; a.xo = X2; a.bstride = (long)T * H; a.T = T; a.len = lengths;
            a.gamma = L.g2.as<float>(); a.beta = L.b2.as<float>();
            // few time tiles (a single clip): split the hidden channels over 4x the blocks; option FR_FFN_SPLIT = 0 / 1 pins the choice
            const int nj = pick_nj(h, B, T);
            const long tiles = (long)((T + 32 * nj - 3) / (32 * nj - 2)) * B;
            const bool split = c.kernel_size == 3 && c.filter_channels == 768 && (size_t)B * T <= FFN_SPLIT_ROWS &&
                               h->opt.geti("FR_FFN_SPLIT", tiles <= 96 ? 1 : 0) != 0;
            if (split && nj == 1) launch_ffn_split_nj<OpT, 1>(h, a, L.f1, L.f2s, L.f2, B, st);
            else if (split) launch_ffn_split_nj<OpT, 2>(h, a, L.f1, L.f2s, L.f2, B, st);
            else launch_ffn<OpT>(h, a, L.f1, L.f2, B, st);
            std::swap(X, X2);
        } else {
            {
                FrConvArgs a = {};
                a.in = X; a.in_bstride = (long)T * H; a.T = T; a.len = lengths; a.premask = 1; a.pad = (c.kernel_size - 1) / 2;
                a.out_op = h->F.p; a.out_op_bstride = (long)T * c.filter_channels;
                launch_conv<OpT, H, 1, 6, FR_RELU_OP>(h, "enc_ffn1", a, L.f1, B, st);
            }
            {
                FrConvArgs a = {};
                a.in = h->F.p; a.in_op = 1; a.in_bstride = (long)T * c.filter_channels; a.T = T; a.len = lengths;
                a.pad = (c.kernel_size - 1) / 2; a.postmask = 1;
                a.out = X; a.out_bstride = (long)T * H; a.out_C = H; a.res = X;
                a.gamma = L.g2.as<float>(); a.beta = L.b2.as<float>();
                launch_conv<OpT, 768, 1, 6, FR_RES_LN>(h, "enc_ffn2_ln", a, L.f2, B, st);
            }
        }
        {
            char nm[32];
            snprintf(nm, sizeof(nm), "layer%d", i);
            tap_copy(tr, nm, X, B, T, H, st);
            if (tr && tr->done) return;
        }
    }
    float* ZP = h->ZP.as<float>();
    {   // proj + prior sample (encoders.py:152-158, synthesizers.py:182-183); rows [fh, T) only
        FrConvArgs a = {};
        a.in = X + (size_t)fh * H; a.in_bstride = (long)T * H; a.T = T2; a.t_off = fh; a.len = lengths; a.premask = 1;
        a.out = ZP; a.out_bstride = (long)T2 * H; a.out_C = H; a.noise = noise; a.H = H;
        launch_conv<OpT, H, 2, 6, FR_PROJ_ZP>(h, "enc_proj_zp", a, h->proj, B, st);
        tap_copy(tr, "z_p", ZP, B, T2, H, st);
        if (tr && tr->done) return;
    }
    // ---- flow, reverse (residuals.py:319-321) ----
    const int gcn = 2 * H * c.flow_n_layers;
    if (c.gin_channels) {
        const int tot = gcn * c.flow_n_flows;
        h->prof.launch("flow_cond", 2.0 * tot * c.gin_channels * B, 0.0, st, [&] {
            hipLaunchKernelGGL(k_cond, dim3((tot + 3) / 4, B), dim3(256), 0, st, g, h->cond_w.as<float>(), h->cond_b.as<float>(),
                               h->GC.as<float>(), c.gin_channels, tot);
        });
        HIP_CHECK(hipGetLastError());
    }
    float* Ha = h->Ha.as<float>();
    float* Hb = h->Hb.as<float>();
    float* SK = h->SK.as<float>();
    for (int f = c.flow_n_flows - 1; f >= 0; --f) {
        FlowLayer& FL = h->flows[f];
        {
            FrConvArgs a = {};
            a.in = ZP; a.in_bstride = (long)T2 * H; a.T = T2; a.t_off = fh; a.len = lengths;
            a.out = Ha; a.out_bstride = (long)T2 * H; a.out_C = H;
            launch_conv<OpT, H, 1, 6, FR_F32_MASK>(h, "flow_pre", a, FL.pre, B, st);
        }
        float* xin = Ha;
        float* xout = Hb;
        for (int l = 0; l < c.flow_n_layers; ++l) {
            FrWnArgs a = {};
            a.x = xin; a.x_out = xout; a.skip = SK; a.first = l == 0; a.bstride = (long)T2 * H; a.T = T2; a.t_off = fh; a.len = lengths;
            a.gc = c.gin_channels ? h->GC.as<float>() + (size_t)f * gcn + (size_t)l * 2 * H : nullptr;
            a.gc_bstride = (long)gcn * c.flow_n_flows;
            if (l == c.flow_n_layers - 1) launch_wn<OpT, true>(h, a, FL.in[l], FL.rs[l], B, st);
            else launch_wn<OpT, false>(h, a, FL.in[l], FL.rs[l], B, st);
            std::swap(xin, xout);
        }
        {
            FrConvArgs a = {};
            a.in = SK; a.in_bstride = (long)T2 * H; a.T = T2; a.t_off = fh; a.len = lengths; a.premask = 1;
            a.out = ZP; a.out_bstride = (long)T2 * H; a.out_C = H; a.phys_base = FL.phys_base;
            launch_conv<OpT, H, 1, 3, FR_COUPLE>(h, "flow_post", a, FL.post, B, st);
        }
        char nm[32];
        snprintf(nm, sizeof(nm), "flow%d", f);
        tap_copy(tr, nm, ZP, B, T2, H, st);
        if (tr && tr->done) return;
    }
    if (z_out) {
        h->prof.launch("front_out", 0.0, (double)B * T2 * H * 8, st, [&] {
            hipLaunchKernelGGL(k_fr_out, dim3((T2 + 31) / 32, (H + 31) / 32, B), dim3(256), 0, st, ZP, z_out, T2, H, fh, lengths);
        });
        HIP_CHECK(hipGetLastError());
    }
}

void front_forward(rvcmi_front* h, int B, int T, const float* phone, const int64_t* pitch, const int64_t* lengths,
                   const float* g, const float* noise, int fh, float* z_out, hipStream_t st, TapReq* tr) {
    if (!h) RVCMI_FAIL(RVCMI_ERR_INVALID, "null handle");
    if (B < 1 || T < 1 || B > h->max_B || T > h->max_T)
        RVCMI_FAIL(RVCMI_ERR_NOMEM, "front: shape B=%d T=%d exceeds the handle's max_B=%d max_T=%d", B, T, h->max_B, h->max_T);
    if (fh < 0 || fh >= T) RVCMI_FAIL(RVCMI_ERR_INVALID, "front: flow_head %d out of range for T=%d", fh, T);
    if (!phone || !noise || (h->cfg.use_f0 && !pitch) || (h->cfg.gin_channels && !g))
        RVCMI_FAIL(RVCMI_ERR_INVALID, "front: null input");
    HIP_CHECK(hipSetDevice(h->device));
    const long long* pl = (const long long*)pitch;
    const long long* ll = (const long long*)lengths;
    const bool bf = h->cfg.operand == RVCMI_OPERAND_BF16;
    if (h->cfg.in_channels == 768) {
        if (bf) front_forward_t<__bf16, 768>(h, B, T, phone, pl, ll, g, noise, fh, z_out, st, tr);
        else front_forward_t<_Float16, 768>(h, B, T, phone, pl, ll, g, noise, fh, z_out, st, tr);
    } else {
        if (bf) front_forward_t<__bf16, 256>(h, B, T, phone, pl, ll, g, noise, fh, z_out, st, tr);
        else front_forward_t<_Float16, 256>(h, B, T, phone, pl, ll, g, noise, fh, z_out, st, tr);
    }
}

void upload_vec(DevBuf& d, const float* p, size_t n) {
    d.alloc(n * sizeof(float));
    HIP_CHECK(hipMemcpy(d.p, p, n * sizeof(float), hipMemcpyHostToDevice));
}

rvcmi_front* front_create(const rvcmi_front_config* cfg, const rvcmi_tensor* weights, int n_weights, int device, int max_B, int max_T) {
    if (!cfg || !weights) RVCMI_FAIL(RVCMI_ERR_INVALID, "null argument");
    const rvcmi_front_config& c = *cfg;
    if (c.operand != RVCMI_OPERAND_BF16 && c.operand != RVCMI_OPERAND_F16)
        RVCMI_FAIL(RVCMI_ERR_INVALID, "front: operand must be fp16 or bf16 (MFMA path only)");
    if (c.hidden_channels != 192 || c.inter_channels != 192 || c.filter_channels != 768 || c.n_heads != 2)
        RVCMI_FAIL(RVCMI_ERR_INVALID, "front: unsupported geometry (hidden %d inter %d filter %d heads %d); every shipped RVC config is 192/192/768/2",
                   c.hidden_channels, c.inter_channels, c.filter_channels, c.n_heads);
    if (c.in_channels != 768 && c.in_channels != 256) RVCMI_FAIL(RVCMI_ERR_INVALID, "front: in_channels must be 768 (v2) or 256 (v1)");
    if (c.flow_dilation_rate != 1) RVCMI_FAIL(RVCMI_ERR_INVALID, "front: flow dilation_rate must be 1");
    if (c.n_layers < 1 || c.n_layers > 16 || c.flow_n_layers < 1 || c.flow_n_layers > 8 || c.flow_n_flows < 1 || c.flow_n_flows > 8 ||
        c.kernel_size < 1 || c.kernel_size % 2 == 0 || c.flow_kernel_size % 2 == 0 || c.window_size < 1 || c.window_size > 15)
        RVCMI_FAIL(RVCMI_ERR_INVALID, "front: unsupported layer counts / kernel sizes");
    if (max_B < 1 || max_T < 1) RVCMI_FAIL(RVCMI_ERR_INVALID, "front: max_B/max_T must be positive");
    HIP_CHECK(hipSetDevice(device));
    WeightMap wm;
    for (int i = 0; i < n_weights; ++i) wm.m[weights[i].name] = &weights[i];
    std::unique_ptr<rvcmi_front> h(new rvcmi_front);
    h->opt.load_env({"FR_NJ", "FR_NO_FFN_FUSION", "FR_FFN_SPLIT", "FR_WN_SPLIT", "FR_STAMPS"});
    h->cfg = c;
    h->device = device;
    h->max_B = max_B;
    h->max_T = max_T;
    h->Tp = (max_T + 31) / 32 * 32 + 32;
    const int H = c.hidden_channels, IN = c.in_channels, FC = c.filter_channels, dk = H / c.n_heads, nb = 2 * c.window_size + 1;
    const int one = 1, zero = 0;
    const int op = c.operand;

    {   // emb_phone: Linear(in, H) = 1x1 conv   encoders.py:142
        const float* W = wdata(wm, "enc_p.emb_phone.weight", {H, IN});
        const float* b = wdata(wm, "enc_p.emb_phone.bias", {H});
        build_conv(h->emb, IN, H, 1, &one, &zero, 1, [&](int co, int ci, int, int) { return W[(size_t)co * IN + ci]; }, b, op);
        if (c.use_f0) upload_vec(h->emb_pitch, wdata(wm, "enc_p.emb_pitch.weight", {256, H}), (size_t)256 * H);
    }
    h->layers.resize(c.n_layers);
    for (int i = 0; i < c.n_layers; ++i) {
        AttnLayer& L = h->layers[i];
        const std::string a = "enc_p.encoder.attn_layers." + std::to_string(i) + ".";
        const float* Wq = wdata(wm, a + "conv_q.weight", {H, H, 1});
        const float* Wk = wdata(wm, a + "conv_k.weight", {H, H, 1});
        const float* Wv = wdata(wm, a + "conv_v.weight", {H, H, 1});
        std::vector<float> bq(3 * H);
        memcpy(bq.data(), wdata(wm, a + "conv_q.bias", {H}), H * 4);
        memcpy(bq.data() + H, wdata(wm, a + "conv_k.bias", {H}), H * 4);
        memcpy(bq.data() + 2 * H, wdata(wm, a + "conv_v.bias", {H}), H * 4);
        build_conv(L.qkv, H, 3 * H, 1, &one, &zero, 1, [&](int co, int ci, int, int) {
            const float* W = co < H ? Wq : (co < 2 * H ? Wk : Wv);
            return W[(size_t)(co % H) * H + ci];
        }, bq.data(), op);
        const float* Wo = wdata(wm, a + "conv_o.weight", {H, H, 1});
        build_conv(L.o, H, H, 1, &one, &zero, 1, [&](int co, int ci, int, int) { return Wo[(size_t)co * H + ci]; },
                   wdata(wm, a + "conv_o.bias", {H}), op);
        // relative embeddings (attentions.py:45-54): keys as A fragments of one 32-row tile, values fp32
        const float* Ek = wdata(wm, a + "emb_rel_k", {1, nb, dk});
        std::vector<uint16_t> pk((size_t)(dk / 16) * 512, 0);
        for (int s = 0; s < dk / 16; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int r = lane & 31;
                if (r >= nb) continue;
                for (int e = 0; e < 8; ++e) {
                    const float v = Ek[(size_t)r * dk + 16 * s + 8 * (lane >> 5) + e];
                    pk[(size_t)s * 512 + lane * 8 + e] = op == RVCMI_OPERAND_BF16 ? f32_to_bf16(v) : f32_to_f16(v);
                }
            }
        L.relk.alloc(pk.size() * 2);
        HIP_CHECK(hipMemcpy(L.relk.p, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
        upload_vec(L.relv, wdata(wm, a + "emb_rel_v", {1, nb, dk}), (size_t)nb * dk);
        const std::string n1 = "enc_p.encoder.norm_layers_1." + std::to_string(i) + ".";
        const std::string n2 = "enc_p.encoder.norm_layers_2." + std::to_string(i) + ".";
        upload_vec(L.g1, wdata(wm, n1 + "gamma", {H}), H);
        upload_vec(L.b1, wdata(wm, n1 + "beta", {H}), H);
        upload_vec(L.g2, wdata(wm, n2 + "gamma", {H}), H);
        upload_vec(L.b2, wdata(wm, n2 + "beta", {H}), H);
        const std::string f = "enc_p.encoder.ffn_layers." + std::to_string(i) + ".";
        const int ks = c.kernel_size;
        const float* W1 = wdata(wm, f + "conv_1.weight", {FC, H, ks});
        const float* W2 = wdata(wm, f + "conv_2.weight", {H, FC, ks});
        build_conv(L.f1, H, FC, 1, &ks, &zero, 1, [&](int co, int ci, int, int tap) { return W1[((size_t)co * H + ci) * ks + tap]; },
                   wdata(wm, f + "conv_1.bias", {FC}), op);
        build_conv(L.f2, FC, H, 1, &ks, &zero, 1, [&](int co, int ci, int, int tap) { return W2[((size_t)co * FC + ci) * ks + tap]; },
                   wdata(wm, f + "conv_2.bias", {H}), op);
        if (FC % FFN_SPLIT == 0) {
            const int FSl = FC / FFN_SPLIT;
            for (int sp = 0; sp < FFN_SPLIT; ++sp)  // conv_2 restricted to the input channels [sp * FSl, (sp + 1) * FSl): a conv of its own
                build_conv(L.f2s[sp], FSl, H, 1, &ks, &zero, 1,
                           [&](int co, int ci, int, int tap) { return W2[((size_t)co * FC + sp * FSl + ci) * ks + tap]; }, nullptr, op);
        }
    }
    auto paired = [&](int cop, int Hh) {  // packed row -> original row for [a-tile, b-tile] pairs (b rows start at Hh)
        const int tile = cop / 32, pair = tile / 2, which = tile % 2;
        return which * Hh + pair * 32 + cop % 32;
    };
    {   // proj: packed as (m tile, logs tile) pairs so that one wave holds both halves of a channel
        const int IC = c.inter_channels;
        const float* W = wdata(wm, "enc_p.proj.weight", {2 * IC, H, 1});
        build_conv(h->proj, H, 2 * IC, 1, &one, &zero, 1, [&](int co, int ci, int, int) { return W[(size_t)paired(co, IC) * H + ci]; },
                   wdata(wm, "enc_p.proj.bias", {2 * IC}), op);
    }
    // flow: coupling f runs after (n_flows - f) flips of the channel axis; the flips are folded into pre / post.
    h->flows.resize(c.flow_n_flows);
    const int IC = c.inter_channels, half = IC / 2, fk = c.flow_kernel_size, gcn = 2 * H * c.flow_n_layers;
    std::vector<float> cw, cb;
    for (int f = 0; f < c.flow_n_flows; ++f) {
        FlowLayer& FL = h->flows[f];
        const std::string p = "flow.flows." + std::to_string(2 * f) + ".";
        const bool odd = ((c.flow_n_flows - f) % 2) == 1;
        const float* Wp = wdata(wm, p + "pre.weight", {H, half, 1});
        build_conv(FL.pre, IC, H, 1, &one, &zero, 1, [&](int co, int ph, int, int) {
            // logical x0[c] = physical channel c (even) / IC-1-c (odd)
            const int cl = odd ? IC - 1 - ph : ph;
            return cl < half ? Wp[(size_t)co * half + cl] : 0.f;
        }, wdata(wm, p + "pre.bias", {H}), op);
        for (int l = 0; l < c.flow_n_layers; ++l) {
            const bool last = l == c.flow_n_layers - 1;
            const float* Wi = wdata(wm, p + "enc.in_layers." + std::to_string(l) + ".weight", {2 * H, H, fk});
            build_conv(FL.in[l], H, 2 * H, 1, &fk, &zero, 1,
                       [&](int co, int ci, int, int tap) { return Wi[((size_t)paired(co, H) * H + ci) * fk + tap]; },
                       wdata(wm, p + "enc.in_layers." + std::to_string(l) + ".bias", {2 * H}), op);
            const int rsn = last ? H : 2 * H;
            const float* Wr = wdata(wm, p + "enc.res_skip_layers." + std::to_string(l) + ".weight", {rsn, H, 1});
            build_conv(FL.rs[l], H, rsn, 1, &one, &zero, 1,
                       [&](int co, int ci, int, int) { return Wr[(size_t)(last ? co : paired(co, H)) * H + ci]; },
                       wdata(wm, p + "enc.res_skip_layers." + std::to_string(l) + ".bias", {rsn}), op);
        }
        // post (mean only): output row o addresses physical channel phys_base + o
        const float* Wo = wdata(wm, p + "post.weight", {half, H, 1});
        const float* bo = wdata(wm, p + "post.bias", {half});
        std::vector<float> bperm(half);
        for (int o = 0; o < half; ++o) bperm[o] = bo[odd ? half - 1 - o : o];
        build_conv(FL.post, H, half, 1, &one, &zero, 1, [&](int o, int ci, int, int) { return Wo[(size_t)(odd ? half - 1 - o : o) * H + ci]; },
                   bperm.data(), op);
        FL.phys_base = odd ? 0 : half;
        if (c.gin_channels) {
            const float* Wc = wdata(wm, p + "enc.cond_layer.weight", {gcn, c.gin_channels, 1});
            const float* bc = wdata(wm, p + "enc.cond_layer.bias", {gcn});
            cw.insert(cw.end(), Wc, Wc + (size_t)gcn * c.gin_channels);
            cb.insert(cb.end(), bc, bc + gcn);
        }
    }
    if (c.gin_channels) {
        upload(h->cond_w, cw);
        upload(h->cond_b, cb);
    }
    // workspace
    const size_t BT = (size_t)max_B * max_T;
    size_t ws = 0;
    auto A = [&](DevBuf& d, size_t bytes) {
        d.alloc(bytes);
        ws += bytes;
    };
    A(h->X, BT * H * 4);
    A(h->X2, BT * H * 4);
    A(h->QK, BT * H * 2);
    A(h->KF, (size_t)max_B * H * h->Tp * 2);
    A(h->VT, (size_t)max_B * H * h->Tp * 2);
    A(h->A, BT * H * 2);
    A(h->F, BT * FC * 2);
    A(h->ZP, BT * IC * 4);
    A(h->Ha, BT * H * 4);
    A(h->Hb, BT * H * 4);
    A(h->SK, BT * H * 4);
    A(h->GC, (size_t)max_B * gcn * c.flow_n_flows * 4 + 16);
    A(h->FP, (size_t)FFN_SPLIT * std::min(BT, FFN_SPLIT_ROWS) * H * 4);
    HIP_CHECK(hipMemset(h->VT.p, 0, h->VT.bytes));  // key padding columns must stay finite
    HIP_CHECK(hipMemset(h->KF.p, 0, h->KF.bytes));
    h->ws_bytes = ws;
    return h.release();
}

}  // namespace

extern "C" {

int rvcmi_front_create(const rvcmi_front_config* cfg, const rvcmi_tensor* weights, int n_weights, int device, int max_B,
                       int max_T, rvcmi_front** out) {
    return guarded([&] {
        if (!out) RVCMI_FAIL(RVCMI_ERR_INVALID, "null out pointer");
        *out = front_create(cfg, weights, n_weights, device, max_B, max_T);
    });
}
int rvcmi_front_destroy(rvcmi_front* h) {
    return guarded([&] { delete h; });
}
int rvcmi_front_forward(rvcmi_front* h, int B, int T, const float* phone, const int64_t* pitch, const int64_t* lengths,
                        const float* g, const float* noise, int flow_head, float* z_out, void* stream) {
    return guarded([&] {
        if (!z_out) RVCMI_FAIL(RVCMI_ERR_INVALID, "null output");
        front_forward(h, B, T, phone, pitch, lengths, g, noise, flow_head, z_out, (hipStream_t)stream, nullptr);
    });
}
size_t rvcmi_front_workspace_bytes(const rvcmi_front* h) { return h ? h->ws_bytes : 0; }
int rvcmi_front_debug_forward(rvcmi_front* h, int B, int T, const float* phone, const int64_t* pitch, const int64_t* lengths,
                              const float* g, const float* noise, int flow_head, const char* what, float* out_host,
                              size_t capacity_floats, int64_t shape_out[3], void* stream) {
    return guarded([&] {
        if (!what || !out_host || !shape_out) RVCMI_FAIL(RVCMI_ERR_INVALID, "null argument");
        TapReq tr;
        tr.what = what;
        tr.out_host = out_host;
        tr.capacity = capacity_floats;
        tr.shape = shape_out;
        front_forward(h, B, T, phone, pitch, lengths, g, noise, flow_head, nullptr, (hipStream_t)stream, &tr);
        if (!tr.done) RVCMI_FAIL(RVCMI_ERR_INVALID, "tap '%s' was not produced", what);
    });
}
int rvcmi_front_set_option(rvcmi_front* h, const char* key, double value) {
    return guarded([&] {
        if (!h || !key) RVCMI_FAIL(RVCMI_ERR_INVALID, "null argument");
        if (!h->opt.set(key, value)) RVCMI_FAIL(RVCMI_ERR_INVALID, "unknown option '%s' for this handle", key);
    });
}
int rvcmi_front_profile_enable(rvcmi_front* h, int enable) {
    return guarded([&] {
        if (!h) RVCMI_FAIL(RVCMI_ERR_INVALID, "null handle");
        h->prof.enabled = enable != 0;
    });
}
int rvcmi_front_profile_read(rvcmi_front* h, rvcmi_kernel_stat* stats, int capacity, int* n, int reset) {
    return guarded([&] {
        if (!h) RVCMI_FAIL(RVCMI_ERR_INVALID, "null handle");
        h->prof.read(stats, capacity, n, reset);
    });
}

}  // extern "C"
