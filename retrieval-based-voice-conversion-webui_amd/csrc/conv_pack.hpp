// Host-side weight preparation shared by the generator (nsf.hip) and the encoder/flow front (front.hip):
// fp32 -> operand conversion, MFMA fragment-order packing of convolution weights, named-tensor lookup.
#pragma once
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.hpp"
#include "nsf_kernels.hpp"

namespace rvcmi {

// ---- host float -> operand conversions (round-to-nearest-even, like the device casts) -----------
static inline uint16_t f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline uint16_t f32_to_f16(float f) {
    _Float16 h = (_Float16)f;
    uint16_t r;
    memcpy(&r, &h, 2);
    return r;
}

// One convolution (or one polyphase transposed convolution = `nphase` small convolutions).
struct ConvLayer {
    int cin = 0, cout = 0;
    int nphase = 1;
    int dstep = 1;
    int ntaps[16] = {0};   // real taps per phase
    int in_off[16] = {0};  // input row of (q, tap 0) relative to q, per phase
    int ntaps_p = 0;       // padded tap count used by the MFMA kernel (uniform over phases)
    long f32_off[16] = {0};
    long pack_off[16] = {0};
    long ct_stride = 0;    // packed elements per 32-channel output tile
    DevBuf w_f32, w_pack, bias;
    double flops_per_pos = 0;  // 2*cin*cout*sum(real taps) per output position q (all phases)
};

using WFn = std::function<float(int co, int ci, int phase, int tap)>;

static void build_conv(ConvLayer& L, int cin, int cout, int nphase, const int* ntaps, const int* in_off, int dstep,
                       const WFn& W, const float* bias, int operand) {
    L.cin = cin;
    L.cout = cout;
    L.nphase = nphase;
    L.dstep = dstep;
    int jmax = 0, jsum = 0;
    for (int p = 0; p < nphase; ++p) {
        L.ntaps[p] = ntaps[p];
        L.in_off[p] = in_off[p];
        jmax = std::max(jmax, ntaps[p]);
        jsum += ntaps[p];
    }
    L.flops_per_pos = 2.0 * cin * cout * jsum;
    if (bias) {
        L.bias.alloc(sizeof(float) * cout);
        HIP_CHECK(hipMemcpy(L.bias.p, bias, sizeof(float) * cout, hipMemcpyHostToDevice));
    }
    if (operand == RVCMI_OPERAND_F32) {
        std::vector<float> w((size_t)jsum * cin * cout);
        long off = 0;
        for (int p = 0; p < nphase; ++p) {
            L.f32_off[p] = off;
            for (int j = 0; j < ntaps[p]; ++j)
                for (int ci = 0; ci < cin; ++ci)
                    for (int co = 0; co < cout; ++co) w[off + ((size_t)j * cin + ci) * cout + co] = W(co, ci, p, j);
            off += (long)ntaps[p] * cin * cout;
        }
        L.w_f32.alloc(w.size() * sizeof(float));
        HIP_CHECK(hipMemcpy(L.w_f32.p, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
        L.ntaps_p = jmax;
        return;
    }
    if (cin % 16) RVCMI_FAIL(RVCMI_ERR_INVALID, "MFMA path needs C_in %% 16 == 0 (got %d)", cin);
    const int CC = cin / 16;
    const int tpg = CC >= KGROUP ? 1 : KGROUP / CC;
    if (CC >= KGROUP ? (CC % KGROUP) : (KGROUP % CC)) RVCMI_FAIL(RVCMI_ERR_INVALID, "unsupported C_in %d", cin);
    L.ntaps_p = (jmax + tpg - 1) / tpg * tpg;
    const int KSP = L.ntaps_p * CC;
    const int ctiles = (cout + 31) / 32;
    L.ct_stride = (long)KSP * 512;
    std::vector<uint16_t> pk((size_t)nphase * ctiles * L.ct_stride, 0);
    for (int p = 0; p < nphase; ++p) {
        L.pack_off[p] = (long)p * ctiles * L.ct_stride;
        for (int ct = 0; ct < ctiles; ++ct)
            for (int tap = 0; tap < ntaps[p]; ++tap)
                for (int cc = 0; cc < CC; ++cc) {
                    uint16_t* dst = pk.data() + L.pack_off[p] + (size_t)ct * L.ct_stride + (size_t)(tap * CC + cc) * 512;
                    for (int lane = 0; lane < 64; ++lane) {
                        const int co = ct * 32 + (lane & 31);
                        if (co >= cout) continue;
                        for (int e = 0; e < 8; ++e) {
                            const int ci = cc * 16 + 8 * (lane >> 5) + e;
                            const float v = W(co, ci, p, tap);
                            dst[lane * 8 + e] = operand == RVCMI_OPERAND_BF16 ? f32_to_bf16(v) : f32_to_f16(v);
                        }
                    }
                }
    }
    L.w_pack.alloc(pk.size() * 2);
    HIP_CHECK(hipMemcpy(L.w_pack.p, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
}

struct WeightMap {
    std::unordered_map<std::string, const rvcmi_tensor*> m;
    const rvcmi_tensor& get(const std::string& name, std::initializer_list<int64_t> shape) const {
        auto it = m.find(name);
        if (it == m.end()) RVCMI_FAIL(RVCMI_ERR_MISSING, "missing weight tensor '%s'", name.c_str());
        const rvcmi_tensor& t = *it->second;
        if ((size_t)t.ndim != shape.size()) RVCMI_FAIL(RVCMI_ERR_INVALID, "weight '%s': ndim %d", name.c_str(), t.ndim);
        int i = 0;
        for (int64_t s : shape) {
            if (t.shape[i] != s)
                RVCMI_FAIL(RVCMI_ERR_INVALID, "weight '%s': dim %d is %lld, expected %lld", name.c_str(), i,
                           (long long)t.shape[i], (long long)s);
            ++i;
        }
        return t;
    }
};

static void upload(DevBuf& d, const std::vector<float>& v) {
    d.alloc(v.size() * sizeof(float));
    HIP_CHECK(hipMemcpy(d.p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
}

}  // namespace rvcmi
