// The bidirectional GRU that ends the RMVPE f0 network (rvc/f0/e2e.py:31-35,50-67: nn.GRU(384, 256, num_layers=1, batch_first=True,
// bidirectional=True)), for gfx950.  NOT a row of SURVEY.md section 8 -- the north star leaves RMVPE on PyTorch-ROCm -- but bench.py --e2e
//  measured this one layer at 94-142 ms of a 125 ms conversion: MIOpen runs its ~2400 recurrent steps as ~2400 launches.
//
// Two kernels:
//   k_gru_xproj   the input projections of every step and both directions at once, GX = x . W_ih^T + b (one MFMA GEMM, operands straight
//                 from global memory in their natural row-major layouts: lane = row, 8 consecutive k = one 16-byte load);
//   k_gru_seq     the recurrence: ONE persistent block per (direction, sequence), the 768 x 256 recurrent matrix resident in the block's
//                 REGISTERS as fp16 pairs (768 threads x 128 VGPRs = 393 KB of the CU's 512 KB register file), h broadcast through LDS
//                 as fp16, fp32 accumulation (v_dot2c_f32_f16), fp32 state and gates.  A step is a 1.5 k-cycle dot-product phase, an
//                 8-lane transposing reduction by DPP, two LDS barriers: measured 1.17 us, i.e. 1.43 ms per 1216-frame clip (both directions
//                 run concurrently on two CUs; 64 clips: 2.2 ms) against 110 ms for torch / MIOpen (tools/gru_time.py).
//
// PyTorch's GRU cell (torch.nn.GRU docs; gate order r, z, n in the stacked weights):
//   r = sigmoid(W_ir x + b_ir + W_hr h + b_hr)      z = sigmoid(W_iz x + b_iz + W_hz h + b_hz)
//   n = tanh(W_in x + b_in + r * (W_hn h + b_hn))   h' = (1 - z) * n + z * h
#include <atomic>
#include <memory>

#include "common.hpp"

using namespace rvcmi;

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GH = 256;        // hidden units
constexpr int GR = 3 * GH;     // rows of the stacked recurrent matrix
constexpr int GNT = GR;        // threads of the recurrence block: (row group of 8 rows) x (k slice of 32 inputs)
constexpr int GPD = 2;         // steps the gate threads request their GX values ahead (registers: the recurrent matrix takes 128 of 168)

// GX[F / GR][row][F % GR] = sum_k x[row][k] * Wih[F][k] + bias[F],  F in [0, 2 * GR): block = 4 waves = 128 features x 32 rows
static __global__ void __launch_bounds__(256) k_gru_xproj(const _Float16* __restrict__ x, const _Float16* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ gx, int M, int K) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, hl = lane >> 5, col = lane & 31;
    const int row0 = blockIdx.x * 32, f0 = (blockIdx.y * 4 + wave) * 32;
    const _Float16* xa = x + (size_t)min(row0 + col, M - 1) * K + 8 * hl;   // (clamped rows: unconditional loads)
    const _Float16* wa = w + (size_t)(f0 + col) * K + 8 * hl;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll 8
    for (int k = 0; k < K; k += 16) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const half8*)(wa + k), *(const half8*)(xa + k), acc, 0, 0, 0);
    const int row = row0 + col;
    if (row >= M) return;
    const int dir = f0 / GR, c0 = f0 - dir * GR;
    float* o = gx + ((size_t)dir * M + row) * GR + c0 + 4 * hl;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 b = *(const f32x4*)(bias + f0 + 8 * g + 4 * hl);
        const f32x4 v = {acc[4 * g] + b[0], acc[4 * g + 1] + b[1], acc[4 * g + 2] + b[2], acc[4 * g + 3] + b[3]};
        *(f32x4*)(o + 8 * g) = v;
    }
}

// (v_rcp_f32, 1 ulp, instead of the IEEE division sequence: the gate chain is serial latency between the two barriers of a step)
__device__ __forceinline__ float sigmoid_f(float v) { return __builtin_amdgcn_rcpf(1.f + __expf(-v)); }
__device__ __forceinline__ float tanh_f(float v) { return 2.f * __builtin_amdgcn_rcpf(1.f + __expf(-2.f * v)) - 1.f; }

// value of lane (l ^ 1), (l ^ 2), (l ^ 4) by DPP (a VALU move, no trip through the LDS crossbar like ds_bpermute): quad permutes for 1 and 2;
// for 4 the two row shifts, each written only to the banks (groups of 4 lanes) it is valid for
__device__ __forceinline__ float lane_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1, 0, 3, 2]
}
__device__ __forceinline__ float lane_xor2(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2, 3, 0, 1]
}
__device__ __forceinline__ float lane_xor4(float v) {
    const int x = __builtin_bit_cast(int, v);
    int r = __builtin_amdgcn_update_dpp(0, x, 0x104, 0xF, 0x5, false);  // row_shl:4 -> banks 0, 2 (lanes with bit 2 clear read lane + 4)
    r = __builtin_amdgcn_update_dpp(r, x, 0x114, 0xF, 0xA, false);      // row_shr:4 -> banks 1, 3 (lanes with bit 2 set read lane - 4)
    return __builtin_bit_cast(float, r);
}

// grid (2 directions, B sequences).  whh: fp16 pairs packed [dir][thread][8 rows][16 pairs]; gx [dir][B * T][GR]; bhn [dir][GH];
// y [B][T][2 * GH] fp32; hn [2][B][GH] fp32 (final states, may be null)
static __global__ void __launch_bounds__(GNT) k_gru_seq(const half2v* __restrict__ whh, const float* __restrict__ gx, const float* __restrict__ bhn,
                                                        float* __restrict__ y, float* __restrict__ hn, int B, int T) {
    __shared__ __attribute__((aligned(16))) _Float16 hbuf[2][GH];
    __shared__ float gh[GR];
    const int dir = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, ks = tid & 7;
    half2v w[8][16];
    {
        const half2v* wp = whh + ((size_t)dir * GNT + tid) * 128;
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) w[r][i] = wp[r * 16 + i];
    }
    if (tid < GH) hbuf[0][tid] = (_Float16)0.f;
    const bool gate = tid < GH;  // threads 0..255 also own hidden unit `tid`
    const float* gxb = gx + ((size_t)dir * B + b) * (size_t)T * GR;
    const float bn = gate ? bhn[dir * GH + tid] : 0.f;
    float hprev = 0.f;
    float g[GPD][3];
    auto tstep = [&](int s) { return dir ? T - 1 - s : s; };
    if (gate) {
#pragma unroll
        for (int u = 0; u < GPD; ++u) {
            const float* p = gxb + (size_t)tstep(min(u, T - 1)) * GR + tid;
            g[u][0] = p[0];
            g[u][1] = p[GH];
            g[u][2] = p[2 * GH];
        }
    }
    __syncthreads();
    for (int s0 = 0; s0 < T; s0 += GPD) {
#pragma unroll
        for (int u = 0; u < GPD; ++u) {
            const int s = s0 + u;
            if (s >= T) break;
            const int cur = s & 1;
            // ---- this thread's 8 rows x 32 inputs of W_hh . h ----
            float a[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) a[r] = 0.f;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {  // (two halves: 8 registers of h at a time)
                const uint4* hp = (const uint4*)(&hbuf[cur][ks * 32 + hf * 16]);
                const uint4 q0 = hp[0], q1 = hp[1];
                const half2v hq[8] = {__builtin_bit_cast(half2v, q0.x), __builtin_bit_cast(half2v, q0.y), __builtin_bit_cast(half2v, q0.z),
                                      __builtin_bit_cast(half2v, q0.w), __builtin_bit_cast(half2v, q1.x), __builtin_bit_cast(half2v, q1.y),
                                      __builtin_bit_cast(half2v, q1.z), __builtin_bit_cast(half2v, q1.w)};
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) a[r] = __builtin_amdgcn_fdot2(w[r][hf * 8 + i], hq[i], a[r], false);
            }
            // ---- transposing reduction over the 8 k-slice lanes: lane ks ends with the full sum of row ks (7 exchanges instead of 24) ----
            float c4[4], c2[2], c1;
            {
                const bool up = ks & 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float keep = up ? a[4 + i] : a[i], send = up ? a[i] : a[4 + i];
                    c4[i] = keep + lane_xor4(send);
                }
            }
            {
                const bool up = ks & 2;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float keep = up ? c4[2 + i] : c4[i], send = up ? c4[i] : c4[2 + i];
                    c2[i] = keep + lane_xor2(send);
                }
            }
            {
                const bool up = ks & 1;
                const float keep = up ? c2[1] : c2[0], send = up ? c2[0] : c2[1];
                c1 = keep + lane_xor1(send);
            }
            gh[tid] = c1;  // (thread tid = 8 * group + ks holds row 8 * group + ks)
            __syncthreads();
            // ---- gates of unit `tid`, new state ----
            if (gate) {
                const float r = sigmoid_f(g[u][0] + gh[tid]);
                const float z = sigmoid_f(g[u][1] + gh[GH + tid]);
                const float n = tanh_f(g[u][2] + r * (gh[2 * GH + tid] + bn));
                hprev = (1.f - z) * n + z * hprev;
                hbuf[cur ^ 1][tid] = (_Float16)hprev;
                const int t = tstep(s);
                y[((size_t)b * T + t) * (2 * GH) + dir * GH + tid] = hprev;
                const float* p = gxb + (size_t)tstep(min(s + GPD, T - 1)) * GR + tid;  // the request for step s + GPD (clamped)
                g[u][0] = p[0];
                g[u][1] = p[GH];
                g[u][2] = p[2 * GH];
            }
            __syncthreads();
        }
    }
    if (gate && hn) hn[((size_t)dir * B + b) * GH + tid] = hprev;
}

}  // namespace

struct rvcmi_gru {
    int device = 0, input = 0;
    DevBuf wih, whh, bias, bhn, gx;
    size_t gx_rows = 0;
};

extern "C" {

int rvcmi_gru_create(int input_size, int hidden_size, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, int device,
                     rvcmi_gru** out) {
    return guarded([&] {
        if (!w_ih || !w_hh || !b_ih || !b_hh || !out) RVCMI_FAIL(RVCMI_ERR_INVALID, "gru_create: null argument");
        if (hidden_size != GH || input_size < 16 || input_size % 16)
            RVCMI_FAIL(RVCMI_ERR_INVALID, "gru_create: hidden_size %d / input_size %d not supported (256; a multiple of 16)", hidden_size, input_size);
        DeviceGuard dg(device);
        std::unique_ptr<rvcmi_gru> h(new rvcmi_gru());
        h->device = device;
        h->input = input_size;
        const int I = input_size;
        {   // W_ih of both directions as fp16, rows = the 2 * 768 stacked features
            std::vector<_Float16> w((size_t)2 * GR * I);
            for (size_t i = 0; i < w.size(); ++i) w[i] = (_Float16)w_ih[i];
            h->wih.alloc(w.size() * 2);
            HIP_CHECK(hipMemcpy(h->wih.p, w.data(), w.size() * 2, hipMemcpyHostToDevice));
        }
        {   // W_hh: thread (group, ks) holds rows 8 * group .. + 7, inputs 32 * ks .. + 31, as 16 fp16 pairs per row
            std::vector<_Float16> w((size_t)2 * GNT * 256);
            for (int d = 0; d < 2; ++d)
                for (int t = 0; t < GNT; ++t)
                    for (int r = 0; r < 8; ++r)
                        for (int k = 0; k < 32; ++k)
                            w[(((size_t)d * GNT + t) * 8 + r) * 32 + k] = (_Float16)w_hh[((size_t)d * GR + (t >> 3) * 8 + r) * GH + (t & 7) * 32 + k];
            h->whh.alloc(w.size() * 2);
            HIP_CHECK(hipMemcpy(h->whh.p, w.data(), w.size() * 2, hipMemcpyHostToDevice));
        }
        {   // b_ih + b_hh for the r and z gates (they add before the sigmoid); b_hn stays inside the r * (...) term
            std::vector<float> bc((size_t)2 * GR), bn((size_t)2 * GH);
            for (int d = 0; d < 2; ++d)
                for (int f = 0; f < GR; ++f) {
                    bc[(size_t)d * GR + f] = b_ih[(size_t)d * GR + f] + (f < 2 * GH ? b_hh[(size_t)d * GR + f] : 0.f);
                    if (f >= 2 * GH) bn[(size_t)d * GH + f - 2 * GH] = b_hh[(size_t)d * GR + f];
                }
            h->bias.alloc(bc.size() * 4);
            HIP_CHECK(hipMemcpy(h->bias.p, bc.data(), bc.size() * 4, hipMemcpyHostToDevice));
            h->bhn.alloc(bn.size() * 4);
            HIP_CHECK(hipMemcpy(h->bhn.p, bn.data(), bn.size() * 4, hipMemcpyHostToDevice));
        }
        *out = h.release();
    });
}

int rvcmi_gru_destroy(rvcmi_gru* h) {
    return guarded([&] { delete h; });
}

int rvcmi_gru_forward(rvcmi_gru* h, int B, int T, const void* x16, float* y, float* hn, void* stream) {
    return guarded([&] {
        if (!h || !x16 || !y) RVCMI_FAIL(RVCMI_ERR_INVALID, "gru_forward: null argument");
        if (B < 1 || T < 1 || (long long)B * T > (1ll << 30)) RVCMI_FAIL(RVCMI_ERR_INVALID, "gru_forward: B = %d, T = %d", B, T);
        DeviceGuard dg(h->device);
        hipStream_t st = (hipStream_t)stream;
        const size_t M = (size_t)B * T;
        if (M > h->gx_rows) {  // (grows with the longest call seen; f0 runs once per file, outside any capture)
            HIP_CHECK(hipStreamSynchronize(st));
            h->gx.alloc(M * 2 * GR * sizeof(float));
            h->gx_rows = M;
        }
        hipLaunchKernelGGL(k_gru_xproj, dim3((unsigned)((M + 31) / 32), 2 * GR / 128), dim3(256), 0, st, (const _Float16*)x16, h->wih.as<_Float16>(),
                           h->bias.as<float>(), h->gx.as<float>(), (int)M, h->input);
        hipLaunchKernelGGL(k_gru_seq, dim3(2, B), dim3(GNT), 0, st, h->whh.as<half2v>(), h->gx.as<float>(), h->bhn.as<float>(), y, hn, B, T);
        HIP_CHECK(hipGetLastError());
    });
}

}  // extern "C"
