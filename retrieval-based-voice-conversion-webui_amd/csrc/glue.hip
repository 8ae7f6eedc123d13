// C entry points of the device-resident glue; kernels in glue_kernels.hpp.
#include <atomic>
#include <cmath>

#include "common.hpp"
#include "glue_kernels.hpp"

using namespace rvcmi;

// hipFuncAttributeMaxDynamicSharedMemorySize is per device: set it once for every device a call is made on (a process-global
// flag would leave the second GPU of a multi-GPU process without it).
template <typename K>
static void ensure_dyn_lds(K kernel, int bytes, std::atomic<unsigned long long>& done_mask) {
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (done_mask.load(std::memory_order_acquire) & bit) return;
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done_mask.fetch_or(bit, std::memory_order_release);
}
static std::atomic<unsigned long long> g_attr_f0{0}, g_attr_sola{0};

extern "C" {

int rvcmi_glue_expand_protect(const float* feats, int64_t nq, int d, int reps, const float* pitchf, float protect, int64_t p_len,
                              float* out, void* stream) {
    return guarded([&] {
        if (!feats || !out || nq < 0 || d < 1 || reps < 1 || p_len < 0 || p_len > nq * reps)
            RVCMI_FAIL(RVCMI_ERR_INVALID, "expand_protect: bad argument (nq %lld d %d reps %d p_len %lld)", (long long)nq, d, reps, (long long)p_len);
        if (!nq || !p_len) return;
        hipLaunchKernelGGL(k_blend_expand, dim3((unsigned)nq), dim3(256), 0, (hipStream_t)stream, feats, nullptr, nullptr, nullptr, d, 0,
                           (int64_t)0, 0.f, 0.f, nullptr, 0, pitchf, protect, p_len, reps, out);
        HIP_CHECK(hipGetLastError());
    });
}

int rvcmi_glue_rmvpe_f0(const float* salience, int n, int nbins, float thred, int p_len, int f0_up_key, double* scratch,
                        int64_t* pitch, float* pitchf, void* stream) {
    return guarded([&] {
        if (!salience || !scratch || !pitch || !pitchf || n < 1 || nbins < 1 || p_len < 1)
            RVCMI_FAIL(RVCMI_ERR_INVALID, "rmvpe_f0: bad argument");
        // work arrays: LDS when (n + p_len) doubles fit, else global (scratch in place + the pitch buffer) -- no length limit
        size_t smem = (size_t)(n + p_len) * sizeof(double);
        const bool in_lds = smem <= 160 * 1024;
        if (!in_lds) smem = 0;
        hipStream_t st = (hipStream_t)stream;
        hipLaunchKernelGGL(k_rmvpe_decode, dim3((n + 3) / 4), dim3(256), 0, st, salience, n, nbins, thred, scratch);
        ensure_dyn_lds(k_f0_post, 160 * 1024, g_attr_f0);
        // the host evaluates the scalars exactly as the reference does (python floats / math.log, rvc/f0/gen.py:18, 70-73)
        const double key_mul = std::pow(2.0, (double)f0_up_key / 12.0);
        const double mel_min = 1127.0 * std::log(1.0 + 50.0 / 700.0), mel_max = 1127.0 * std::log(1.0 + 1100.0 / 700.0);
        hipLaunchKernelGGL(k_f0_post, dim3(1), dim3(256), smem, st, scratch, n, p_len, 1, 1, key_mul, mel_min, mel_max, pitch, pitchf,
                           in_lds ? nullptr : scratch, in_lds ? nullptr : reinterpret_cast<double*>(pitch));
        HIP_CHECK(hipGetLastError());
    });
}

int rvcmi_glue_f0_post(const double* f0, int n, int f0_up_key, int64_t* pitch, float* pitchf, void* stream) {
    return guarded([&] {
        if (!f0 || !pitch || !pitchf || n < 1) RVCMI_FAIL(RVCMI_ERR_INVALID, "f0_post: bad argument");
        size_t smem = (size_t)(2 * n) * sizeof(double);
        const bool in_lds = smem <= 160 * 1024;
        if (!in_lds) smem = 0;
        ensure_dyn_lds(k_f0_post, 160 * 1024, g_attr_f0);
        const double key_mul = std::pow(2.0, (double)f0_up_key / 12.0);
        const double mel_min = 1127.0 * std::log(1.0 + 50.0 / 700.0), mel_max = 1127.0 * std::log(1.0 + 1100.0 / 700.0);
        hipLaunchKernelGGL(k_f0_post, dim3(1), dim3(256), smem, (hipStream_t)stream, f0, n, n, 0, 0, key_mul, mel_min, mel_max, pitch, pitchf,
                           (double*)nullptr, in_lds ? nullptr : reinterpret_cast<double*>(pitch));
        HIP_CHECK(hipGetLastError());
    });
}

int rvcmi_glue_scale_int16_range(float* audio, int64_t n, float* scratch256, void* stream) {
    return guarded([&] {
        if (!audio || !scratch256 || n < 0) RVCMI_FAIL(RVCMI_ERR_INVALID, "scale_int16_range: bad argument");
        if (!n) return;
        const int nb = (int)std::min<int64_t>(256, (n + 255) / 256);
        hipStream_t st = (hipStream_t)stream;
        hipLaunchKernelGGL(k_absmax_partial, dim3(nb), dim3(256), 0, st, audio, n, scratch256);
        hipLaunchKernelGGL(k_scale_int16_range, dim3(nb), dim3(256), 0, st, audio, n, scratch256, nb);
        HIP_CHECK(hipGetLastError());
    });
}

int rvcmi_glue_sola(const float* infer_wav, int64_t n, float* sola_buffer, int Lb, int Ls, const float* fade_in,
                    const float* fade_out, int block_frame, float* out_block, int* offset_out, void* stream) {
    return guarded([&] {
        if (!infer_wav || !sola_buffer || !fade_in || !fade_out || !out_block || Lb < 1 || Ls < 0 || block_frame < 1)
            RVCMI_FAIL(RVCMI_ERR_INVALID, "sola: bad argument");
        if ((int64_t)Ls + block_frame + Lb > n)
            RVCMI_FAIL(RVCMI_ERR_INVALID, "sola: chunk of %lld samples is shorter than search %d + block %d + buffer %d", (long long)n, Ls,
                       block_frame, Lb);
        const size_t smem = (size_t)(2 * Lb + Ls) * sizeof(float);
        if (smem > 150 * 1024) RVCMI_FAIL(RVCMI_ERR_NOMEM, "sola: buffer + search window too large");
        ensure_dyn_lds(k_sola, 150 * 1024, g_attr_sola);  // + 2 KB static
        hipLaunchKernelGGL(k_sola, dim3(1), dim3(256), smem, (hipStream_t)stream, infer_wav, sola_buffer, Lb, Ls, fade_in, fade_out,
                           block_frame, out_block, offset_out);
        HIP_CHECK(hipGetLastError());
    });
}

int rvcmi_glue_resample_poly(const float* x, int64_t n, const float* kernel, int orig, int new_, int K, int width, float* out, int64_t n_out,
                             void* stream) {
    return guarded([&] {
        if (!x || !kernel || !out || n < 1 || orig < 1 || new_ < 1 || K < 1 || width < 0 || n_out < 0) RVCMI_FAIL(RVCMI_ERR_INVALID, "resample: bad argument");
        if (n_out == 0) return;
        hipLaunchKernelGGL(k_resample_poly, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n, kernel, orig, new_, K,
                           width, out, n_out);
        HIP_CHECK(hipGetLastError());
    });
}

int rvcmi_glue_change_rms(const float* data1, int64_t n1, int sr1, float* data2, int64_t n2, int sr2, float rate, float* scratch,
                          void* stream) {
    return guarded([&] {
        if (!data1 || !data2 || !scratch || n1 < 1 || n2 < 1 || sr1 < 2 || sr2 < 2) RVCMI_FAIL(RVCMI_ERR_INVALID, "change_rms: bad argument");
        const int h1 = sr1 / 2, h2 = sr2 / 2;
        const int nf1 = 1 + (int)(n1 / h1), nf2 = 1 + (int)(n2 / h2);
        hipStream_t st = (hipStream_t)stream;
        hipLaunchKernelGGL(k_frame_rms, dim3(nf1), dim3(256), 0, st, data1, n1, 2 * h1, h1, nf1, scratch);
        hipLaunchKernelGGL(k_frame_rms, dim3(nf2), dim3(256), 0, st, (const float*)data2, n2, 2 * h2, h2, nf2, scratch + nf1);
        // torch.pow(rms, torch.tensor(1 - rate)): the exponent is the python double rounded to float32
        const float e1 = (float)(1.0 - (double)rate), e2 = (float)((double)rate - 1.0);
        hipLaunchKernelGGL(k_change_rms, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, data2, n2, scratch, nf1, scratch + nf1, nf2, e1, e2);
        HIP_CHECK(hipGetLastError());
    });
}

}  // extern "C"
