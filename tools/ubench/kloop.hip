// Micro-benchmark of the generator's K loop (conv_prefetch / conv_run from nsf_kernels.hpp) in isolation:
// one block of 4 waves per CU, each wave runs REPS convs of `k` taps on a resident LDS tile; reports cycles per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../retrieval-based-voice-conversion-webui_amd/csrc/nsf_kernels.hpp"
using namespace rvcmi;
#ifndef VARIANT
#define VARIANT 0
#endif
template <int C, int MI, int NJ, int KG, int NB, bool SHARED = false, int NWV = 4, int KL = 1>
__global__ void __launch_bounds__(64 * NWV, 1) kloop(const _Float16* w, long ct, int k_p, int dil, int reps, int dbg, float* out,
                                                unsigned long long* ticks) {
    using TL = Tile<C>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int rows = (SHARED ? 1 : 4) * 32 * NJ + 64;  // SHARED: all waves read the same rows (k_rb_stream: waves split channels)
    for (int i = threadIdx.x; i < rows * TL::STRIDE / 4; i += 64 * NWV) ((float*)smem)[i] = 0.001f * (i % 97);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const char* xl = smem + (size_t)((SHARED ? 0 : (wave & 3)) * 32 * NJ + (lane & 31)) * TL::STRIDE + (lane >> 5) * 16;
    f32x16 acc[MI][NJ];
    for (int mi = 0; mi < MI; ++mi) for (int jt = 0; jt < NJ; ++jt) for (int e = 0; e < 16; ++e) acc[mi][jt][e] = 0.f;
    typename Op<_Float16>::frag A[NB][KG][MI];
    typename Op<_Float16>::frag A8[8][MI];
    if (SHARED) w += (size_t)__builtin_amdgcn_readfirstlane(wave & 3) * MI * ct;  // each wave streams its own output-channel slice
    unsigned long long t0, t1;
    if constexpr (KL == 2) {  // the lean loop (kconv): raw buffer loads, in-place 8-deep ring
        const __amdgpu_buffer_rsrc_t r = weight_rsrc(w);
        kconv_prefetch<_Float16, MI>(A8, r, lane * 16, (unsigned)(ct * 2));
        t0 = __builtin_readcyclecounter();
        for (int rr = 0; rr < reps; ++rr) {
            kconv<_Float16, C, MI, NJ, TL::STRIDE>(acc, A8, lds_address(xl + (32 - dil * (k_p - 1) / 2) * TL::STRIDE), r, lane * 16, (unsigned)(ct * 2), k_p, dil);
            kconv_prefetch<_Float16, MI>(A8, r, lane * 16, (unsigned)(ct * 2));
        }
        t1 = __builtin_readcyclecounter();
    } else {
        conv_prefetch<_Float16, C, MI, KG, NB>(A, w + lane * 8, ct, k_p);
        t0 = __builtin_readcyclecounter();
        for (int r = 0; r < reps; ++r) {
            conv_run<_Float16, C, MI, NJ, KG, NB>(acc, A, xl, w + lane * 8, ct, k_p, 32 - dil * (k_p - 1) / 2, dil, dbg);
            conv_prefetch<_Float16, C, MI, KG, NB>(A, w + lane * 8, ct, k_p);
        }
        t1 = __builtin_readcyclecounter();
    }
    float s = 0.f;
    for (int mi = 0; mi < MI; ++mi) for (int jt = 0; jt < NJ; ++jt) for (int e = 0; e < 16; ++e) s += acc[mi][jt][e];
    out[blockIdx.x * 64 * NWV + threadIdx.x] = s;
    if (lane == 0) ticks[blockIdx.x * NWV + wave] = t1 - t0;
}
template <int C, int MI, int NJ, int KG, int NB, bool SHARED = false, int NWV = 4, int KL = 1>
void run(int k, int dil, int dbg) {
    using TL = Tile<C>;
    const int CC = C / 16;
    const int tpg = CC >= KG ? 1 : KG / CC;
    const int k_p = (k + tpg - 1) / tpg * tpg;
    const long ct = (long)k_p * CC * 512;
    std::vector<_Float16> hw((size_t)MI * ct * 4);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (_Float16)(0.01f * ((int)(i % 13) - 6));
    _Float16* w; float* out; unsigned long long* ticks;
    const int blocks = 256, reps = 200;
    hipMalloc(&w, hw.size() * 2); hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&out, blocks * 64 * NWV * 4); hipMalloc(&ticks, blocks * NWV * 8);
    const size_t smem = (size_t)((SHARED ? 1 : 4) * 32 * NJ + 64) * TL::STRIDE;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&kloop<C, MI, NJ, KG, NB, SHARED, NWV, KL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((kloop<C, MI, NJ, KG, NB, SHARED, NWV, KL>), dim3(blocks), dim3(64 * NWV), smem, 0, w, ct, k_p, dil, 2, dbg, out, ticks);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((kloop<C, MI, NJ, KG, NB, SHARED, NWV, KL>), dim3(blocks), dim3(64 * NWV), smem, 0, w, ct, k_p, dil, reps, dbg, out, ticks);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * NWV);
    hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= h.size();
    const double nm = (double)reps * k_p * CC * MI * NJ;
    printf("%sC=%d MI=%d NJ=%d KG=%d NB=%d waves=%d k=%d dil=%d: %.1f cycles/MFMA per wave = %.1f per SIMD-MFMA (%.0f cycles per conv)\n", KL == 2 ? "[kconv] " : "", C, MI, NJ, KG, NB, NWV, k, dil,
           avg / nm, avg / nm / (NWV / 4), avg / reps);
    hipFree(w); hipFree(out); hipFree(ticks);
}
int main() {
#ifdef RVCMI_KLOOP_V1
    printf("K loop V1 (round 2: scheduler-ordered reads)\n");
#else
    printf("K loop V2 (pinned slot order)\n");
#endif
#ifdef RVCMI_KLOOP_ABLATE
    printf("ABLATE=%d (1: no weight loads, 2: no B reads)\n", RVCMI_KLOOP_ABLATE);
#endif
    run<128, 1, 6, 4, 2, true, 4, 2>(11, 5, 0);
    run<128, 1, 6, 4, 2, true, 4, 2>(3, 1, 0);
    run<128, 1, 6, 4, 2, true>(3, 1, 0);
    // latency hypothesis: is the K loop bound by the weight loads' L2 latency vs the ring's prefetch distance ((NB-1) * KG k-steps)?
    run<128, 1, 6, 4, 2, true>(11, 5, 0);
    run<128, 1, 6, 4, 3, true>(11, 5, 0);
    run<128, 1, 6, 4, 4, true>(11, 5, 0);
    run<128, 1, 3, 4, 2, true>(11, 5, 0);
    run<128, 1, 3, 4, 3, true>(11, 5, 0);
    run<128, 1, 3, 4, 4, true>(11, 5, 0);
    run<128, 1, 3, 4, 2, true, 8>(11, 5, 0);
    run<128, 1, 3, 4, 4, true, 8>(11, 5, 0);
    run<128, 1, 4, 4, 2, true>(11, 5, 0);
    run<128, 1, 4, 4, 4, true>(11, 5, 0);
    run<64, 2, 2, 4, 2, false, 8>(11, 3, 0);
    run<64, 2, 2, 4, 4, false, 8>(11, 3, 0);
    run<32, 1, 3, 4, 3>(11, 3, 0);
    run<32, 1, 3, 4, 4>(11, 3, 0);
    return 0;
}
