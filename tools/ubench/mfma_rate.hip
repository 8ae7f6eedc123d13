// Micro-benchmark: MFMA issue rate and s_memtime tick rate under load, one wave per SIMD (256 blocks x 256 threads).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
template <int NACC>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* ticks, int iters, float seed) {
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(seed * (threadIdx.x % 7 + e) * 0.01f); b[e] = (_Float16)(seed * (threadIdx.x % 5 - e) * 0.02f); }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int NACC>
void run(int blocks, int iters, float seed) {
    float* out; unsigned long long* ticks;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&ticks, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, ticks, 10, seed);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, ticks, iters, seed);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), ticks, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    double n = (double)iters * NACC;
    double tf = n * 4 * blocks * 32768.0 / (ms * 1e-3) / 1e12;
    printf("NACC=%d blocks=%d seed=%.1f: %.3f ms, %.1f ticks/MFMA, tick rate %.3f GHz, => %.1f ns/MFMA/SIMD, %.0f TFLOP/s\n", NACC, blocks, seed,
           ms, avg / n, avg / (ms * 1e6), ms * 1e6 / n, tf);
    hipFree(out); hipFree(ticks);
}
int main() {
    run<6>(256, 20000, 0.f);   // zeros
    run<6>(256, 20000, 1.f);   // non-trivial operands
    run<2>(256, 20000, 1.f);
    run<8>(256, 20000, 1.f);
    run<6>(32, 20000, 1.f);    // few CUs busy
    return 0;
}
