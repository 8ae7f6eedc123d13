// Dev micro-benchmark: what HBM bandwidth do plain streaming kernels reach on this box (the practical ceiling that the
// activation-streaming kernels should be judged against)?   hipcc --offload-arch=gfx950 -O3 hbm_bw.hip -o hbm_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_copy(const float4* __restrict__ a, float4* __restrict__ o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) o[i] = a[i];
}
__global__ void __launch_bounds__(256) k_sum3(const float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c,
                                              float4* __restrict__ o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 x = a[i], y = b[i], z = c[i];
        o[i] = make_float4(x.x + y.x + z.x, x.y + y.y + z.y, x.z + y.z + z.z, x.w + y.w + z.w);
    }
}
__global__ void __launch_bounds__(256) k_read(const float4* __restrict__ a, float* __restrict__ o, size_t n) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 x = a[i];
        s += x.x + x.y + x.z + x.w;
    }
    if (s == 123.456f) o[0] = s;
}
__global__ void __launch_bounds__(256) k_write(float4* __restrict__ o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) o[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main() {
    const size_t bytes = (size_t)73600000 / 16 * 16 * 4;  // ~294 MB per buffer: 4 activation streams of the c128 stage
    const size_t n = bytes / 16;
    float4 *a, *b, *c, *o;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&o, bytes));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(c, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int grid : {1024, 4096, 16384}) {
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, o, n);
                if (mode == 1) hipLaunchKernelGGL(k_sum3, dim3(grid), dim3(256), 0, 0, a, b, c, o, n);
                if (mode == 2) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, (float*)o, n);
                if (mode == 3) hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, o, n);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double moved = mode == 0 ? 2.0 * bytes : mode == 1 ? 4.0 * bytes : (double)bytes;
            printf("grid %5d %-6s: %.3f ms  %.2f TB/s\n", grid, mode == 0 ? "copy" : mode == 1 ? "sum3" : mode == 2 ? "read" : "write", best,
                   moved / best / 1e9);
        }
    }
    return 0;
}
