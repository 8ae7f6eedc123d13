// Dev micro-benchmark (round 5): what does a grid-wide barrier cost on an MI355X (8 XCDs, one L2 each) against a kernel boundary?
// The front at B = 1 is ~47 dependent launches of a few dozen blocks; a persistent kernel per coupling layer / encoder
// layer with grid barriers between its phases only pays if a barrier is much cheaper than the launch it replaces.
//   hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier && ./grid_barrier
// Variants: (a) counter barrier with agent-scope release / acquire fences around it (what exchanging data through global memory
// across XCDs needs), each block also writes and then reads 4 KB of a neighbour's data per phase; (b) the same without the data;
// (c) N dependent empty kernel launches (graph replay) for comparison; (d) N dependent launches that move the same 4 KB per block.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);  // agent scope (default for __atomic on global memory)
        while (__atomic_load_n(counter, __ATOMIC_ACQUIRE) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

__global__ void __launch_bounds__(256) k_persist(unsigned* counter, float4* buf, int phases, int with_data) {
    const unsigned G = gridDim.x;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = 0; p < phases; ++p) {
        if (with_data) {  // write my 4 KB, later read my neighbour's
            buf[(size_t)blockIdx.x * 256 + threadIdx.x] = make_float4((float)p, acc.x, acc.y, acc.z);
            __threadfence();
        }
        grid_barrier(counter, (unsigned)(p + 1) * G);
        if (with_data) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // every wave: its loads must not hit stale lines
            const float4 v = buf[(size_t)((blockIdx.x + 37) % G) * 256 + threadIdx.x];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    if (acc.x == 123.456f) buf[0] = acc;
}
__global__ void __launch_bounds__(256) k_phase(float4* buf, int p, int with_data) {
    if (with_data) {
        const float4 v = buf[(size_t)((blockIdx.x + 37) % gridDim.x) * 256 + threadIdx.x];
        buf[(size_t)gridDim.x * 256 + (size_t)blockIdx.x * 256 + threadIdx.x] = make_float4((float)p, v.x, v.y, v.z);
    }
}

int main() {
    unsigned* counter;
    float4* buf;
    CK(hipMalloc(&counter, 256));
    CK(hipMalloc(&buf, (size_t)2 * 1024 * 256 * 16));
    CK(hipMemset(buf, 0, (size_t)2 * 1024 * 256 * 16));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int phases = 64;
    for (int G : {38, 152, 256}) {
        for (int with_data = 0; with_data < 2; ++with_data) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipMemsetAsync(counter, 0, 4, st));
                CK(hipEventRecord(e0, st));
                hipLaunchKernelGGL(k_persist, dim3(G), dim3(256), 0, st, counter, buf, phases, with_data);
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) best = ms < best ? ms : best;
            }
            printf("persistent  G=%3d data=%d: %6.2f us per phase (%d phases, launch included)\n", G, with_data, 1e3f * best / phases, phases);
            // the same number of dependent launches, replayed from a graph
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(k_phase, dim3(G), dim3(256), 0, st, buf, p, with_data);
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipEventRecord(e0, st));
                CK(hipGraphLaunch(ge, st));
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) best = ms < best ? ms : best;
            }
            printf("graph of launches G=%3d data=%d: %6.2f us per launch\n", G, with_data, 1e3f * best / phases);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
    }
    return 0;
}
