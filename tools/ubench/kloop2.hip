// Issue-model micro-benchmark for one- and two-wave-per-SIMD MFMA K loops on gfx950 (round 3).  Every variant runs the same
// structure as the generator's K loop -- per k-step MI*NJ v_mfma_f32_32x32x16_f16, NJ ds_read_b128 (B fragments, LA k-steps ahead),
// MI global_load_dwordx4 (A fragments, one 8-k-step ring refilled in place) -- plus FILL independent VALU instructions per MFMA,
// and reports cycles per MFMA per SIMD.  It answers: what does one extra instruction in an MFMA gap cost a lone wave (the
// model T = max(32, a + b n) fitted in DESIGN.md), does one s_waitcnt per k-step instead of one per MFMA matter, what do
// MI = 2 tiles (half the B reads) buy, and how much of it a second wave per SIMD hides.
// Build / run:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o kloop2 kloop2.hip && ./kloop2   (output of round 3:
// profiles/r03_ubench_kloop2_issue_model.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
using lds_cptr = const __attribute__((address_space(3))) char*;
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(lds_cptr)p; }
__device__ __forceinline__ f16x8 lds_ld(unsigned a) { return *(const __attribute__((address_space(3))) f16x8*)(size_t)a; }

// WAIT: 0 = the compiler's own s_waitcnt (one per MFMA), 1 = one manual s_waitcnt lgkmcnt per k-step in front of the MFMAs
// LOADS: bit 0 = A global loads on, bit 1 = B ds_reads on
template <int MI, int NJ, int LA, int FILL, int WAIT, int LOADS, int NWV>
__global__ void __launch_bounds__(64 * NWV, NWV / 4) kloop2(const _Float16* w, int ksteps, int reps, float* out, unsigned long long* ticks) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    constexpr int STRIDE = 272, ROWS = 32 * NJ + 64;
    for (int i = threadIdx.x; i < ROWS * STRIDE / 4; i += 64 * NWV) ((float*)smem)[i] = 0.001f * (i % 97);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned xl = lds_addr(smem) + (unsigned)(lane & 31) * STRIDE + (unsigned)(lane >> 5) * 16;
    const _Float16* wl = w + (size_t)(wave & 3) * MI * ksteps * 512 + lane * 8;
    f32x16 acc[MI][NJ];
    for (int mi = 0; mi < MI; ++mi) for (int jt = 0; jt < NJ; ++jt) for (int e = 0; e < 16; ++e) acc[mi][jt][e] = 0.f;
    float fz[8];
    for (int i = 0; i < 8; ++i) fz[i] = 0.25f * (lane + i);
    f16x8 A[8][MI], Bf[LA + 1][NJ];
    for (int k = 0; k < 8; ++k) for (int mi = 0; mi < MI; ++mi) A[k][mi] = *(const f16x8*)(wl + (size_t)(mi * ksteps + k) * 512);
    for (int q = 0; q < LA; ++q) for (int jt = 0; jt < NJ; ++jt) Bf[q][jt] = lds_ld(xl + q * 32 + jt * 32 * STRIDE);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        unsigned xr = xl;            // running B base: one LDS row further per 8 k-steps (a tap), like the conv's tap loop
        const _Float16* wr = wl;     // running A base
        for (int k0 = 0; k0 < ksteps; k0 += 8 * (LA + 1)) {  // unrolled so that ring indices are static
            const _Float16* wn = (k0 + 8 * (LA + 1) < ksteps) ? wr + 8 * (LA + 1) * 512 : wl;  // next group (wraps at the end of the conv)
#pragma unroll
            for (int kk = 0; kk < 8 * (LA + 1); ++kk) {
                const unsigned nb = xr + (unsigned)(((kk + LA) & 7) * 32) + (unsigned)(((kk + LA) >> 3) * STRIDE);
                const _Float16* an = (kk + 8 < 8 * (LA + 1)) ? wr + (kk + 8) * 512 : wn + (kk + 8 - 8 * (LA + 1)) * 512;
                if (WAIT == 1 && (LOADS & 2)) __builtin_amdgcn_s_waitcnt(0xC07F | ((NJ * (LA - 1) > 15 ? 15 : NJ * (LA - 1)) << 8));
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int jt = 0; jt < NJ; ++jt) {
                        acc[mi][jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[kk & 7][mi], Bf[kk % (LA + 1)][jt], acc[mi][jt], 0, 0, 0);
                        const int i = mi * NJ + jt;
                        if ((LOADS & 2) && i < NJ) Bf[(kk + LA) % (LA + 1)][i] = lds_ld(nb + i * 32 * STRIDE);
                        if ((LOADS & 1) && i >= MI * NJ - MI) A[kk & 7][i - (MI * NJ - MI)] = *(const f16x8*)(an + (size_t)(i - (MI * NJ - MI)) * ksteps * 512);
#pragma unroll
                        for (int f = 0; f < FILL; ++f) {  // one VALU each, pinned to this gap (pure arithmetic is otherwise re-ordered by ISel)
                            float t = fz[f & 7];
                            asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(t));
                            fz[f & 7] = t;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
            }
            xr += (LA + 1) * STRIDE;
            wr += 8 * (LA + 1) * 512;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int mi = 0; mi < MI; ++mi) for (int jt = 0; jt < NJ; ++jt) for (int e = 0; e < 16; ++e) s += acc[mi][jt][e];
    for (int i = 0; i < 8; ++i) s += fz[i];
    out[blockIdx.x * 64 * NWV + threadIdx.x] = s;
    if (lane == 0) ticks[blockIdx.x * NWV + wave] = t1 - t0;
}

template <int MI, int NJ, int LA, int FILL, int WAIT, int LOADS, int NWV>
void run(const char* note) {
    const int ksteps = 96, blocks = 256, reps = 100;  // 96 k-steps = a k = 11 conv at C = 128 (88) rounded to the unroll
    std::vector<_Float16> hw((size_t)4 * MI * ksteps * 512);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (_Float16)(0.01f * ((int)(i % 13) - 6));
    _Float16* w; float* out; unsigned long long* ticks;
    hipMalloc(&w, hw.size() * 2); hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&out, blocks * 64 * NWV * 4); hipMalloc(&ticks, blocks * NWV * 8);
    const size_t smem = (size_t)(32 * NJ + 64) * 272;
    auto kern = &kloop2<MI, NJ, LA, FILL, WAIT, LOADS, NWV>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * NWV), smem, 0, w, ksteps, 2, out, ticks);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * NWV), smem, 0, w, ksteps, reps, out, ticks);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * NWV);
    hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= h.size();
    const double nm = (double)reps * ksteps * MI * NJ;
    const double tf = 2.0 * 32 * 32 * 16 * nm * NWV * blocks / (ms * 1e-3) / 1e12;
    printf("MI=%d NJ=%d LA=%d FILL=%d WAIT=%d LOADS=%d waves/SIMD=%d : %6.1f cycles per SIMD-MFMA  (%.0f TF/s, eff. clock %.2f GHz)  %s\n", MI, NJ, LA, FILL,
           WAIT, LOADS, NWV / 4, avg / nm / (NWV / 4), tf, avg / (ms * 1e-3) / 1e9, note);
    hipFree(w); hipFree(out); hipFree(ticks);
}

int main() {
    printf("-- one wave per SIMD, MI=1 NJ=6 (k_rb_stream's tile): ablations\n");
    run<1, 6, 1, 0, 0, 0, 4>("MFMA only");
    run<1, 6, 1, 0, 0, 1, 4>("A loads only");
    run<1, 6, 1, 0, 0, 2, 4>("B reads only");
    run<1, 6, 1, 0, 0, 3, 4>("the K loop as shipped (B one k-step ahead, compiler waits)");
    run<1, 6, 2, 0, 0, 3, 4>("B two k-steps ahead");
    run<1, 6, 2, 0, 1, 3, 4>("B two k-steps ahead, ONE s_waitcnt per k-step");
    printf("-- cost of fillers for a lone wave (MI=1 NJ=6, full loads)\n");
    run<1, 6, 2, 1, 1, 3, 4>("");
    run<1, 6, 2, 2, 1, 3, 4>("");
    run<1, 6, 2, 4, 1, 3, 4>("");
    run<1, 6, 2, 6, 1, 3, 4>("");
    printf("-- fillers beside bare MFMAs (no loads)\n");
    run<1, 6, 1, 1, 0, 0, 4>("");
    run<1, 6, 1, 2, 0, 0, 4>("");
    run<1, 6, 1, 3, 0, 0, 4>("");
    run<1, 6, 1, 4, 0, 0, 4>("");
    run<1, 6, 1, 6, 0, 0, 4>("");
    run<1, 6, 1, 8, 0, 0, 4>("");
    printf("-- MI=2 tiles (half the B reads per MFMA)\n");
    run<2, 3, 2, 0, 0, 3, 4>("compiler waits");
    run<2, 3, 2, 0, 1, 3, 4>("one wait per k-step");
    run<2, 3, 2, 2, 1, 3, 4>("");
    run<2, 3, 2, 4, 1, 3, 4>("");
    run<2, 4, 2, 0, 1, 3, 4>("MI=2 NJ=4");
    printf("-- NJ=3 halves (k_rb_stream3's slots), one wave per SIMD\n");
    run<1, 3, 2, 0, 0, 3, 4>("compiler waits");
    run<1, 3, 2, 0, 1, 3, 4>("one wait per k-step");
    run<1, 3, 2, 4, 1, 3, 4>("");
    printf("-- two waves per SIMD (8-wave block), NJ=3\n");
    run<1, 3, 2, 0, 0, 3, 8>("compiler waits");
    run<1, 3, 2, 0, 1, 3, 8>("one wait per k-step");
    run<1, 3, 2, 2, 1, 3, 8>("");
    run<1, 3, 2, 4, 1, 3, 8>("");
    run<1, 3, 2, 6, 1, 3, 8>("");
    run<1, 3, 2, 8, 1, 3, 8>("");
    run<2, 2, 2, 0, 1, 3, 8>("MI=2 NJ=2");
    run<2, 2, 2, 4, 1, 3, 8>("MI=2 NJ=2");
    return 0;
}
