// Single-rounded fp32 operations that the optimiser can never contract into an FMA.
//
// hipcc's default is -ffp-contract=fast-honor-pragmas, and the HIP header intrinsics __fmul_rn / __fadd_rn are plain
// `a * b` / `a + b` compiled under the HEADER's contraction state: after inlining, `__fadd_rn(__fmul_rn(a, b), c)` is
// fused into one v_fma / v_fmac even inside a function that says `#pragma clang fp contract(off)` (seen in the ISA of
// k_sine_source: rad = f0/sr*n + phase became v_fmac, which moved the NSF excitation by 1-2 ulp of the phase against
// torch's separately rounded multiply and add).  The pragma INSIDE these helpers strips the `contract` flag from the
// instructions themselves, wherever they are inlined.
#pragma once
#include <hip/hip_runtime.h>

namespace rvcmi {

__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}
__device__ __forceinline__ float div_rn(float a, float b) {
#pragma clang fp contract(off)
    return a / b;
}

// x / 3 -- the stage mean of the three ResBlocks (rvc/layers/nsf.py:186, `xs / self.num_kernels`) -- in THREE instructions instead of
// the ~10 of an IEEE division (v_div_scale x2, v_rcp, four fma, v_div_fmas, v_div_fixup), bit-identical to it: Markstein's
// correction q' = q + (x - 3 q) * y with the correctly rounded reciprocal y = RN(1/3).  tools/check_div3.py compares it with the
// IEEE quotient for ALL 2^23 mantissas x 2 signs at seven exponents (scaling by a power of two is exact, so that covers every
// x whose quotient is normal): 0 mismatches.  (x = +-inf gives NaN instead of +-inf, -0 gives +0: neither occurs / matters for
// activations.)
__device__ __forceinline__ float div3_exact(float x) {
    const float y = 0x1.555556p-2f;  // RN32(1/3) = 0x3EAAAAAB
    const float q = mul_rn(x, y);
    const float r = __builtin_fmaf(-3.0f, q, x);  // the exact residual
    return __builtin_fmaf(r, y, q);
}

}  // namespace rvcmi
