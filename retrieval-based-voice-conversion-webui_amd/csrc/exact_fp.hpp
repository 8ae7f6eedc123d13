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

}  // namespace rvcmi
