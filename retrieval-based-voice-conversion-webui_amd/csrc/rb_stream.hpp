// Host interface of the streaming fused ResBlock kernel (rb_stream_kernels.hpp); implemented in rb_stream.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace rvcmi {

struct RbStreamDesc {  // one resblock (ND = 3: all its pairs) or one pair level of it (ND = 1)
    const float* src;
    float* dst;
    const void* w1[3];
    const void* w2[3];
    const float* b1[3];
    const float* b2[3];
    long ct1, ct2;
    int k, k_p;
    int dil[3];
    int y_half;  // (jobs[0] decides for the launch) dst streams are fp16; only k_rb_stream itself honours it
    int x_half;  // (jobs[0]) src is an fp16 stream (round 5: X0 of the streaming stage, option X0_F16); needs the lean-K-loop instantiation
    const int* lens;  // (jobs[0]) ragged batch: item b holds lens[b] * lmul rows (nsf_kernels.hpp item_rows); nullptr = all L
    int lmul;
};

// CUs of the current device (cached per device).
int num_cus();
// Sets the per-device kernel attributes (dynamic LDS > 64 KB) of every instantiation; called from rvcmi_nsf_create.
void rb_stream_prepare();
// Channel counts / fusion depths the kernel is instantiated for.
bool rb_stream_supported(int operand, int C, int nd);
// Option keys read from `opt` (dev / tests): RS_SMALL (0 = the larger time tiles), RS_KL (2 = k_rb_stream with the lean K loop
// kconv), RS_C0 (planning constant), RS_STAMPS (print phase stamps; syncs).
inline void rb_stream_load_env(Options& opt) { opt.load_env({"RS_SMALL", "RS_KL", "RS_C0", "RS_STAMPS"}); }
// Plans strips for `B` utterances of `L` rows and launches ONE kernel covering all `njobs` resblocks.  Returns false
// (nothing launched) when the strips would be too short for the persistent walk to pay and `force` is not set.
// `dry_run`: plan only (same return value), launch nothing.
bool rb_stream_launch(int operand, int C, int nd, const RbStreamDesc* jobs, int njobs, int L, int B, long bstride, bool force,
                      hipStream_t st, const Options& opt, bool dry_run = false);

}  // namespace rvcmi
