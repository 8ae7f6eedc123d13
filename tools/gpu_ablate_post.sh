#!/bin/bash
# Timing ablations of conv_post (option POST_DBG, wrong results): the register-staged k_post against the LDS-DMA kernel, whole and staging / streaming only.
export RVCMI_BENCH_ABLATION=1
for r in 1 2; do
  for e in "RVCMI_POST_DMA=0" "RVCMI_POST_DMA=0 RVCMI_POST_DBG=2" "RVCMI_POST_DMA=1" "RVCMI_POST_DMA=1 RVCMI_POST_DBG=2"; do
    env $e python bench.py --steps 5 --warmup 2 --repeats 0 --no-cpu-baseline --no-gpu-torch-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']
print('r$r [$e] conv_post=%.4f ups_c64=%.4f' % (k['conv_post'], k['ups_c64']))"
  done
done
