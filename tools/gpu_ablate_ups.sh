#!/bin/bash
# Timing ablations of k_ups (RVCMI_DBG bit mask: 2 = no MFMA K loop, 4 = no VALU noise conv, 16 = no stores; outputs are wrong by construction).
# One lease, interleaved: tools/gpu_ablate_ups.sh [rounds]
export RVCMI_BENCH_ABLATION=1
rounds=${1:-2}
for r in $(seq 1 $rounds); do
  for m in 0 2 16 18; do
    RVCMI_DBG=$m python bench.py --steps 5 --warmup 2 --repeats 0 --no-cpu-baseline --no-gpu-torch-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']
print('r$r dbg=%-3s' % '$m', ' '.join('%s=%.4f' % (a, b) for a,b in k.items() if a.startswith('ups_')))"
  done
done
