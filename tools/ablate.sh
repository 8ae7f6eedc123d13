#!/bin/bash
# Timing ablations of the generator kernels (RVCMI_DBG bit mask; outputs are wrong by construction).
cd ${GRAFT_REPO_ROOT:-/root/repo}
for m in 0 1 2 4 6 8 16 7 31; do
  RVCMI_DBG=$m python bench.py --steps 3 --warmup 1 --no-cpu-baseline --graph 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']
print('dbg=%-3s' % '$m', ' '.join('%s=%.3f' % (a.replace('rb_pair_','rb').replace('ups_','up'), b) for a,b in k.items() if a.startswith(('rb_','ups_'))))"
done
