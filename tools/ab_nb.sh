#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for nb in 2 3 4; do
 for dbg in 0 6; do
  RVCMI_NB=$nb RVCMI_DBG=$dbg python bench.py --steps 5 --warmup 1 --no-cpu-baseline --graph 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']
print('nb=$nb dbg=$dbg', ' '.join('%s=%.3f' % (a, b) for a,b in k.items() if a.startswith(('rb_'))))"
 done
done
