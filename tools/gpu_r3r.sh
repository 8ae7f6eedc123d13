#!/bin/bash
# planning constant of the strip planner for the final build (k + c0)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for c in 3.8 3.4 3.0 2.6 3.8; do
RVCMI_RS_C0=$c timeout 300 python bench.py --batch 1 --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/r3r_c$c.json 2>/dev/null
python - <<PY
import json
d=json.load(open('gpurun_out/r3r_c$c.json')); k=d['roofline']['kernels_ms_per_step']
print('c0=$c', round(d['ms_per_step'],4), 'median', round(d['repeats']['ms_per_step_median'],4), 'rb_stream', k['rb_stream_c128'])
PY
done
RVCMI_RS_C0=3.0 timeout 300 python bench.py --batch 16 --steps 10 --warmup 3 --repeats 2 --no-cpu-baseline --no-gpu-torch-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('b16 c0=3.0', d['ms_per_step']/16, d['roofline']['kernels_ms_per_step']['rb_stream_c128']/16)"
RVCMI_RS_C0=3.8 timeout 300 python bench.py --batch 16 --steps 10 --warmup 3 --repeats 2 --no-cpu-baseline --no-gpu-torch-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('b16 c0=3.8', d['ms_per_step']/16, d['roofline']['kernels_ms_per_step']['rb_stream_c128']/16)"
