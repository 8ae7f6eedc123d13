#!/bin/bash
# exact 3-instruction division by 3 in k_ups / k_post staging: parity (generator + front + dropin goldens) and kernel times
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_generator.py tests/test_gpu_front.py tests/test_gpu_dropin.py -m gpu -x -q -k "not batch_64 and not batch_16 and not streaming" 2>&1 | tail -4 > gpurun_out/r3m_pytest.txt
tail -2 gpurun_out/r3m_pytest.txt
for i in 1 2; do
timeout 600 python bench.py --batch 1 --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/r3m_b1_$i.json 2>/dev/null
python - <<PY
import json
d=json.load(open('gpurun_out/r3m_b1_$i.json'))
k=d['roofline']['kernels_ms_per_step']
print('b1', round(d['ms_per_step'],4), 'median', round(d['repeats']['ms_per_step_median'],4), 'rtf', round(d['value'],1), {n: v for n,v in k.items() if n.startswith('ups') or n.startswith('conv_post') or n.startswith('rb_')})
PY
done
