#!/bin/bash
# split FFN for small grids: front parity + whole-infer bench with per-kernel front times
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_front.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r3o_pytest.txt
tail -3 gpurun_out/r3o_pytest.txt
for v in 0 1; do
RVCMI_FR_FFN_SPLIT=$v timeout 600 python bench.py --batch 1 --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/r3o_b1_split$v.json 2>gpurun_out/r3o_b1_split$v.err
python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r3o_b1_split$v.json'))
    w=d['whole_infer']
    print('split=$v whole ms', round(w['ms_per_step'],4), 'front', w['front_ms_per_step'], w['front_kernels_ms_per_step'])
except Exception as e:
    print('FAILED', e); print(open('gpurun_out/r3o_b1_split$v.err').read()[-1500:])
PY
done
