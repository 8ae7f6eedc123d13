#!/bin/bash
# IVF scan: list-sorted XCD-contiguous query order + fp64 query registers; k_rb_stream default = lean K loop
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ivf.py tests/test_gpu_glue.py tests/test_gpu_dropin.py -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r3l_pytest.txt
tail -3 gpurun_out/r3l_pytest.txt
run() {  # name batch env...
  name=$1; b=$2; shift; shift
  env "$@" timeout 600 python bench.py --batch $b --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/r3l_$name.json 2>gpurun_out/r3l_$name.err
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r3l_$name.json'))
    k=d['roofline']['kernels_ms_per_step']
    print('$name', 'ms/clip', round(d['ms_per_step']/$b,4), 'median', round(d['repeats']['ms_per_step_median']/$b,4), 'rtf', round(d['value'],1), {n: round(v/$b,4) for n,v in k.items() if n.startswith('rb_stream') or n.startswith('ivf')}, d['roofline'].get('ivf_clustered_index',{}).get('scan_us'))
except Exception as e:
    print('$name FAILED', e); print(open('gpurun_out/r3l_$name.err').read()[-1500:])
PY
}
run b1_sort0 1 RVCMI_IVF_SORT=0
run b1_sort1 1 RVCMI_IVF_SORT=1
run b16_sort0 16 RVCMI_IVF_SORT=0
run b16_sort1 16 RVCMI_IVF_SORT=1
timeout 300 python tools/bench_ivf.py 2>&1 | tail -6
