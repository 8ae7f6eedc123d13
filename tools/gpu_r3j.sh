#!/bin/bash
# k_rb_stream2x (two anti-phased strips per block): parity (generator GPU tests) + A/B against k_rb_stream with phase stamps
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_generator.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r3j_pytest.txt
tail -4 gpurun_out/r3j_pytest.txt
run() {  # name batch env...
  name=$1; b=$2; shift; shift
  env "$@" timeout 600 python bench.py --batch $b --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/r3j_$name.json 2>gpurun_out/r3j_$name.err
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r3j_$name.json'))
    k=d['roofline']['kernels_ms_per_step']
    print('$name', 'ms/clip', round(d['ms_per_step']/$b,4), 'median', round(d['repeats']['ms_per_step_median']/$b,4), 'rtf', round(d['value'],1), {n: round(v/$b,4) for n,v in k.items() if n.startswith('rb_')})
except Exception as e:
    print('$name FAILED', e); print(open('gpurun_out/r3j_$name.err').read()[-1500:])
PY
}
run b1_v1 1 RVCMI_RS_V2X=0
run b1_v2x 1 RVCMI_RS_V2X=1
run b1_v2x2 1 RVCMI_RS_V2X=2
run b1_v2x_c05 1 RVCMI_RS_V2X=1 RVCMI_RS_C0=0.5
run b1_v2x_c2 1 RVCMI_RS_V2X=1 RVCMI_RS_C0=2.0
run b16_v1 16 RVCMI_RS_V2X=0
run b16_v2x 16 RVCMI_RS_V2X=1
run b16_v2x2 16 RVCMI_RS_V2X=2
for v in 1 2; do
  RVCMI_RS_V2X=$v RVCMI_RS_STAMPS=1 timeout 300 python bench.py --batch 1 --steps 1 --warmup 1 --repeats 0 --no-cpu-baseline --no-gpu-torch-baseline --graph 0 2>&1 >/dev/null | grep "rs stamps" | tail -3
done
