#!/bin/bash
# k_rb_stream with the lean K loop (kconv) + dual-written X tail + LDS-only barriers (RS_KL=2): ubench, parity, A/B, stamps
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 120 tools/ubench/kloop_v2 > gpurun_out/r3k_kloop.txt 2>&1; head -8 gpurun_out/r3k_kloop.txt
timeout 900 python -m pytest tests/test_gpu_generator.py -m gpu -x -q -k "kl2 or full_clip_voiced" 2>&1 | tail -6 > gpurun_out/r3k_pytest.txt
tail -3 gpurun_out/r3k_pytest.txt
run() {  # name batch env...
  name=$1; b=$2; shift; shift
  env "$@" timeout 600 python bench.py --batch $b --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/r3k_$name.json 2>gpurun_out/r3k_$name.err
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r3k_$name.json'))
    k=d['roofline']['kernels_ms_per_step']
    print('$name', 'ms/clip', round(d['ms_per_step']/$b,4), 'median', round(d['repeats']['ms_per_step_median']/$b,4), 'rtf', round(d['value'],1), {n: round(v/$b,4) for n,v in k.items() if n.startswith('rb_')})
except Exception as e:
    print('$name FAILED', e); print(open('gpurun_out/r3k_$name.err').read()[-1500:])
PY
}
run b1_kl1 1 RVCMI_RS_KL=1
run b1_kl2 1 RVCMI_RS_KL=2
run b1_kl2_c4 1 RVCMI_RS_KL=2 RVCMI_RS_C0=3.8
run b16_kl1 16 RVCMI_RS_KL=1
run b16_kl2 16 RVCMI_RS_KL=2
for v in 1 2; do
  RVCMI_RS_KL=$v RVCMI_RS_STAMPS=1 timeout 300 python bench.py --batch 1 --steps 1 --warmup 1 --repeats 0 --no-cpu-baseline --no-gpu-torch-baseline --graph 0 2>&1 >/dev/null | grep "rs stamps" | tail -3
done
