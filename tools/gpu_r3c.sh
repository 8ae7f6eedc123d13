#!/bin/bash
# round 3, call C: k_rb_stream2 bring-up: parity (generator tests incl. forced-stream cases), A/B vs k_rb_stream, phase stamps
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_generator.py -x -q 2>&1 | tail -8 > gpurun_out/r3c_pytest.txt
cat gpurun_out/r3c_pytest.txt
for v in 1 0; do
  RVCMI_RS_V2=$v timeout 300 python bench.py --steps 50 --warmup 5 --repeats 3 --no-cpu-baseline > gpurun_out/r3c_bench_v2_$v.json 2> gpurun_out/r3c_bench_v2_$v.err
  python - <<PY
import json
d=json.load(open('gpurun_out/r3c_bench_v2_$v.json'))
print('RS_V2=$v', round(d['ms_per_step'],4), round(d['roofline']['frac'],4), d['roofline']['kernels_ms_per_step']['rb_stream_c128'])
PY
done
for c0 in 0.5 2.0 3.0; do
  RVCMI_RS_C0=$c0 timeout 300 python bench.py --steps 30 --warmup 5 --repeats 0 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/r3c_bench_c0_$c0.json 2>/dev/null
  python - <<PY
import json
d=json.load(open('gpurun_out/r3c_bench_c0_$c0.json'))
print('C0=$c0', round(d['ms_per_step'],4), d['roofline']['kernels_ms_per_step']['rb_stream_c128'])
PY
done
RVCMI_RS_STAMPS=1 timeout 300 python bench.py --steps 2 --warmup 1 --repeats 0 --no-cpu-baseline --no-gpu-torch-baseline --graph 0 2>&1 >/dev/null | grep "rs stamps" | tail -3 > gpurun_out/r3c_stamps.txt
cat gpurun_out/r3c_stamps.txt
for b in 16; do
  timeout 300 python bench.py --batch $b --steps 10 --warmup 2 --repeats 0 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/r3c_bench_b$b.json 2>/dev/null
  python - <<PY
import json
d=json.load(open('gpurun_out/r3c_bench_b$b.json'))
print('B=$b', round(d['ms_per_step'],4), round(d['value'],1), d['roofline']['kernels_ms_per_step'].get('rb_stream_c128'))
PY
done
