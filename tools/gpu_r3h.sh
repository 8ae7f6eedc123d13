#!/bin/bash
# round 3, call 1 of the re-entered session: full GPU suite, then A/B of the C = 128 streaming ResBlock kernels
# (k_rb_stream vs k_rb_stream3) with phase stamps, the B = 16 point, the realtime chunk.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3h_pytest.txt
tail -5 gpurun_out/r3h_pytest.txt
run() {  # name batch env...
  name=$1; b=$2; shift; shift
  env "$@" timeout 600 python bench.py --batch $b --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/r3h_$name.json 2>gpurun_out/r3h_$name.err
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r3h_$name.json'))
    k=d['roofline']['kernels_ms_per_step']
    print('$name', 'ms/clip', round(d['ms_per_step']/$b,4), 'median', round(d['repeats']['ms_per_step_median']/$b,4), 'rtf', round(d['value'],1), {n: round(v/$b,4) for n,v in k.items() if n.startswith('rb_') or n.startswith('ups') or n.startswith('ivf')})
except Exception as e:
    print('$name FAILED', e); print(open('gpurun_out/r3h_$name.err').read()[-1500:])
PY
}
run b1_v1 1 RVCMI_RS_V3=0
run b1_v3 1 RVCMI_RS_V3=1
run b1_v3_c25 1 RVCMI_RS_V3=1 RVCMI_RS_C0=2.5
run b1_v3_c15 1 RVCMI_RS_V3=1 RVCMI_RS_C0=1.5
run b16_v1 16 RVCMI_RS_V3=0
run b16_v3 16 RVCMI_RS_V3=1
for v in 0 1; do
  RVCMI_RS_V3=$v RVCMI_RS_STAMPS=1 timeout 300 python bench.py --batch 1 --steps 1 --warmup 1 --repeats 0 --no-cpu-baseline --no-gpu-torch-baseline --graph 0 2>&1 >/dev/null | grep "rs stamps" | tail -3
done
timeout 600 python bench.py --stream --steps 300 --warmup 20 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/r3h_stream.json 2> gpurun_out/r3h_stream.err
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/r3h_stream.json'))
    print('stream', {k: d[k] for k in d if k in ('value','ms_per_step','unit','metric')})
    print({k: v for k, v in d.items() if 'p50' in str(k) or 'p99' in str(k) or k in ('latency','chunk','hot_path','whole_chunk')})
except Exception as e:
    print('stream FAILED', e); print(open('gpurun_out/r3h_stream.err').read()[-1500:])
PY
