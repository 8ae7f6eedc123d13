#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
run() {  # name batch env...
  name=$1; b=$2; shift; shift
  env "$@" timeout 600 python bench.py --batch $b --steps 10 --warmup 3 --repeats 0 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/r3g_$name.json 2>/dev/null
  python - <<PY
import json
d=json.load(open('gpurun_out/r3g_$name.json'))
print('$name', 'ms/clip', round(d['ms_per_step']/$b,4), 'rb_stream_c128/clip', round(d['roofline']['kernels_ms_per_step']['rb_stream_c128']/$b,4), 'rtf', round(d['value'],1))
PY
}
run b1_v1 1 RVCMI_RS_V2=0
run b1_v2 1 RVCMI_RS_V2=1
run b4_v1 4 RVCMI_RS_V2=0
run b4_v2 4 RVCMI_RS_V2=1
run b16_v1 16 RVCMI_RS_V2=0
run b16_v2 16 RVCMI_RS_V2=1
run b64_v1 64 RVCMI_RS_V2=0
run b64_v2 64 RVCMI_RS_V2=1
run b16_v2_prio 16 RVCMI_RS_V2=1 RVCMI_RS_PRIO=1
run b16_v2_skew0 16 RVCMI_RS_V2=1 RVCMI_RS_SKEW=0
RVCMI_RS_V2=1 RVCMI_RS_STAMPS=1 timeout 300 python bench.py --batch 16 --steps 1 --warmup 1 --repeats 0 --no-cpu-baseline --no-gpu-torch-baseline --graph 0 2>&1 >/dev/null | grep "rs stamps" | tail -3
