#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --repeats 0 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/r3d_$name.json 2>/dev/null
  python - <<PY
import json
d=json.load(open('gpurun_out/r3d_$name.json'))
print('$name', round(d['ms_per_step'],4), d['roofline']['kernels_ms_per_step']['rb_stream_c128'])
PY
}
run skew0 RVCMI_RS_SKEW=0
run skew1 RVCMI_RS_SKEW=1
run skew2 RVCMI_RS_SKEW=2
run skew3 RVCMI_RS_SKEW=3
run skew2_prio RVCMI_RS_SKEW=2 RVCMI_RS_PRIO=1
run skew0_prio RVCMI_RS_SKEW=0 RVCMI_RS_PRIO=1
run skew2_c3 RVCMI_RS_SKEW=2 RVCMI_RS_C0=3.0
run skew2_c2 RVCMI_RS_SKEW=2 RVCMI_RS_C0=2.0
run skew2_prio_c3 RVCMI_RS_SKEW=2 RVCMI_RS_PRIO=1 RVCMI_RS_C0=3.0
RVCMI_RS_SKEW=2 RVCMI_RS_STAMPS=1 timeout 300 python bench.py --steps 2 --warmup 1 --repeats 0 --no-cpu-baseline --no-gpu-torch-baseline --graph 0 2>&1 >/dev/null | grep "rs stamps" | tail -3
RVCMI_RS_SKEW=2 RVCMI_RS_PRIO=1 RVCMI_RS_STAMPS=1 timeout 300 python bench.py --steps 2 --warmup 1 --repeats 0 --no-cpu-baseline --no-gpu-torch-baseline --graph 0 2>&1 >/dev/null | grep "rs stamps" | tail -3
