#!/bin/bash
# -mllvm -amdgpu-mfma-vgpr-form=1 build: full GPU suite + A/B against the previous build (librvcmi_base.so)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r3q_pytest.txt; tail -2 gpurun_out/r3q_pytest.txt
run() {  # name batch env...
  name=$1; b=$2; shift; shift
  env "$@" timeout 600 python bench.py --batch $b --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/r3q_$name.json 2>gpurun_out/r3q_$name.err
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r3q_$name.json'))
    k=d['roofline']['kernels_ms_per_step']
    w=d.get('whole_infer') or {}
    print('$name', 'ms/clip', round(d['ms_per_step']/$b,4), 'median', round(d['repeats']['ms_per_step_median']/$b,4), 'rtf', round(d['value'],1), {n: round(v/$b,4) for n,v in k.items() if n.startswith('rb_') or n.startswith('ups') or n=='conv_post'}, 'whole', w.get('ms_per_step'), 'front', w.get('front_ms_per_step'))
except Exception as e:
    print('$name FAILED', e); print(open('gpurun_out/r3q_$name.err').read()[-1500:])
PY
}
BASE=$PWD/retrieval-based-voice-conversion-webui_amd/librvcmi_base.so
run b1_base 1 RVCMI_LIB=$BASE
run b1_vf 1 A=1
run b1_base2 1 RVCMI_LIB=$BASE
run b1_vf2 1 A=1
run b16_base 16 RVCMI_LIB=$BASE
run b16_vf 16 A=1
