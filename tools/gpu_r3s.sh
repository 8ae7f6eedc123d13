#!/bin/bash
# LDS-only barriers in k_rb_pair / k_rb_full: parity subset + same-box A/B against the previous build
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_generator.py -m gpu -x -q -k "reference_golden or full_clip_voiced or every_shipped or full_clip_size" 2>&1 | tail -2
BASE=$PWD/retrieval-based-voice-conversion-webui_amd/librvcmi_base.so
for v in base new base new; do
  if [ $v = base ]; then export RVCMI_LIB=$BASE; else unset RVCMI_LIB; fi
  timeout 300 python bench.py --batch 1 --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline --no-gpu-torch-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']; print('$v', round(d['ms_per_step'],4), round(d['repeats']['ms_per_step_median'],4), {n:k[n] for n in ('rb_pair_c256','rb_full_c64','rb_full_c32','rb_stream_c128')})"
done
