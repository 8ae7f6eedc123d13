#!/bin/bash
# round 6, lease g: attention with eight waves per block -- parity + ABAB on whole infer
mkdir -p gpurun_out
python -m pytest tests/test_gpu_front.py tests/test_gpu_dropin.py -m gpu -q -x -k "not sweep" 2>&1 | tail -3
bash tools/gpu_variants.sh attn 3 "RVCMI_FR_ATTN_W8=0" "RVCMI_FR_ATTN_W8=1" 2>&1 | cut -c1-60
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/var_attn_*_*.json')):
    d = json.load(open(f)); w = d['whole_infer']; k = w['front_kernels_ms_per_step']
    print(f, 'whole', round(w['ms_per_step'], 4), 'front', w['front_ms_per_step'], 'attn', k.get('enc_attn'))
PY
