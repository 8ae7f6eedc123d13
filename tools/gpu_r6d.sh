#!/bin/bash
# round 6, lease d: k_post_dma with inline-asm LDS-DMA (depth 1 / 2) -- parity + ABAB; e2e
mkdir -p gpurun_out
RVCMI_POST_DMA=2 python -m pytest tests/test_gpu_generator.py -m gpu -q -x 2>&1 | tail -2
python -m pytest tests/test_gpu_generator.py -m gpu -q -x 2>&1 | tail -2
bash tools/gpu_variants.sh postd 3 "RVCMI_POST_DMA=0" "RVCMI_POST_DMA=1" "RVCMI_POST_DMA=2" "RVCMI_POST_DMA=2 RVCMI_POST_DMA_OCC=1" "RVCMI_POST_DMA=1 RVCMI_POST_DBG=2" "RVCMI_POST_DMA=2 RVCMI_POST_DBG=2" 2>&1 | sed -e 's/noise_mfma_c256 [0-9.]* //' | cut -c1-120
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/var_postd_*_*.json')):
    try:
        d = json.load(open(f)); print(f, d['roofline']['kernels_ms_per_step'].get('conv_post'), round(d['ms_per_step'], 4))
    except Exception as e:
        print(f, 'ERR')
PY
python bench.py --e2e > gpurun_out/r6d_e2e.json 2> gpurun_out/r6d_e2e.err; tail -3 gpurun_out/r6d_e2e.err | cut -c1-300
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r6d_e2e.json"))
    print("e2e value", d["value"])
    for k, c in d["cases"].items():
        print(k, "wall/clip", c["wall_ms_per_clip"], "rtf", round(c["rtf"], 1), "groups", c["groups_ms_per_clip"], "long pole", c["long_pole"])
        print("   split", c["split_ms_per_clip"])
except Exception as e:
    print("e2e FAILED", e)
PY
