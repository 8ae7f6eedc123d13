#!/bin/bash
# round 6, lease c: k_post_dma ablations, stage-0 conv-by-conv (RB_SPLIT_BIG), e2e, the range test
mkdir -p gpurun_out
python -m pytest tests/test_gpu_front.py -m gpu -q -k "fp16_range" 2>&1 | grep -v "^ \|^E\|^$" | cut -c1-400 | tail -8
RVCMI_RB_SPLIT_BIG=1 python -m pytest tests/test_gpu_generator.py -m gpu -q -x -k "full_clip or golden" 2>&1 | tail -3
bash tools/gpu_variants.sh postab 2 "" "RVCMI_POST_DBG=1" "RVCMI_POST_DBG=2" "RVCMI_POST_DBG=3" "RVCMI_RB_SPLIT_BIG=1" 2>&1 | sed -e 's/noise_mfma_c256 [0-9.]* //' | cut -c1-420
python bench.py --e2e > gpurun_out/r6c_e2e.json 2> gpurun_out/r6c_e2e.err; tail -3 gpurun_out/r6c_e2e.err | cut -c1-300
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r6c_e2e.json"))
    print("e2e value", d["value"])
    for k, c in d["cases"].items():
        print(k, "wall/clip", c["wall_ms_per_clip"], "rtf", round(c["rtf"], 1), "groups", c["groups_ms_per_clip"], "long pole", c["long_pole"])
        print("   split", c["split_ms_per_clip"])
except Exception as e:
    print("e2e FAILED", e)
PY
