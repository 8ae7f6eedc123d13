#!/bin/bash
# k_ups with batched unconditional staging loads: parity (whole generator suite) + step time + per-kernel times, against the committed r06 line
mkdir -p gpurun_out
python -m pytest tests/test_gpu_generator.py -m gpu -q -x 2>&1 | tail -2
for r in 1 2 3; do
python bench.py --no-cpu-baseline --no-gpu-torch-baseline --no-extra --repeats 3 > gpurun_out/ups_new_$r.json 2>/dev/null
python - $r <<'PY'
import json, sys
d = json.load(open("gpurun_out/ups_new_%s.json" % sys.argv[1])); k = d["roofline"]["kernels_ms_per_step"]
print("run", sys.argv[1], "step", round(d["ms_per_step"], 4), "median", round(d["repeats"]["ms_per_step_median"], 4), {n: v for n, v in k.items() if n.startswith("ups_") or n == "conv_post"}, "whole", round(d["whole_infer"]["ms_per_step"], 4))
PY
done
