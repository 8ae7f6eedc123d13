#!/bin/bash
# round 6, first lease: the new tests + the default bench line with the batch64 / stream / bf16 objects
mkdir -p gpurun_out
python -m pytest tests/test_gpu_front.py tests/test_gpu_dropin.py -m gpu -q -x -k "sweep or saturate or launcher_defaults" 2>&1 | tail -25 > gpurun_out/r6a_tests.txt; cat gpurun_out/r6a_tests.txt
( time python bench.py > gpurun_out/r6a_bench.json 2> gpurun_out/r6a_bench.err ) 2>&1 | tail -3; tail -3 gpurun_out/r6a_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6a_bench.json"))
print("step", d["ms_per_step"], d["value"], d["roofline"]["frac"])
for k in ("bf16", "batch64", "stream"):
    print(k, json.dumps(d.get(k))[:1500])
print("ubench", d["roofline"].get("ubench_ceiling"))
print(json.dumps(d["roofline"]["kernels_ms_per_step"]))
PY
