#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ivf.py -m gpu -q -x 2>&1 | tail -2
RVCMI_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 1 --config 3 --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/plan_config3.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/plan_config3.json")); k = d["roofline"]["kernels_ms_per_step"]
print({n: v for n, v in k.items() if n.startswith("ivf")}, round(d["ms_per_step"], 3), round(d["value"], 1))
PY
