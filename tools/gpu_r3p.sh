#!/bin/bash
# last sanity of the committed build: smoke + the tests nearest to the latest changes + one default bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python __graft_entry__.py smoke 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_generator.py tests/test_gpu_front.py -m gpu -x -q -k "kl2 or full_clip or front_matches or smoke or unfused" 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/r3p_b1.json 2>/dev/null
python -c "
import json
d=json.load(open('gpurun_out/r3p_b1.json')); print(round(d['ms_per_step'],4), round(d['value'],1), d['roofline']['frac'], d['whole_infer']['ms_per_step'])"
