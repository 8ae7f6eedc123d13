#!/bin/bash
# round 3, call A: K-loop A/B micro-benchmark, generator/front parity with the pinned K loop, a short bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( tools/ubench/kloop_v1; tools/ubench/kloop ) > gpurun_out/r3a_kloop.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r3a_pytest.txt
timeout 300 python bench.py --steps 50 --warmup 5 --repeats 3 --no-cpu-baseline > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err
cat gpurun_out/r3a_kloop.txt gpurun_out/r3a_pytest.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3a_bench.json'))
print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernels_ms_per_step'])
PY
