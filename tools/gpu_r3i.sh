#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_generator.py tests/test_gpu_glue.py -x -q -k "not batch_64 and not batch_16 and not full_clip" 2>&1 | tail -6
for v in 0 1; do
RVCMI_NO_RB_SPLIT=$v timeout 300 python bench.py --stream > gpurun_out/r3i_stream_nosplit$v.json 2>/dev/null
python - <<PY
import json
d=json.load(open('gpurun_out/r3i_stream_nosplit$v.json'))
print('NO_RB_SPLIT=$v hot', d['hot_path']['p50_ms'], d['hot_path']['p99_ms'], 'whole', d['whole_chunk']['p50_ms'], d['whole_chunk']['p99_ms'])
PY
done
