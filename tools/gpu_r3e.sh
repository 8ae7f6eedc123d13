#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_glue.py -x -q 2>&1 | tail -15
