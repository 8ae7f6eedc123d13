#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -X faulthandler -m pytest tests -m gpu -x -q 2>&1 | grep -v "^  File \"/usr" | tail -12
