#!/bin/bash
# dev: phase stamps of k_rb_full (library built with -DRVCMI_DEV_STAMPS)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
RVCMI_DBG=32 timeout 300 python bench.py --batch 1 --steps 1 --warmup 1 --repeats 0 --no-cpu-baseline --no-gpu-torch-baseline --graph 0 2>&1 >/dev/null | grep "rvcmi ts" | tail -12 > gpurun_out/r3n_rbfull_stamps.txt
cat gpurun_out/r3n_rbfull_stamps.txt
