#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
tools/ubench/kloop > gpurun_out/r3b_kloop_2waves.txt 2>&1
cat gpurun_out/r3b_kloop_2waves.txt
