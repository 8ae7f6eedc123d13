#!/bin/bash
# K-loop issue-model micro-benchmarks (tools/ubench/kloop2.hip) and the shipped K loop in both orderings
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 300 tools/ubench/kloop2 > gpurun_out/r3i_kloop2.txt 2>&1; cat gpurun_out/r3i_kloop2.txt
timeout 120 tools/ubench/kloop_v2 > gpurun_out/r3i_kloop_v2.txt 2>&1; head -6 gpurun_out/r3i_kloop_v2.txt
timeout 120 tools/ubench/kloop_v1 > gpurun_out/r3i_kloop_v1.txt 2>&1; head -6 gpurun_out/r3i_kloop_v1.txt
