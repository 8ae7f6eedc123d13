#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_front.py -m gpu -q -k "sweep" > gpurun_out/r6f_sweep.txt 2>&1; grep -v "^ \|^E\|^$" gpurun_out/r6f_sweep.txt | cut -c1-300 | tail -12
