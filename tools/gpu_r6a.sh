#!/bin/bash
# round 6: the new tests with their printed figures
mkdir -p gpurun_out
python -m pytest tests/test_gpu_front.py tests/test_gpu_dropin.py -m gpu -q -k "sweep or fp16_range or launcher_defaults" > gpurun_out/r6a_tests.txt 2>&1; grep -v "^ \|^E\|^$" gpurun_out/r6a_tests.txt | cut -c1-600 | tail -40
