python -m pytest tests/test_gpu_glue.py -m gpu -q -x -k realtime_vc 2>&1 | grep -v "^$" | tail -25
