python -m pytest tests/test_gpu_generator.py -m gpu -q -x -k "batch_64" 2>&1 | tail -3
