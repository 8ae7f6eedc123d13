python -m pytest tests/test_gpu_ivf.py tests/test_gpu_glue.py tests/test_gpu_dropin.py tests/test_gpu_front.py -q 2>&1 | tail -15
python bench.py --stream > gpurun_out/bench_stream.json 2> gpurun_out/bench_stream.err; tail -2 gpurun_out/bench_stream.err; cat gpurun_out/bench_stream.json
python bench.py --stream --graph 0 > gpurun_out/bench_stream_eager.json 2>/dev/null; cat gpurun_out/bench_stream_eager.json
