python -m pytest tests/test_gpu_ivf.py tests/test_gpu_glue.py -q 2>&1 | tail -3
RVCMI_IVF_STAMPS=1 python bench.py --no-cpu-baseline --no-roofline --repeats 0 --steps 3 --warmup 1 --graph 0 2>&1 | grep "ivf stamps" | tail -2
for i in 1 2; do python bench.py --no-cpu-baseline --repeats 0 --steps 50 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print(round(d['ms_per_step'],4),{k:v for k,v in r['kernels_ms_per_step'].items() if k.startswith('ivf')})"; done
python bench.py --stream 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('  stream', d['hot_path']['p50_ms'], d['whole_chunk']['p50_ms'])"
python tools/bench_ivf.py 2>&1 | tail -4
