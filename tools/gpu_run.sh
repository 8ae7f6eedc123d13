pr() { python -c "
import json,sys
d=json.load(open('gpurun_out/$1.json')); k=d['roofline']['kernels_ms_per_step']
print('$1', round(d['ms_per_step'],4), d.get('repeats',{}).get('ms_per_step_median'), {n:v for n,v in k.items() if n.startswith('rb_')})
"; }
python -m pytest tests/test_gpu_generator.py tests/test_gpu_dropin.py -m gpu -q -x 2>&1 | tail -2
python bench.py --steps 50 --warmup 5 --repeats 5 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/b_main.json 2>gpurun_out/b_main.err; pr b_main; tail -2 gpurun_out/b_main.err
RVCMI_NO_FORK=1 python bench.py --steps 50 --warmup 5 --repeats 5 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/b_nofork.json 2>/dev/null; pr b_nofork
python bench.py --graph 0 --steps 50 --warmup 5 --repeats 3 --no-cpu-baseline --no-gpu-torch-baseline --no-roofline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('eager', round(d['ms_per_step'],4))"
python bench.py --stream 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('  stream', d['hot_path']['p50_ms'], d['whole_chunk']['p50_ms'])"
RVCMI_NO_FORK=1 python bench.py --stream 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('  stream nofork', d['hot_path']['p50_ms'], d['whole_chunk']['p50_ms'])"
