python -m pytest tests/test_gpu_generator.py -m gpu -q -x -k "golden or full" 2>&1 | tail -2
pr() { python -c "
import json,sys
d=json.load(open('gpurun_out/$1.json')); k=d['roofline']['kernels_ms_per_step']
print('$1', round(d['ms_per_step'],4), d.get('repeats',{}).get('ms_per_step_median'), {n:v for n,v in k.items() if n.startswith('ups') or n.startswith('rb_') or n=='conv_post'})
"; }
python bench.py --steps 50 --warmup 5 --repeats 5 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/b_main.json 2>/dev/null; pr b_main
