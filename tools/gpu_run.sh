python -m pytest tests/test_gpu_ivf.py tests/test_gpu_glue.py -m gpu -q -x 2>&1 | tail -3
pr() { python -c "
import json,sys
d=json.load(open('gpurun_out/$1.json')); k=d['roofline']['kernels_ms_per_step']
print('$1', round(d['ms_per_step'],4), d.get('repeats',{}).get('ms_per_step_median'), {n:v for n,v in k.items() if n.startswith('ivf')}, d['roofline'].get('ivf_clustered_index',{}).get('scan_us'))
"; }
python bench.py --steps 50 --warmup 5 --repeats 5 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/b_main.json 2>/dev/null; pr b_main
RVCMI_IVF_NOCHUNK=1 python bench.py --steps 50 --warmup 5 --repeats 5 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/b_nochunk.json 2>/dev/null; pr b_nochunk
python bench.py --batch 16 --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/b_16.json 2>/dev/null; pr b_16
RVCMI_IVF_NOCHUNK=1 python bench.py --batch 16 --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/b_16n.json 2>/dev/null; pr b_16n
