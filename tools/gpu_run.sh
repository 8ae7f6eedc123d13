pr() { python -c "
import json,sys
d=json.load(open('gpurun_out/$1.json')); k=d['roofline']['kernels_ms_per_step']
print('$1', round(d['ms_per_step'],4), d.get('repeats',{}).get('ms_per_step_median'), {n:v for n,v in k.items() if n.startswith('rb_')})
"; }
python bench.py --steps 50 --warmup 5 --repeats 5 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/b_main.json 2>/dev/null; pr b_main
RVCMI_LIB=$PWD/build_alt/librvcmi_nt.so python bench.py --steps 50 --warmup 5 --repeats 5 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/b_nt.json 2>/dev/null; pr b_nt
RVCMI_LIB=$PWD/build_alt/librvcmi_nt.so python bench.py --batch 16 --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/b_nt16.json 2>/dev/null; pr b_nt16
python bench.py --batch 16 --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/b_16.json 2>/dev/null; pr b_16
RVCMI_LIB=$PWD/build_alt/librvcmi_nt.so RVCMI_RS_STAMPS=1 python bench.py --steps 2 --warmup 1 --repeats 0 --graph 0 --no-cpu-baseline --no-gpu-torch-baseline --no-roofline 2>&1 >/dev/null | grep "rs stamps" | tail -3
