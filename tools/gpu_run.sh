python -m pytest tests -m gpu -q 2>&1 | tail -5
python bench.py --batch 64 --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline > gpurun_out/r02_bench_b64.json 2> gpurun_out/r02_bench_b64.err; tail -2 gpurun_out/r02_bench_b64.err
python bench.py --batch 16 --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline > gpurun_out/r02_bench_b16.json 2> gpurun_out/r02_bench_b16.err
python bench.py --whole --batch 16 --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline > gpurun_out/r02_bench_whole_b16.json 2> gpurun_out/r02_bench_whole_b16.err
python -c "
import json
for f in ('r02_bench_b64','r02_bench_b16','r02_bench_whole_b16'):
    d=json.load(open('gpurun_out/%s.json'%f)); r=d.get('roofline',{}); print(f, d['ms_per_step'], d['value'], d.get('repeats',{}).get('ms_per_step_median'), r.get('frac'), {k:v for k,v in r.get('kernels_ms_per_step',{}).items() if k.startswith('rb')})
"
