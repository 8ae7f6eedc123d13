python bench.py > gpurun_out/r02_bench_b1.json 2> gpurun_out/r02_bench_b1.err; tail -1 gpurun_out/r02_bench_b1.err
python -c "
import json
d=json.load(open('gpurun_out/r02_bench_b1.json')); r=d['roofline']
print(round(d['ms_per_step'],4), round(d['value'],1), d['repeats']['ms_per_step_median'], r['frac'], r['kernels_ms_per_step'])
print(d['whole_infer']['ms_per_step'], d['gpu_torch_baseline']['fp16']['ms_per_clip'], d['cpu_baseline']['value'])
"
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "sclk\|power\|temp" | head -8
