#!/bin/bash
# round 3: full validation + bench lines + rocprofv3 / PMC profile of the final build
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
echo "(full GPU suite: see r3q_pytest.txt of the same build)"
python __graft_entry__.py smoke 2>&1 | tail -1
python bench.py > gpurun_out/r03_bench_b1.json 2> gpurun_out/r03_bench_b1.err; tail -1 gpurun_out/r03_bench_b1.err
python bench.py --batch 64 --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/r03_bench_b64.json 2>/dev/null
python bench.py --batch 16 --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/r03_bench_b16.json 2>/dev/null
python bench.py --whole --batch 16 --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/r03_bench_whole_b16.json 2>/dev/null
python bench.py --stream > gpurun_out/r03_bench_stream.json 2>/dev/null
bash tools/profile.sh r03 > gpurun_out/prof_r03.log 2>&1; tail -1 gpurun_out/prof_r03.log
RVCMI_RS_STAMPS=1 timeout 300 python bench.py --batch 1 --steps 1 --warmup 1 --repeats 0 --no-cpu-baseline --no-gpu-torch-baseline --graph 0 2>&1 >/dev/null | grep "rs stamps" | tail -3 > gpurun_out/r03_rs_stamps.txt; cat gpurun_out/r03_rs_stamps.txt
python -c "
import json
for f in ('r03_bench_b1','r03_bench_b64','r03_bench_b16','r03_bench_whole_b16'):
    d=json.load(open('gpurun_out/%s.json'%f)); r=d.get('roofline',{}); print(f, round(d['ms_per_step'],4), round(d['value'],1), d.get('repeats',{}).get('ms_per_step_median'), r.get('frac'), r.get('traffic'))
d=json.load(open('gpurun_out/r03_bench_stream.json')); print('stream', d['hot_path'], d['whole_chunk']['p50_ms'], d['whole_chunk']['p99_ms'])
d=json.load(open('gpurun_out/r03_bench_b1.json')); print(d['whole_infer']['ms_per_step'], d['whole_infer']['value'], d['gpu_torch_baseline']['fp16']['ms_per_clip'], d['gpu_torch_baseline']['fp32']['ms_per_clip'], d['cpu_baseline']['value'])
print(d['roofline']['kernels_ms_per_step'])
"
