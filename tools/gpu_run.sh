python bench.py --steps 30 --warmup 3 --repeats 2 --no-cpu-baseline --no-gpu-torch-baseline > gpurun_out/b_ivf.json 2>gpurun_out/b_ivf.err; tail -2 gpurun_out/b_ivf.err
python -c "
import json
d=json.load(open('gpurun_out/b_ivf.json')); r=d['roofline']
print(d['ms_per_step']); print(json.dumps(r['ivf_scan_hbm'],indent=0)); print(json.dumps(r.get('ivf_clustered_index'),indent=0))
"
