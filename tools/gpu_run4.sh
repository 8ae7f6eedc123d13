python -m pytest tests -m gpu -q 2>&1 | tail -15
python bench.py --no-cpu-baseline --repeats 3 > gpurun_out/bench_r2c.json 2> gpurun_out/bench_r2c.err
python -c "
import json;d=json.load(open('gpurun_out/bench_r2c.json'));r=d['roofline'];print('auto',d['ms_per_step'],d.get('repeats',{}).get('ms_per_step_median'),r['kernel'],round(r['frac'],3),{k:v for k,v in r['kernels_ms_per_step'].items() if k.startswith('rb')})"
python bench.py --no-cpu-baseline --repeats 2 --batch 16 --steps 10 --warmup 2 > gpurun_out/bench_r2c_b16.json 2> gpurun_out/bench_r2c_b16.err
python -c "
import json;d=json.load(open('gpurun_out/bench_r2c_b16.json'));r=d['roofline'];print('b16',d['ms_per_step'],d['value'],r['kernel'],round(r['frac'],3),{k:v for k,v in r['kernels_ms_per_step'].items() if k.startswith('rb')})"
RVCMI_RS_STAMPS=1 python - <<'PY' 2>&1 | grep "rs stamps" | tail -6
import sys; sys.path.insert(0,'.')
import torch, rvc_amd
from oracle import nsf_oracle, synth
cfg=nsf_oracle.CONFIGS["v2_48k"]; w=synth.make_dec_weights(cfg,1234)
dev=torch.device("cuda:0")
for B in (1,):
    T=1198
    z,f0,g=synth.make_dec_inputs(cfg,B,T,1234); noise=nsf_oracle.reference_noise(B,T,cfg.upp,1)
    gen=rvc_amd.NSFGeneratorHIP(vars(cfg),w,device=dev,operand="fp16",max_B=B,max_T=T)
    a=(z.to(dev),f0.to(dev),g.to(dev))
    gen(*a,noise=noise.to(dev)); torch.cuda.synchronize()
    gen(*a,noise=noise.to(dev)); torch.cuda.synchronize()
PY
