for m in full pair; do
if [ $m = pair ]; then export RVCMI_RS_PAIR128=1; else unset RVCMI_RS_PAIR128; fi
python bench.py --no-cpu-baseline --repeats 0 --steps 50 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('$m',round(d['ms_per_step'],4),{k:v for k,v in r['kernels_ms_per_step'].items() if k.startswith('rb')})"
done
export RVCMI_RS_PAIR128=1
python bench.py --no-cpu-baseline --repeats 0 --batch 16 --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];print('pair b16',round(d['ms_per_step'],4),{k:v for k,v in r['kernels_ms_per_step'].items() if k.startswith('rb')})"
RVCMI_RS_STAMPS=1 python - <<'PY' 2>&1 | grep "rs stamps" | grep "C=128" | tail -3
import sys; sys.path.insert(0,'.')
import torch, rvc_amd
from oracle import nsf_oracle, synth
cfg=nsf_oracle.CONFIGS["v2_48k"]; w=synth.make_dec_weights(cfg,1234)
dev=torch.device("cuda:0")
B,T=1,1198
z,f0,g=synth.make_dec_inputs(cfg,B,T,1234); noise=nsf_oracle.reference_noise(B,T,cfg.upp,1)
gen=rvc_amd.NSFGeneratorHIP(vars(cfg),w,device=dev,operand="fp16",max_B=B,max_T=T)
a=(z.to(dev),f0.to(dev),g.to(dev))
gen(*a,noise=noise.to(dev)); torch.cuda.synchronize()
PY
python -m pytest tests/test_gpu_generator.py -q -k "full_clip_voiced or streaming" 2>&1 | tail -3
