python -m pytest tests/test_gpu_generator.py -q -k "full_clip or batch_16" 2>&1 | tail -12
export RVCMI_RS_STAMPS=1
for small in 0 1; do
RVCMI_RS_SMALL=$small RVCMI_RB_STREAM=1 python - <<'PY' 2>&1 | grep "rs stamps"
import sys; sys.path.insert(0,'.')
import torch, rvc_amd
from oracle import nsf_oracle, synth
cfg=nsf_oracle.CONFIGS["v2_48k"]; w=synth.make_dec_weights(cfg,1234)
dev=torch.device("cuda:0")
for B in (1,8):
    T=1198
    z,f0,g=synth.make_dec_inputs(cfg,B,T,1234); noise=nsf_oracle.reference_noise(B,T,cfg.upp,1)
    gen=rvc_amd.NSFGeneratorHIP(vars(cfg),w,device=dev,operand="fp16",max_B=B,max_T=T)
    a=(z.to(dev),f0.to(dev),g.to(dev))
    gen(*a,noise=noise.to(dev)); torch.cuda.synchronize()
    print("---- B=%d second call"%B, file=sys.stderr)
    gen(*a,noise=noise.to(dev)); torch.cuda.synchronize()
    del gen
PY
done
