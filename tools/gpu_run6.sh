for v in "" NOINT OLDCOPY; do
if [ -n "$v" ]; then export RVCMI_LIB=$PWD/retrieval-based-voice-conversion-webui_amd/librvcmi_$v.so; else unset RVCMI_LIB; fi
echo "== variant '$v'"
RVCMI_RB_STREAM=1 RVCMI_RS_SMALL=1 python - <<'PY' 2>&1 | tail -8
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch, rvc_amd, numpy as np
from conftest import load_golden, golden_config_and_weights
dev=torch.device("cuda:0")
for name in ("dec_v1_40k_B1_T20",):
    d=load_golden(name); cfg,w=golden_config_and_weights(d)
    for op in ("fp16","bf16"):
        gen=rvc_amd.NSFGeneratorHIP(vars(cfg),w,device=dev,operand=op,max_B=2,max_T=80)
        a=(torch.from_numpy(d["z"]).to(dev),torch.from_numpy(d["f0"]).to(dev),torch.from_numpy(d["g"]).to(dev))
        nz=torch.from_numpy(d["noise"]).to(dev)
        out=gen(*a,noise=nz)
        print(name,op,"out finite",bool(torch.isfinite(out).all()))
        t=gen.debug_tap("stage2",*a,noise=nz); bad=~torch.isfinite(t)
        if bad.any():
            idx=bad.nonzero(); ts=sorted(set(idx[:,2].tolist()))
            print("   stage2 nonfinite",int(bad.sum()),"chans",sorted(set(idx[:,1].tolist())),"time runs:",[ (ts[i]) for i in range(len(ts)) if i==0 or ts[i]!=ts[i-1]+1][:12],"n times",len(ts))
        del gen
PY
done
