RVCMI_RB_STREAM=1 RVCMI_RS_SMALL=1 python - <<'PY' 2>&1 | tail -40
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch, rvc_amd, numpy as np
from conftest import load_golden, golden_config_and_weights
dev=torch.device("cuda:0")
for rep in range(2):
  for name in ("dec_v2_48k_B1_T70","dec_v1_40k_B1_T20","dec_v1_32k_B1_T16"):
    d=load_golden(name); cfg,w=golden_config_and_weights(d)
    for op in ("fp16","bf16"):
        gen=rvc_amd.NSFGeneratorHIP(vars(cfg),w,device=dev,operand=op,max_B=2,max_T=80)
        a=(torch.from_numpy(d["z"]).to(dev),torch.from_numpy(d["f0"]).to(dev),torch.from_numpy(d["g"]).to(dev))
        nz=torch.from_numpy(d["noise"]).to(dev)
        out=gen(*a,noise=nz)
        print(rep,name,op,"out finite",bool(torch.isfinite(out).all()))
        if not torch.isfinite(out).all():
          for tap in ["up0","stage0","up1","stage1","up2","stage2","up3","stage3"]+(["up4","stage4"] if len(cfg.upsample_rates)==5 else []):
            t=gen.debug_tap(tap,*a,noise=nz)
            bad=~torch.isfinite(t)
            if bad.any():
                idx=bad.nonzero()
                print("   ",tap,tuple(t.shape),"nonfinite",int(bad.sum()),"first",idx[0].tolist(),"last",idx[-1].tolist(),"times",sorted(set(idx[:,2].tolist()))[:12],"chans",sorted(set(idx[:,1].tolist()))[:12])
                break
        del gen
PY
