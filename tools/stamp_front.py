import os, sys, torch
sys.path.insert(0, "/root/repo")
import rvc_amd
from oracle import synth
from oracle.front_oracle import FrontConfig
gpu = torch.device("cuda:0"); B=1; T=1198
fcfg = FrontConfig(); wf = synth.make_front_weights(fcfg, 1)
fr = rvc_amd.FrontHIP(vars(fcfg), wf, device=gpu, operand="fp16", max_B=B, max_T=T)
phone = synth.make_phone(1, T, 768, 1).to(gpu); pitch = synth.make_pitch(synth.make_f0(B, T)).to(gpu)
g = wf["emb_g.weight"][:B].to(gpu); nz = torch.randn(B, 192, T, device=gpu)
for _ in range(3): fr(phone, pitch, None, g, 0, noise=nz)
torch.cuda.synchronize()
