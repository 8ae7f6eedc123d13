#!/bin/bash
# round 6, lease e: split WN layer -- bit-identity vs the fused kernel, parity, ABAB on whole infer
mkdir -p gpurun_out
python - <<'PY'
import sys, torch
sys.path.insert(0, '.')
import rvc_amd
from oracle import synth
from oracle.front_oracle import FrontConfig
dev = torch.device('cuda:0')
fcfg = FrontConfig(); wf = synth.make_front_weights(fcfg, 1234)
for T, B in ((1198, 1), (300, 2), (31, 1)):
    fr = rvc_amd.FrontHIP(vars(fcfg), wf, device=dev, operand='fp16', max_B=B, max_T=T)
    phone = synth.make_phone(B, T, 768, 5).to(dev); pitch = synth.make_pitch(synth.make_f0(B, T)).to(dev)
    lengths = torch.tensor([T, max(1, T - 37)][:B], device=dev); g = wf['emb_g.weight'][:B].unsqueeze(-1).to(dev)
    nz = torch.randn(B, 192, T, generator=torch.Generator().manual_seed(3)).to(dev)
    out = {}
    for v in (0, 2):
        fr.set_option('FR_WN_SPLIT', v)
        out[v] = fr(phone, pitch, lengths, g, 0, noise=nz)
    print('T', T, 'B', B, 'split vs fused bit-equal:', torch.equal(out[0], out[1]), 'max diff', float((out[0] - out[1]).abs().max()))
PY
python -m pytest tests/test_gpu_front.py -m gpu -q -x -k "not sweep" 2>&1 | tail -3; python -m pytest tests/test_gpu_dropin.py -m gpu -q -x 2>&1 | tail -3
bash tools/gpu_variants.sh wn 3 "RVCMI_FR_WN_SPLIT=0" "RVCMI_FR_WN_SPLIT=1" "RVCMI_FR_WN_SPLIT=2" 2>&1 | cut -c1-60
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/var_wn_*_*.json')):
    d = json.load(open(f)); w = d['whole_infer']; k = w['front_kernels_ms_per_step']
    print(f, 'whole', round(w['ms_per_step'], 4), 'front', w['front_ms_per_step'], {n: v for n, v in k.items() if n.startswith('flow_wn')})
PY
