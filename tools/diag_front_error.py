"""CPU diagnostic (round 6): WHERE does the front's fp16 error come from?  The oracle's enc_p + z_p + flow^-1 with the MFMA operand rounding
emulated (inputs and weights of one GROUP of convs rounded to fp16, fp32 accumulation; q / k / v also stored as fp16) one group at a time,
against the fp32 result.  Outcome (T = 400, seed 1234; z RMS 1.52): every one of the ten groups contributes 4.1e-4 .. 6.7e-4, their root sum
of squares 1.66e-3 = all groups at once 1.67e-3 (the GPU measures 1.8e-3): no dominant source, hence no cheap fix for the whole-infer parity head
room.  python tools/diag_front_error.py"""
import sys, math, types
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import front_oracle as fo, synth, nsf_oracle
from oracle.front_oracle import FrontConfig
torch.set_num_threads(8)
cfg = FrontConfig(); w = synth.make_front_weights(cfg, 1234)
T = 400
phone = synth.make_phone(1, T, 768, 1234); pitchf = synth.make_f0(1, T); pitch = synth.make_pitch(pitchf)
lengths, sid = torch.tensor([T]), torch.tensor([0])
nz = torch.randn(1, 192, T, generator=torch.Generator().manual_seed(8))
key_of = {id(v): k for k, v in w.items()}
active = set()
h16 = lambda t: t.half().float()
def group(key):
    if 'emb_phone' in key: return 'emb'
    if 'conv_q' in key or 'conv_k' in key or 'conv_v' in key: return 'qkv'
    if 'conv_o' in key: return 'attn_o'
    if 'ffn_layers' in key and 'conv_1' in key: return 'ffn1'
    if 'ffn_layers' in key and 'conv_2' in key: return 'ffn2'
    if 'enc_p.proj' in key: return 'proj'
    if '.pre.' in key: return 'flow_pre'
    if 'in_layers' in key: return 'wn_in'
    if 'res_skip' in key: return 'wn_rs'
    if '.post.' in key: return 'flow_post'
    if 'cond_layer' in key: return 'cond'
    return 'other'
class FP:
    def __getattr__(self, n): return getattr(F, n)
    def conv1d(self, x, weight, bias=None, **kw):
        g = group(key_of.get(id(weight), '?'))
        if g in active and g != 'cond':
            y = F.conv1d(h16(x), h16(weight), bias, **kw)
            if g == 'qkv': y = h16(y)
            return y
        return F.conv1d(x, weight, bias, **kw)
    def linear(self, x, weight, bias=None):
        g = group(key_of.get(id(weight), '?'))
        if g in active: return F.linear(h16(x), h16(weight), bias)
        return F.linear(x, weight, bias)
fo.F = FP()
def run():
    with torch.no_grad():
        z, m1, g = fo.infer_front(cfg, w, phone, pitch, lengths, sid, nz)
    return z * m1
z0 = run()
rms = lambda a, b: float((a - b).pow(2).mean().sqrt())
print('z rms', float(z0.pow(2).mean().sqrt()))
groups = ['emb', 'qkv', 'attn_o', 'ffn1', 'ffn2', 'proj', 'flow_pre', 'wn_in', 'wn_rs', 'flow_post']
tot = 0
for g_ in groups:
    active.clear(); active.add(g_)
    e = rms(run(), z0); tot += e * e
    print('%-10s z err %.3e' % (g_, e))
active.clear(); active.update(groups)
print('all', rms(run(), z0), 'rss of parts', math.sqrt(tot))
active.clear(); active.update([g_ for g_ in groups if g_ not in ('qkv',)])
print('all but qkv', rms(run(), z0))
active.clear(); active.update([g_ for g_ in groups if g_ not in ('proj', 'flow_post')])
print('all but proj+post', rms(run(), z0))
active.clear(); active.update([g_ for g_ in groups if g_ not in ('wn_in', 'wn_rs')])
print('all but wn', rms(run(), z0))
