"""MEASUREMENT STAND-INS for the two feeders the north star leaves on PyTorch-ROCm: HuBERT-base and the RMVPE f0 network.

No checkpoint of either exists offline and neither fairseq nor the reference checkout is on the GPU box, so `bench.py --e2e` times
randomly initialised networks of the SAME ARCHITECTURE ("architecture proxies, timed separately and reported") on
stock PyTorch-ROCm kernels.  Their outputs are meaningless as features / pitch -- only their cost, shapes and call protocol are real.
Nothing here is product code and nothing is checked for parity.

* `HubertProxy`: `transformers.HubertModel(HubertConfig())` = HuBERT-base (7 conv layers 512 ch, kernels 10,3,3,3,3,2,2, strides
  5,2,2,2,2,2,2; 12 x 768-d transformer layers), called like the fairseq model in infer/modules/vc/pipeline.py:103-110.
* `RmvpeProxy`: the network of rvc/f0/e2e.py:9-48 (`E2E(4, 1, (2, 2))`, rvc/f0/models.py) restated layer for layer -- the deep U-Net of
  rvc/f0/deepunet.py (5 encoder levels of 4 residual units, 16 -> 256 channels, 2x2 average pools; 4 x 4 residual units at 512 channels;
  5 decoder levels of a stride-2 transposed conv + skip concat + 4 residual units), a 3-channel 3x3 head, a bidirectional GRU(384 -> 256)
  and Linear(512 -> 360) + sigmoid -- behind the mel front end of rvc/f0/mel.py:39-71 (n_fft 1024, hop 160, 128 mel bins, log clamp 1e-5)
  and the pad-to-32 / crop of rvc/f0/rmvpe.py:144-163.  Exposes what `rvc_amd.pipeline._rmvpe_on_device` uses of an `RMVPE` instance.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class HubertProxy:
    def __init__(self, device, half=True, seed=0):
        from transformers import HubertConfig, HubertModel

        torch.manual_seed(seed)
        self.m = HubertModel(HubertConfig()).eval().to(device)
        self.proj = nn.Linear(768, 256).to(device)  # final_proj, v1 models only
        if half:
            self.m, self.proj = self.m.half(), self.proj.half()
        self.calls = 0

    def extract_features(self, source, padding_mask, output_layer):
        self.calls += 1
        with torch.no_grad():
            return (self.m(source.to(next(self.m.parameters()).dtype)).last_hidden_state,)

    def final_proj(self, x):
        return self.proj(x)


def _unit(cin, cout):
    """two 3x3 convs (no bias) each followed by BatchNorm + ReLU; identity or 1x1 shortcut (deepunet.py:7-46)"""
    body = nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(),
                         nn.Conv2d(cout, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU())
    return nn.ModuleDict({"body": body, **({"short": nn.Conv2d(cin, cout, 1)} if cin != cout else {})})


def _run_units(units, x):
    for u in units:
        x = u["body"](x) + (u["short"](x) if "short" in u else x)
    return x


class _SalienceNet(nn.Module):
    def __init__(self, levels=5, blocks=4, inter=4, c0=16):
        super().__init__()
        self.bn = nn.BatchNorm2d(1)
        self.enc, self.dec_up, self.dec = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        cin, c = 1, c0
        for _ in range(levels):  # deepunet.py:49-88, 91-124
            self.enc.append(nn.ModuleList([_unit(cin if b == 0 else c, c) for b in range(blocks)]))
            cin, c = c, 2 * c
        self.mid = nn.ModuleList([_unit(cin if (i == 0 and b == 0) else c, c) for i in range(inter) for b in range(blocks)])  # :127-145
        for _ in range(levels):  # deepunet.py:148-196
            self.dec_up.append(nn.Sequential(nn.ConvTranspose2d(c, c // 2, 3, stride=2, padding=1, output_padding=1, bias=False),
                                             nn.BatchNorm2d(c // 2), nn.ReLU()))
            self.dec.append(nn.ModuleList([_unit(c if b == 0 else c // 2, c // 2) for b in range(blocks)]))
            c //= 2
        self.head = nn.Conv2d(c, 3, 3, padding=1)                                   # e2e.py:29
        self.gru = nn.GRU(3 * 128, 256, num_layers=1, batch_first=True, bidirectional=True)  # e2e.py:31-35, 50-67
        self.out = nn.Linear(512, 360)

    def forward(self, mel):  # [B, 128, T], T a multiple of 32 -> [B, T, 360]
        x = self.bn(mel.transpose(-1, -2).unsqueeze(1))
        skips = []
        for lvl in self.enc:
            x = _run_units(lvl, x)
            skips.append(x)
            x = F.avg_pool2d(x, 2)
        x = _run_units(self.mid, x)
        for up, lvl in zip(self.dec_up, self.dec):
            x = _run_units(lvl, torch.cat((up(x), skips.pop()), dim=1))
        x = self.head(x).transpose(1, 2).flatten(-2)
        return torch.sigmoid(self.out(self.gru(x)[0]))


class RmvpeProxy:
    def __init__(self, device, half=True, seed=0):
        torch.manual_seed(seed)
        self.device, self.is_half = device, half
        self.model = _SalienceNet().eval().to(device)
        if half:
            self.model = self.model.half()
        self.mel_basis = (torch.rand(128, 513, device=device) / 513.0)  # (librosa's filter bank is not installable; same shape and cost)
        self.window = torch.hann_window(1024, device=device)

    def mel_extractor(self, audio, center=True):  # [1, n] -> log-mel [1, 128, n // 160 + 1]     mel.py:39-71
        mag = torch.stft(audio.float(), 1024, hop_length=160, win_length=1024, window=self.window, center=center, return_complex=True).abs()
        mel = torch.matmul(self.mel_basis, mag)
        return torch.log(torch.clamp(mel.half() if self.is_half else mel, min=1e-5))

    def _mel2hidden(self, mel):  # rmvpe.py:144-163
        with torch.no_grad():
            n = mel.shape[-1]
            pad = 32 * ((n - 1) // 32 + 1) - n
            if pad > 0:
                mel = F.pad(mel, (0, pad))
            return self.model(mel.half() if self.is_half else mel.float())[:, :n]
