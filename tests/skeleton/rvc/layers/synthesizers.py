from torch import nn

from ._shell import NoCompute, mlist
from .generators import Generator
from .nsf import NSFGenerator


class _Attn(NoCompute):
    def __init__(self, c, heads, window):
        super().__init__()
        import torch

        self.window_size = window
        for n in ("conv_q", "conv_k", "conv_v", "conv_o"):
            setattr(self, n, nn.Conv1d(c, c, 1))
        self.emb_rel_k = nn.Parameter(torch.zeros(1, 2 * window + 1, c // heads))
        self.emb_rel_v = nn.Parameter(torch.zeros(1, 2 * window + 1, c // heads))


class _FFN(NoCompute):
    def __init__(self, c, f, k):
        super().__init__()
        self.conv_1, self.conv_2 = nn.Conv1d(c, f, k), nn.Conv1d(f, c, k)


class _LN(NoCompute):
    def __init__(self, c):
        super().__init__()
        import torch

        self.gamma, self.beta = nn.Parameter(torch.ones(c)), nn.Parameter(torch.zeros(c))


class _Encoder(NoCompute):
    def __init__(self, c, f, heads, layers, k, window=10):
        super().__init__()
        self.attn_layers = mlist(_Attn(c, heads, window) for _ in range(layers))
        self.norm_layers_1 = mlist(_LN(c) for _ in range(layers))
        self.ffn_layers = mlist(_FFN(c, f, k) for _ in range(layers))
        self.norm_layers_2 = mlist(_LN(c) for _ in range(layers))


class _TextEncoder(NoCompute):
    def __init__(self, in_channels, out_channels, hidden, filt, heads, layers, k, f0):
        super().__init__()
        self.out_channels, self.hidden_channels, self.filter_channels = out_channels, hidden, filt
        self.n_heads, self.n_layers, self.kernel_size = heads, layers, k
        self.emb_phone = nn.Linear(in_channels, hidden)
        if f0:
            self.emb_pitch = nn.Embedding(256, hidden)
        self.encoder = _Encoder(hidden, filt, heads, layers, k)
        self.proj = nn.Conv1d(hidden, 2 * out_channels, 1)


class _WN(NoCompute):
    def __init__(self, hidden, k, layers, gin):
        super().__init__()
        self.in_layers = mlist(nn.Conv1d(hidden, 2 * hidden, k) for _ in range(layers))
        self.res_skip_layers = mlist(nn.Conv1d(hidden, 2 * hidden if i < layers - 1 else hidden, 1) for i in range(layers))
        if gin:
            self.cond_layer = nn.Conv1d(gin, 2 * hidden * layers, 1)


class _Coupling(NoCompute):
    def __init__(self, ch, hidden, k, layers, gin):
        super().__init__()
        self.pre, self.enc, self.post = nn.Conv1d(ch // 2, hidden, 1), _WN(hidden, k, layers, gin), nn.Conv1d(hidden, ch // 2, 1)


class _Flip(NoCompute):
    pass


class _Flow(NoCompute):
    def __init__(self, ch, hidden, k, dil, layers, n_flows, gin):
        super().__init__()
        self.n_flows, self.n_layers, self.kernel_size, self.dilation_rate, self.gin_channels = n_flows, layers, k, dil, gin
        mods = []
        for _ in range(n_flows):
            mods += [_Coupling(ch, hidden, k, layers, gin), _Flip()]
        self.flows = mlist(mods)


class SynthesizerTrnMsNSFsid(NoCompute):
    """Same positional config list as the reference's .pth `config` entry."""

    def __init__(self, spec_channels, segment_size, inter_channels, hidden_channels, filter_channels, n_heads, n_layers,
                 kernel_size, p_dropout, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                 upsample_initial_channel, upsample_kernel_sizes, spk_embed_dim, gin_channels, sr, encoder_dim=768, use_f0=True):
        super().__init__()
        self.enc_p = _TextEncoder(encoder_dim, inter_channels, hidden_channels, filter_channels, n_heads, n_layers, kernel_size, use_f0)
        if use_f0:
            self.dec = NSFGenerator(inter_channels, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                                    upsample_initial_channel, upsample_kernel_sizes, gin_channels, sr)
        else:
            self.dec = Generator(inter_channels, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                                 upsample_initial_channel, upsample_kernel_sizes, gin_channels)
        self.enc_q = NoCompute()
        self.flow = _Flow(inter_channels, hidden_channels, 5, 1, 3, 4, gin_channels)
        self.emb_g = nn.Embedding(spk_embed_dim, gin_channels)

    def remove_weight_norm(self):
        return None

    def infer(self, phone, phone_lengths, sid, pitch=None, pitchf=None, skip_head=None, return_length=None, return_length2=None):
        raise NotImplementedError("skeleton infer: the HIP path must have replaced it")
