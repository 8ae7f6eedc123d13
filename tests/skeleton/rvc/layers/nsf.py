import math

from torch import nn

from ._shell import NoCompute, mlist
from .generators import Generator


class _SineShell(NoCompute):
    def __init__(self, sr):
        super().__init__()
        self.sampling_rate = sr


class _SourceShell(NoCompute):
    def __init__(self, sr):
        super().__init__()
        self.l_sin_gen = _SineShell(sr)
        self.l_linear = nn.Linear(1, 1)


class NSFGenerator(Generator):
    """f0 decoder shell: Generator's layers + m_source + noise_convs.  Like the reference it narrows __call__'s signature,
    which is what the drop-in's dynamic subclass has to get around."""

    def __init__(self, initial_channel, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                 upsample_initial_channel, upsample_kernel_sizes, gin_channels, sr):
        super().__init__(initial_channel, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                         upsample_initial_channel, upsample_kernel_sizes, gin_channels)
        self.upp = math.prod(upsample_rates)
        self.m_source = _SourceShell(sr)
        c0, n = upsample_initial_channel, len(upsample_rates)
        convs = []
        for i in range(n):
            cout = c0 >> (i + 1)
            if i + 1 < n:
                s = math.prod(upsample_rates[i + 1:])
                convs.append(nn.Conv1d(1, cout, 2 * s, stride=s, padding=s // 2))
            else:
                convs.append(nn.Conv1d(1, cout, 1))
        self.noise_convs = mlist(convs)

    def __call__(self, x, f0, g=None, n_res=None):
        return super().__call__(x, f0, g=g, n_res=n_res)
