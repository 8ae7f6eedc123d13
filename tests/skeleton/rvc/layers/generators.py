from torch import nn

from ._shell import NoCompute, mlist


class _ResBlockShell(NoCompute):
    def __init__(self, ch, k, dils):
        super().__init__()
        self.convs1 = mlist(nn.Conv1d(ch, ch, k, dilation=d) for d in dils)
        self.convs2 = mlist(nn.Conv1d(ch, ch, k) for _ in dils)


class Generator(NoCompute):
    """no-f0 decoder shell: conv_pre, ups, resblocks, conv_post, cond."""

    def __init__(self, initial_channel, resblock, resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                 upsample_initial_channel, upsample_kernel_sizes, gin_channels=0):
        super().__init__()
        self.num_kernels, self.num_upsamples = len(resblock_kernel_sizes), len(upsample_rates)
        c0 = upsample_initial_channel
        self.conv_pre = nn.Conv1d(initial_channel, c0, 7)
        self.ups = mlist(nn.ConvTranspose1d(c0 >> i, c0 >> (i + 1), k, stride=u) for i, (u, k) in enumerate(zip(upsample_rates, upsample_kernel_sizes)))
        self.resblocks = mlist(_ResBlockShell(c0 >> (i + 1), k, d) for i in range(self.num_upsamples)
                               for k, d in zip(resblock_kernel_sizes, resblock_dilation_sizes))
        self.conv_post = nn.Conv1d(c0 >> self.num_upsamples, 1, 7, bias=False)
        if gin_channels:
            self.cond = nn.Conv1d(gin_channels, c0, 1)

    def remove_weight_norm(self):
        return None
