"""The retrieval call pattern of the conversion pipeline: module-level `import faiss`, read_index + reconstruct_n once,
search(k=8) + inverse-square weights + blend per chunk."""
import faiss
import numpy as np
from scipy import signal

bh, ah = signal.butter(N=5, Wn=48, btype="high", fs=16000)  # the module-level high-pass the conversion entry applies


def load_index(file_index):
    index = faiss.read_index(file_index)
    return index, index.reconstruct_n(0, index.ntotal)


def blend(index, big_npy, npy, index_rate):
    score, ix = index.search(npy, k=8)
    weight = np.square(1 / score)
    weight /= weight.sum(axis=1, keepdims=True)
    got = np.sum(big_npy[ix] * np.expand_dims(weight, axis=2), axis=1)
    return got * index_rate + (1 - index_rate) * npy, ix


class Pipeline:
    """The state of the conversion object (window / padding geometry from the config) and its two entry points.  Compute-free:
    both raise, so whatever converts audio through this class in the tests is the HIP path bound over them."""

    def __init__(self, tgt_sr, config):
        self.x_pad, self.x_query, self.x_center, self.x_max, self.is_half = (config.x_pad, config.x_query, config.x_center,
                                                                             config.x_max, config.is_half)
        self.sr, self.window = 16000, 160
        self.t_pad = self.sr * self.x_pad
        self.t_pad_tgt = tgt_sr * self.x_pad
        self.t_pad2 = self.t_pad * 2
        self.t_query = self.sr * self.x_query
        self.t_center = self.sr * self.x_center
        self.t_max = self.sr * self.x_max
        self.device = config.device
        self.f0_gen = None

    def vc(self, model, net_g, sid, audio0, pitch, pitchf, times, index, big_npy, index_rate, version, protect):
        raise NotImplementedError("skeleton: no compute")

    def pipeline(self, model, net_g, sid, audio, times, f0_up_key, f0_method, file_index, index_rate, if_f0, filter_radius, tgt_sr,
                 resample_sr, rms_mix_rate, version, protect, f0_file=None):
        raise NotImplementedError("skeleton: no compute")
