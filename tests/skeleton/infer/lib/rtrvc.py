import faiss
from rvc.synthesizer import load_synthesizer


class RVC:
    def __init__(self, pth_path, index_path, device):
        self.index = faiss.read_index(index_path)
        self.big_npy = self.index.reconstruct_n(0, self.index.ntotal)
        self.net_g, self.cpt = load_synthesizer(pth_path, device)

    def infer(self, input_wav, block_frame_16k, skip_head, return_length, f0method, protect=1.0):
        raise NotImplementedError("skeleton: no compute")
