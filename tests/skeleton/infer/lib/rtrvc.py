import faiss
from rvc.synthesizer import load_synthesizer


class RVC:
    def __init__(self, pth_path, index_path, device):
        self.index = faiss.read_index(index_path)
        self.big_npy = self.index.reconstruct_n(0, self.index.ntotal)
        self.net_g, self.cpt = load_synthesizer(pth_path, device)
