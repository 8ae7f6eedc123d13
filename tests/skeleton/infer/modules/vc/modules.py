"""Takes the loader names the way the reference's VC class does (bound at import time)."""
from rvc.synthesizer import get_synthesizer, load_synthesizer


class VC:
    def __init__(self, device, is_half=False):
        self.device, self.is_half, self.net_g, self.cpt = device, is_half, None, None

    def get_vc(self, pth_path):
        self.net_g, self.cpt = load_synthesizer(pth_path, self.device)
        self.net_g = self.net_g.half() if self.is_half else self.net_g.float()
        return self.net_g
