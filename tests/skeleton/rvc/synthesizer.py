"""Loader shell with the reference's two entry points (same names, arguments and return values)."""
import torch

from .layers.synthesizers import SynthesizerTrnMsNSFsid


def get_synthesizer(cpt, device=torch.device("cpu")):
    cpt["config"][-3] = cpt["weight"]["emb_g.weight"].shape[0]
    net_g = SynthesizerTrnMsNSFsid(*cpt["config"], encoder_dim=768 if cpt.get("version", "v1") == "v2" else 256,
                                   use_f0=cpt.get("f0", 1) == 1)
    del net_g.enc_q
    missing = net_g.load_state_dict(cpt["weight"], strict=False)
    assert not missing.missing_keys, missing.missing_keys[:3]
    net_g = net_g.float().eval().to(device)
    net_g.remove_weight_norm()
    return net_g, cpt


def load_synthesizer(pth_path, device=torch.device("cpu")):
    return get_synthesizer(torch.load(pth_path, map_location=torch.device("cpu"), weights_only=True), device)
