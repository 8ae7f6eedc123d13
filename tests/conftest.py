import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _cpu_quota():
    """CPUs the cgroup lets this process use, None when unlimited / unknown (cgroup v2 cpu.max, v1 cfs quota)."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else max(1, int(q) // int(per))
    except Exception:  # noqa
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        return None if q <= 0 else max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()))
    except Exception:  # noqa
        return None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # torch sizes its CPU pool by the HOST's cores (128 threads on the GPU box, whose container has a 16-CPU CFS quota): the oracle legs then spin
    # the quota away and the whole process is throttled for most of every 100 ms period.  The results do not depend on it.
    quota = _cpu_quota()
    if quota and torch.get_num_threads() > quota:
        torch.set_num_threads(quota)


def golden_names(prefix="dec_"):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))


def load_golden(name):
    """A fixture written by oracle/make_golden.py from the REAL reference modules."""
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    for k in ("cfg_name", "weights_sha256"):
        if k in d:
            d[k] = str(d[k])
    return d


def golden_config_and_weights(d):
    """Regenerate the seeded weights of a golden case and verify they are the ones the fixture was made with."""
    from oracle import synth
    from oracle.nsf_oracle import CONFIGS, GenConfig

    cfg = CONFIGS[d["cfg_name"]]
    if "use_f0" in d and not bool(d["use_f0"]):
        cfg = GenConfig(**{**vars(cfg), "use_f0": False})
    w = synth.make_dec_weights(cfg, int(d["seed"]))
    assert synth.weights_sha256(w) == d["weights_sha256"], \
        "seeded weights differ from the ones the golden fixture was generated with (torch RNG changed?)"
    return cfg, w


def rms(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).pow(2).mean().sqrt())


@pytest.fixture(scope="session")
def gpu():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")
