"""MI355X-native hot path of RVC inference (imported as ``rvc_amd``).

faiss IVF-Flat retrieval + NSF-HiFi-GAN generator as hand-written HIP for gfx950 behind the
reference's own Python call surface.  See DESIGN.md / INTEGRATION.md.
"""
from . import _lib
from ._lib import RvcmiError, build
from .ivf import IVFFlatHIP, kmeans, read_index, reduce_features, train_index, write_index
from .front import FrontHIP, front_config_from_reference, infer_hip
from .nsf import GeneratorHIP, NSFGeneratorHIP, config_from_reference
from .pipeline import retrieve_blend
from . import glue
from .gru import GRUHIP, accelerate_rmvpe
from .synthesizer import accelerate_synthesizer, get_synthesizer, load_synthesizer
from . import dist
from .install import install, uninstall
from .realtime import PitchCache, RealtimeVC, SincResample, f0_extractor_frame, sinc_resample_kernel

__all__ = [
    "RvcmiError", "build", "IVFFlatHIP", "read_index", "write_index", "train_index", "reduce_features", "kmeans", "GeneratorHIP", "NSFGeneratorHIP",
    "config_from_reference", "FrontHIP", "front_config_from_reference", "infer_hip", "retrieve_blend", "accelerate_synthesizer", "get_synthesizer", "load_synthesizer", "dist", "glue", "install", "uninstall", "RealtimeVC", "PitchCache", "f0_extractor_frame", "SincResample", "sinc_resample_kernel", "GRUHIP", "accelerate_rmvpe",
]
