"""The faiss index object as the reference uses it, backed by the HIP IVF-Flat kernels.

Mirrors exactly the surface the reference touches:

    index = faiss.read_index(path)                 -> read_index(path)            pipeline.py:214
    big_npy = index.reconstruct_n(0, index.ntotal) -> index.reconstruct_n(0, n)   pipeline.py:215
    score, ix = index.search(npy, k=8)             -> index.search(npy, k=8)      pipeline.py:126
    faiss.extract_index_ivf(index).nprobe = 1      -> index.nprobe = 1            web.py:551-552
    faiss.write_index(index, path)                 -> write_index(index, path)    web.py:571

plus the device-resident fast path ``search_blend`` that fuses pipeline.py:126-138 and removes the
two D2H + two H2D hops around retrieval.  numpy in -> numpy out (drop-in); torch.cuda in -> torch.cuda out.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple, Union

import numpy as np
import torch

from . import _lib

ArrayLike = Union[np.ndarray, torch.Tensor]


class IVFFlatHIP:
    def __init__(self, handle: C.c_void_p, device: torch.device, keepalive=None):
        self._h = handle
        self.device = device
        self._keepalive = keepalive  # e.g. the torch tensor that owns a broadcast blob

    # -- construction ----------------------------------------------------------------------------
    @classmethod
    def from_arrays(cls, centroids: np.ndarray, list_offsets: np.ndarray, ids: np.ndarray, vecs: np.ndarray,
                    nprobe: int = 1, device="cuda:0") -> "IVFFlatHIP":
        """What ``index.train(); index.add()`` leave behind (web.py:553-563): centroids [nlist,d],
        list_offsets [nlist+1], and ids [n] / vecs [n,d] in list-major order."""
        dev = _cuda(device)
        cent = np.ascontiguousarray(centroids, dtype=np.float32)
        off = np.ascontiguousarray(list_offsets, dtype=np.int64)
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        vecs = np.ascontiguousarray(vecs, dtype=np.float32)
        nlist, d = cent.shape
        n = ids.shape[0]
        if off.shape != (nlist + 1,) or vecs.shape != (n, d):
            raise ValueError("inconsistent IVF arrays")
        h = C.c_void_p(None)
        _lib.check(_lib.lib().rvcmi_ivf_create(d, n, nlist, int(nprobe), cent.ctypes.data_as(C.c_void_p),
                                               off.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p),
                                               vecs.ctypes.data_as(C.c_void_p), _idx(dev), C.byref(h)))
        return cls(h, dev)

    @classmethod
    def train(cls, big_npy: np.ndarray, nlist: int = None, niter: int = 10, seed: int = 1234, device="cuda:0",
              return_objective: bool = False):
        """``index = faiss.index_factory(d, "IVF%s,Flat" % n_ivf); index.train(big_npy); index.add(big_npy)`` (web.py:544-563)
        on the GPU.  ``nlist`` defaults to the reference's ``min(int(16 * sqrt(N)), N // 39)``; ``nprobe`` is 1 (web.py:552).
        ids are the row numbers of ``big_npy`` (sequential ``add``)."""
        dev = _cuda(device)
        x = np.ascontiguousarray(big_npy, dtype=np.float32)
        if x.ndim != 2:
            raise ValueError("big_npy must be [N, d]")
        n, d = x.shape
        if nlist is None:
            nlist = max(1, min(int(16 * np.sqrt(n)), n // 39))
        obj = np.zeros(int(niter) + 1, dtype=np.float64)
        h = C.c_void_p(None)
        _lib.check(_lib.lib().rvcmi_ivf_build(d, n, x.ctypes.data_as(C.c_void_p), int(nlist), int(niter), int(seed), _idx(dev),
                                              obj.ctypes.data_as(C.c_void_p) if return_objective else C.c_void_p(None), C.byref(h)))
        idx = cls(h, dev)
        return (idx, obj) if return_objective else idx

    @classmethod
    def from_blob(cls, blob: torch.Tensor) -> "IVFFlatHIP":
        """Adopt a device blob (uint8 CUDA tensor), e.g. one received by an RCCL broadcast."""
        if blob.dtype != torch.uint8 or blob.device.type != "cuda" or not blob.is_contiguous():
            raise ValueError("blob must be a contiguous uint8 CUDA tensor")
        h = C.c_void_p(None)
        _lib.check(_lib.lib().rvcmi_ivf_create_from_blob(C.c_void_p(blob.data_ptr()), blob.numel(), _idx(blob.device), 0,
                                                         C.byref(h)))
        return cls(h, blob.device, keepalive=blob)

    def blob(self) -> torch.Tensor:
        """A uint8 CUDA tensor holding a COPY of the whole index (header+centroids+lists)."""
        p, n = C.c_void_p(None), C.c_size_t(0)
        _lib.check(_lib.lib().rvcmi_ivf_blob(self._h, C.byref(p), C.byref(n)))
        out = torch.empty(n.value, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            st = torch.cuda.current_stream().cuda_stream
            _lib.check(_lib.lib().rvcmi_ivf_blob_copy(self._h, C.c_void_p(out.data_ptr()), n.value, C.c_void_p(st)))
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().rvcmi_ivf_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # -- faiss attribute surface -------------------------------------------------------------------
    @property
    def ntotal(self) -> int:
        return int(_lib.lib().rvcmi_ivf_ntotal(self._h))

    @property
    def d(self) -> int:
        return int(_lib.lib().rvcmi_ivf_d(self._h))

    @property
    def nlist(self) -> int:
        return int(_lib.lib().rvcmi_ivf_nlist(self._h))

    @property
    def nprobe(self) -> int:
        return int(_lib.lib().rvcmi_ivf_nprobe(self._h))

    @nprobe.setter
    def nprobe(self, v: int) -> None:
        _lib.check(_lib.lib().rvcmi_ivf_set_nprobe(self._h, int(v)))

    def reserve(self, max_nq: int) -> "IVFFlatHIP":
        _lib.check(_lib.lib().rvcmi_ivf_reserve(self._h, int(max_nq)))
        return self

    def centroids(self) -> np.ndarray:
        """The coarse quantizer's centres [nlist, d] (``faiss.extract_index_ivf(index).quantizer.reconstruct_n(0, nlist)``)."""
        out = np.empty((self.nlist, self.d), dtype=np.float32)
        _lib.check(_lib.lib().rvcmi_ivf_centroids(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def reconstruct_n(self, i0: int = 0, n: int = None) -> np.ndarray:
        n = self.ntotal - i0 if n is None else int(n)
        out = np.empty((n, self.d), dtype=np.float32)
        _lib.check(_lib.lib().rvcmi_ivf_reconstruct_n(self._h, int(i0), n, out.ctypes.data_as(C.c_void_p)))
        return out

    def search(self, x: ArrayLike, k: int = 8) -> Tuple[ArrayLike, ArrayLike]:
        """``index.search(x, k)`` -> (D squared-L2 ascending, I int64, -1/FLT_MAX padded)."""
        is_np = isinstance(x, np.ndarray)
        if is_np:
            if x.dtype != np.float32:
                raise TypeError("queries must be float32 (faiss raises on anything else)")
            xt = torch.from_numpy(np.ascontiguousarray(x)).to(self.device)
        else:
            xt = x.to(self.device, torch.float32).contiguous()
        if xt.dim() != 2 or xt.shape[1] != self.d:
            raise ValueError("queries must be [nq, %d], got %s" % (self.d, tuple(xt.shape)))  # faiss: assert d == self.d
        nq = xt.shape[0]
        D = torch.empty(nq, k, dtype=torch.float32, device=self.device)
        I = torch.empty(nq, k, dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            st = torch.cuda.current_stream().cuda_stream
            _lib.check(_lib.lib().rvcmi_ivf_search(self._h, nq, C.c_void_p(xt.data_ptr()), int(k), C.c_void_p(D.data_ptr()),
                                                   C.c_void_p(I.data_ptr()), C.c_void_p(st)))
        if is_np:
            return D.cpu().numpy(), I.cpu().numpy()
        return D, I

    def search_blend(self, feats: torch.Tensor, index_rate: float, k: int = 8, skip_if_short: bool = False) -> torch.Tensor:
        """pipeline.py:126-138 fused on the device; ``feats`` [nq,d] fp32 CUDA is updated IN PLACE and returned.
        ``skip_if_short`` reproduces the realtime guard ``if (ix >= 0).all()`` of rtrvc.py:173."""
        if feats.device.type != "cuda" or feats.dtype != torch.float32 or not feats.is_contiguous():
            raise ValueError("feats must be a contiguous float32 CUDA tensor")
        if feats.dim() != 2 or feats.shape[1] != self.d:
            raise ValueError("feats must be [nq, %d]" % self.d)
        with torch.cuda.device(self.device):
            st = torch.cuda.current_stream().cuda_stream
            _lib.check(_lib.lib().rvcmi_ivf_search_blend(self._h, feats.shape[0], C.c_void_p(feats.data_ptr()),
                                                         float(index_rate), int(k), 1 if skip_if_short else 0, C.c_void_p(st)))
        return feats

    def set_option(self, key: str, value=None) -> None:
        """Dev / test option of this index (``rvcmi_ivf_set_option``: ``IVF_COARSE_F64``, ``IVF_GENERIC``, ``IVF_LM`` 0 = never the list-major
        kernels, ``IVF_LM_MIN`` their routing threshold in queries (default 16), ``IVF_SORT``); ``None`` = default; an unknown key is an error."""
        _lib.set_option(_lib.lib().rvcmi_ivf_set_option, self._h, key, value)

    def profile(self, enable: bool) -> None:
        _lib.check(_lib.lib().rvcmi_ivf_profile_enable(self._h, 1 if enable else 0))

    def profile_read(self, reset: bool = True) -> List[dict]:
        return _lib.read_stats(_lib.lib().rvcmi_ivf_profile_read, self._h, reset)


def read_index(path: str, device="cuda:0") -> IVFFlatHIP:
    """``faiss.read_index(path)`` for IVF-Flat/L2 files (IwFl)."""
    dev = _cuda(device)
    h = C.c_void_p(None)
    _lib.check(_lib.lib().rvcmi_ivf_create_from_file(str(path).encode(), _idx(dev), C.byref(h)))
    return IVFFlatHIP(h, dev)


def write_index(index: IVFFlatHIP, path: str) -> None:
    """``faiss.write_index(index, path)``."""
    _lib.check(_lib.lib().rvcmi_ivf_write_file(index._h, str(path).encode()))


def _cuda(device) -> torch.device:
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.RvcmiError("the HIP IVF index needs a GPU device (got %s); there is no CPU fallback" % device)
    return dev


def _idx(dev: torch.device) -> int:
    return dev.index if dev.index is not None else torch.cuda.current_device()


def reduce_features(big_npy: np.ndarray, n_clusters: int = 10000, threshold: float = 2e5, niter: int = 10, seed: int = 1234,
                    device="cuda:0") -> np.ndarray:
    """The training-set reduction of web.py:522-536: more than 2e5 feature rows are replaced by ``n_clusters`` k-means
    centres before the index is trained.  The reference runs sklearn's ``MiniBatchKMeans(init="random")`` on the host
    (stochastic mini-batches, its own RNG: not reproducible elsewhere); this runs full-batch Lloyd iterations on the GPU with
    the exact coarse-assignment kernels of the search path (same objective, lower final inertia).  Smaller sets are returned
    unchanged, like the reference's ``if big_npy.shape[0] > 2e5``."""
    x = np.ascontiguousarray(big_npy, dtype=np.float32)
    if x.shape[0] <= threshold:
        return x
    return kmeans(x, int(n_clusters), niter=niter, seed=seed, device=device)


def kmeans(x: np.ndarray, k: int, niter: int = 10, seed: int = 1234, device="cuda:0") -> np.ndarray:
    """``k`` k-means centres of the rows of ``x`` (``rvcmi_kmeans``): the Lloyd iterations of :meth:`IVFFlatHIP.train` without the
    index -- no ``add`` pass over the N rows, no list-major copy in HBM.  A cluster that loses all its points is re-seeded by
    splitting the largest one, so all ``k`` rows returned are valid centres."""
    dev = _cuda(device)
    x = np.ascontiguousarray(x, dtype=np.float32)
    if x.ndim != 2:
        raise ValueError("x must be [N, d]")
    n, d = x.shape
    out = np.empty((int(k), d), dtype=np.float32)
    _lib.check(_lib.lib().rvcmi_kmeans(d, n, x.ctypes.data_as(C.c_void_p), int(k), int(niter), int(seed), _idx(dev), C.c_void_p(None),
                                       out.ctypes.data_as(C.c_void_p)))
    return out


def train_index(big_npy: np.ndarray, path: str = None, nlist: int = None, niter: int = 10, seed: int = 1234, device="cuda:0") -> IVFFlatHIP:
    """The index recipe of web.py:544-571 (``train`` + ``add`` + optional ``write_index``) on the GPU."""
    idx = IVFFlatHIP.train(big_npy, nlist=nlist, niter=niter, seed=seed, device=device)
    if path is not None:
        write_index(idx, path)
    return idx
