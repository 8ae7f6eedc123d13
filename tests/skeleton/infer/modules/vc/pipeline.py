"""The retrieval call pattern of the conversion pipeline: module-level `import faiss`, read_index + reconstruct_n once,
search(k=8) + inverse-square weights + blend per chunk."""
import faiss
import numpy as np


def load_index(file_index):
    index = faiss.read_index(file_index)
    return index, index.reconstruct_n(0, index.ntotal)


def blend(index, big_npy, npy, index_rate):
    score, ix = index.search(npy, k=8)
    weight = np.square(1 / score)
    weight /= weight.sum(axis=1, keepdims=True)
    got = np.sum(big_npy[ix] * np.expand_dims(weight, axis=2), axis=1)
    return got * index_rate + (1 - index_rate) * npy, ix
