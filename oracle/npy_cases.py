"""Scripted write sequences for the streaming `.npy` writer goldens.  TEST INFRASTRUCTURE."""
import numpy as np


def _g(seed):
    return np.random.Generator(np.random.Philox(key=[seed, 77]))


def _bert_like(w):            # preprocess/bert.py style: extend() with a batch of [L, D] rows per call
    g = _g(1)
    for _ in range(3):
        w.extend(g.standard_normal(size=(4, 6, 8), dtype=np.float32))


def _resnet_regions(w):       # [P, R] items appended one by one, then re-declared as [n, P*R]
    g = _g(2)
    for _ in range(5):
        w.append(g.standard_normal(size=(7, 16), dtype=np.float32))
    w.reshape((5, -1))


def _topk_objects(w):         # items written flat, re-declared [-1, K, 1, R] (leading -1)
    g = _g(3)
    for _ in range(12):
        w.append(g.standard_normal(size=(16,)).astype(np.float16))
    w.reshape((-1, 3, 1, 16))


def _similarity_scalars(w):   # 0-d items -> 1-D file
    g = _g(4)
    for v in g.standard_normal(size=9):
        w.append(np.array(v, dtype=np.float64))


def _positions(w):            # int64 items of shape [2]
    g = _g(5)
    w.extend(g.integers(0, 128, size=(10, 2)))


def _answers_u8(w):
    g = _g(6)
    w.extend(g.integers(0, 2, size=(6, 10)).astype(np.uint8))
    w.reshape((3, 2, 10))


def _big_endian_free(w):      # uint16 / int8 use the '<u2' / '|i1' descr spellings
    w.append(np.arange(5, dtype=np.uint16))
    w.append(np.arange(5, 10, dtype=np.uint16))


def _int8(w):
    w.append(np.arange(-3, 3, dtype=np.int8).reshape(2, 3))


NPY_CASES = {
    "bert_like": _bert_like,
    "resnet_regions": _resnet_regions,
    "topk_objects": _topk_objects,
    "similarity_scalars": _similarity_scalars,
    "positions": _positions,
    "answers_u8": _answers_u8,
    "uint16": _big_endian_free,
    "int8": _int8,
}


def replay(writer_cls, path, name):
    w = writer_cls(path)
    NPY_CASES[name](w)
    w.close()
