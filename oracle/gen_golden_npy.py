"""Golden files of the reference's streaming `.npy` writer (`common/utils.py:103-220`).  TEST INFRASTRUCTURE.

`common/utils.py` cannot be imported here (`torchmetrics` missing - an ordinary ModuleNotFoundError), and the
class is self-contained numpy/ctypes code, so - as for `TripletLoss` in `gen_golden.py` - its class body is
taken from that file with `ast` and executed.  Stored: the complete bytes of each file it writes for the
scripted sequences of `NPY_CASES` (shared with `tests/test_npy_stream.py`, which replays them through
`drin_amd.npy_stream.NpyWriter`).  Run in the build container only: `python oracle/gen_golden_npy.py`.
"""
import ast
import os
import sys
import tempfile
from ctypes import c_uint8

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF = "/root/reference"

from oracle.npy_cases import NPY_CASES, replay  # noqa: E402


def _reference_writer():
    tree = ast.parse(open(os.path.join(REF, "common/utils.py")).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "NpyWriter")
    ns = {"np": np, "c_uint8": c_uint8}
    exec(compile(ast.Module(body=[cls], type_ignores=[]), "common/utils.py", "exec"), ns)
    return ns["NpyWriter"]


def main():
    Ref = _reference_writer()
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for name in NPY_CASES:
            path = os.path.join(d, name + ".npy")
            replay(Ref, path, name)
            out[name] = np.frombuffer(open(path, "rb").read(), dtype=np.uint8)
            try:
                print(name, out[name].size, "bytes ->", np.load(path).shape)
            except ValueError:
                # reshape(-1) leaves a numpy scalar in the shape (utils.py:193); under numpy >= 2 its repr is
                # "np.int64(112)" and the reference's own file is unreadable.  Stored as written; the test
                # compares against the numpy-1.x spelling the reference's pinned numpy 1.24 produced ("112").
                print(name, out[name].size, "bytes -> header not loadable under numpy 2:", bytes(out[name][10:128]).strip())
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "npy_writer.npz"), **out)


if __name__ == "__main__":
    main()
