"""The streaming `.npy` writer against files written by the reference's `NpyWriter` (`common/utils.py:103-220`,
goldens from `oracle/gen_golden_npy.py`), and the header reader / memory map the loader uses."""
import os
import re

import numpy as np
import pytest

from drin_amd.npy_stream import PREAMBLE_BYTES, NpyWriter, open_npy, read_npy_header
from oracle.npy_cases import NPY_CASES, replay


def _numpy1_spelling(raw: bytes) -> bytes:
    """numpy >= 2 prints the scalar that `reshape(-1)` leaves in the shape (utils.py:193) as `np.int64(112)`,
    which makes the reference's own file unreadable; its pinned numpy 1.24 printed `112`.  Same length
    preamble (newline padded), so only the dict literal changes."""
    head = re.sub(rb"np\.int64\((\d+)\)", rb"\1", raw[10:PREAMBLE_BYTES].rstrip(b"\n"))
    return raw[:10] + head.ljust(PREAMBLE_BYTES - 10, b"\n") + raw[PREAMBLE_BYTES:]


@pytest.mark.parametrize("name", list(NPY_CASES))
def test_writer_is_byte_identical_to_reference(tmp_path, golden_dir, name):
    golden = np.load(os.path.join(golden_dir, "npy_writer.npz"))[name].tobytes()
    path = str(tmp_path / f"{name}.npy")
    replay(NpyWriter, path, name)
    mine = open(path, "rb").read()
    assert len(mine) == len(golden)
    assert mine[PREAMBLE_BYTES:] == golden[PREAMBLE_BYTES:], "data bytes"
    assert mine == _numpy1_spelling(golden), "preamble"
    if b"np.int64" not in golden[:PREAMBLE_BYTES]:
        assert mine == golden
    # numpy reads it, the memory map agrees, and the header reader reports the same geometry
    arr = np.load(path)
    dtype, shape, off = read_npy_header(path)
    assert (dtype, shape, off) == (arr.dtype, arr.shape, PREAMBLE_BYTES)
    mm = open_npy(path)
    assert mm.shape == arr.shape and np.array_equal(np.asarray(mm), arr)


def test_writer_errors_follow_reference(tmp_path):
    """The same conditions raise the same exception TYPE as the reference's writer (utils.py:168-202); the wording is ours."""
    w = NpyWriter(str(tmp_path / "a.npy"))
    with pytest.raises(RuntimeError, match="integer or floating"):
        w.append([1, 2, 3])                                   # utils.py:168-173: ndarray only
    with pytest.raises(RuntimeError, match="integer or floating"):
        w.append(np.array(["a"]))
    w.append(np.zeros((2, 3), np.float32))
    assert w.shape == (2, 3)
    with pytest.raises(RuntimeError, match="one shape"):
        w.append(np.zeros((3, 2), np.float32))
    with pytest.raises(RuntimeError, match="one dtype"):
        w.append(np.zeros((2, 3), np.float64))
    with pytest.raises(RuntimeError, match="at most one dimension"):
        w.reshape((-1, -1))
    with pytest.raises(RuntimeError, match="6 elements have been written"):
        w.reshape((4, 5))
    w.close()
    assert np.load(str(tmp_path / "a.npy")).shape == (1, 2, 3)


def test_writer_context_manager_and_guards(tmp_path):
    p = str(tmp_path / "b.npy")
    with NpyWriter(p) as w:
        w.extend(np.arange(12, dtype=np.int32).reshape(4, 3))
    assert np.array_equal(np.load(p), np.arange(12, dtype=np.int32).reshape(4, 3))
    with pytest.raises(RuntimeError, match="nothing was appended"):
        NpyWriter(str(tmp_path / "c.npy")).close()
    # an unclosed file has no header: refused loudly rather than read as garbage
    w = NpyWriter(str(tmp_path / "d.npy"))
    w.append(np.zeros(3))
    w._fh.flush()
    with pytest.raises(ValueError, match="not an .npy file"):
        read_npy_header(str(tmp_path / "d.npy"))
    w.close()
    # a header that cannot fit the fixed 118 bytes must not overwrite data (the reference would)
    w = NpyWriter(str(tmp_path / "e.npy"))
    w.append(np.zeros((1,) * 30, np.float32))
    with pytest.raises(RuntimeError, match="does not fit"):
        w.close()
    # truncated data
    with NpyWriter(str(tmp_path / "f.npy")) as w:
        w.extend(np.zeros((4, 8), np.float32))
    with open(str(tmp_path / "f.npy"), "r+b") as f:
        f.truncate(PREAMBLE_BYTES + 100)
    with pytest.raises(ValueError, match="truncated"):
        open_npy(str(tmp_path / "f.npy"))


def test_reads_numpy_written_files(tmp_path):
    a = np.random.default_rng(0).standard_normal((5, 7)).astype(np.float32)
    np.save(str(tmp_path / "n.npy"), a)
    assert np.array_equal(np.asarray(open_npy(str(tmp_path / "n.npy"))), a)
    np.save(str(tmp_path / "f.npy"), np.asfortranarray(a))
    with pytest.raises(ValueError, match="Fortran"):
        open_npy(str(tmp_path / "f.npy"))
