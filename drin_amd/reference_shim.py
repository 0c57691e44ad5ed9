"""The two modules the reference's driver selects for `model_type = "drin"` (`train.py:13-14`:
`from drin import data as data_module, model as model_module`), backed by this library.

In a checkout of the reference the whole binding is one changed line of `train.py`:

    from drin_amd import reference_shim as data_module, reference_shim as model_module      # was: from drin import ...

`model_module.Model()` (`train.py:136`; `drin/model.py:157-162`) takes no argument and reads the star-imported globals
of `common.args`; `data_module.create_datasets()` (`train.py:135`; `drin/data.py:158-200`) likewise.  Both are rebuilt
here from the SAME module (`common.args`, imported at call time, so values patched before the call are seen - the
reference freezes them at import).  Nothing else of the reference is imported.
"""
from __future__ import annotations

import importlib
import os

from .config import config_from_reference_args
from .data import create_datasets as _create_datasets
from .model import Model as _Model


def _args():
    return importlib.import_module("common.args")


class Model(_Model):
    """`model_module.Model()`: geometry, layer count, edge switches and activations from `common.args`
    (`args.py:24-40,45,52-57,72,77,83-101`).  Contractions run split-bf16 (`precision="bf16x3"`: <= 1.4e-6 on the scores
    against the reference's fp32 forward, the 1e-4 bar of the path); `DRIN_PRECISION=f32` in the environment selects the
    exact fp32 MFMA kernels."""

    def __init__(self):
        super().__init__(config_from_reference_args(_args()), precision=os.environ.get("DRIN_PRECISION", "bf16x3"))


def create_datasets():
    """`data_module.create_datasets()`: [train, valid, test] loaders over `args.preprocess_dir` with `args.batch_size`,
    `args.dataloader_workers`, `args.mention_mmap` / `args.entity_mmap` (`args.py:73-74,79,105,118,126`)."""
    a = _args()
    return _create_datasets(config_from_reference_args(a), a.preprocess_dir, a.batch_size, getattr(a, "dataloader_workers", 0),
                            mention_mmap=getattr(a, "mention_mmap", None), entity_mmap=getattr(a, "entity_mmap", None))
