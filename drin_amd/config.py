"""Explicit configuration object for the DRIN scoring path.

The reference keeps every hyper-parameter as a module global that is star-imported
(`common/args.py:1-137`); the names below are the same names so a reader of the
reference finds them, but they travel as one frozen dataclass instead of import-time
globals.  Only the values the hot path (`drin/model.py:156-209`) reads are kept.
"""
from __future__ import annotations

from dataclasses import dataclass, field, replace
from typing import Tuple


@dataclass(frozen=True)
class DrinConfig:
    # dataset geometry (common/args.py:76-101)
    dataset_name: str = "wikidiverse"          # "wikidiverse" | "wikimel"
    num_candidates_data: int = 10              # args.py:83,93
    # encoders (args.py:42-57)
    bert_embed_dim: int = 768                  # D  (args.py:45)
    resnet_embed_dim: int = 2048               # R  (args.py:52)
    resnet_num_region: int = 49                # P  (args.py:53)
    max_mention_sentence_len: int = 128        # L  (args.py:72)
    max_entity_attr_token_len: int = 128       # T  (args.py:85,95) - only used by wikimel layout
    object_topk_mention: int = 3               # Km (args.py:57)
    object_topk_entity: int = 1                # Ke (args.py:57)
    # model (args.py:24-40)
    gcn_embed_dim: int = 768
    num_gcn_layers: int = 2
    gcn_edge_type: str = "dynamic"             # "static" | "dynamic"
    gcn_edge_feature: str = "scaler"           # reference spelling; "vector": edges are [B, N, D] (model.py:112-116)
    gcn_edge_enabled: Tuple[float, float, float, float] = (1, 1, 1, 1)
    gcn_vertex_activation: str = "gelu"
    gcn_edge_activation: str = "sigmoid"
    # train (args.py:104-126)
    seed: int = 0
    num_epoch: int = 30
    test_epoch_interval: int = 10
    learning_rate: float = 1e-3
    triplet_margin: float = 0.25
    batch_size: int = 64
    metrics_topk: Tuple[int, ...] = (1, 3, 5)
    acc_correction: Tuple[float, float, float] = (2292 / 13205, 250 / 1552, 282 / 1570)
    shuffle_train_data: bool = True
    # numerical constants pinned in SURVEY.md §8(c)
    layer_norm_eps: float = 1e-5               # nn.LayerNorm default (drin/model.py:119)
    cosine_eps: float = 1e-8                   # nn.CosineSimilarity default (drin/model.py:57,162)
    miei_eps: float = 1e-9                     # drin/model.py:92
    clip_logit_scale: float = 100.0            # drin/model.py:203

    @property
    def num_candidates_model(self) -> int:     # args.py:101
        return self.num_candidates_data + 1

    @property
    def token_level_entities(self) -> bool:
        """WikiMEL delivers entity text as token features + mask (baselines/ghmfc.py:241-249)."""
        return self.dataset_name == "wikimel"

    def with_(self, **kw) -> "DrinConfig":
        return replace(self, **kw)

    def validate(self) -> None:
        if self.dataset_name not in ("wikidiverse", "wikimel"):
            raise ValueError(f"unknown dataset_name {self.dataset_name!r}")
        if self.gcn_edge_type not in ("static", "dynamic"):
            raise ValueError(f"unknown gcn_edge_type {self.gcn_edge_type!r}")
        if self.gcn_edge_feature not in ("scaler", "vector"):
            raise ValueError(f"unknown gcn_edge_feature {self.gcn_edge_feature!r} (reference spelling: 'scaler' | 'vector')")
        if self.gcn_edge_feature == "vector" and self.gcn_embed_dim % 8:
            raise ValueError("vector edges split gcn_embed_dim in two halves of 16-byte rows: it must be a multiple of 8")
        # the reference takes getattr(torch.nn.functional, name) (model.py:117-118): any name.  Built: see include/drin_hip.h
        if self.gcn_vertex_activation not in ("gelu", "relu", "tanh", "silu", "sigmoid"):
            raise NotImplementedError(f"gcn_vertex_activation {self.gcn_vertex_activation!r}: built are gelu, relu, tanh, silu, sigmoid")
        if self.gcn_edge_activation not in ("sigmoid", "tanh", "relu", "gelu", "silu"):
            raise NotImplementedError(f"gcn_edge_activation {self.gcn_edge_activation!r}: built are sigmoid, tanh, relu, gelu, silu")
        if self.gcn_embed_dim != self.bert_embed_dim:
            raise ValueError("gcn_embed_dim must equal bert_embed_dim (args.py:38-39 force the output dims)")
        if len(self.gcn_edge_enabled) != 4:
            raise ValueError("gcn_edge_enabled has one entry per edge type (tt, ti, it, ii)")


# names of common/args.py the scoring path and its caller read, -> DrinConfig fields of the same name
_ARGS_FIELDS = ("dataset_name", "num_candidates_data", "bert_embed_dim", "resnet_embed_dim", "resnet_num_region",
                "max_mention_sentence_len", "max_entity_attr_token_len", "gcn_embed_dim", "num_gcn_layers", "gcn_edge_type",
                "gcn_edge_feature", "gcn_vertex_activation", "gcn_edge_activation", "seed", "num_epoch", "test_epoch_interval",
                "learning_rate", "triplet_margin", "batch_size", "shuffle_train_data")


def config_from_reference_args(args=None) -> DrinConfig:
    """The `DrinConfig` a reference checkout's `common.args` module describes (`common/args.py:24-40,45,52-57,72,77,83-101,
    109-126`): what the reference's no-argument `Model()` (`drin/model.py:157-162`, `train.py:136`) reads as star-imported
    globals.  `args`: the module (default: `import common.args`).  `model_type` must be "drin" (`args.py:7`)."""
    if args is None:
        import importlib
        args = importlib.import_module("common.args")
    if getattr(args, "model_type", "drin") != "drin":
        raise ValueError(f"common.args.model_type = {args.model_type!r}: this library is the 'drin' model only (train.py:9-14)")
    kw = {k: getattr(args, k) for k in _ARGS_FIELDS if hasattr(args, k)}
    if hasattr(args, "object_topk"):                                   # args.py:57
        kw["object_topk_mention"], kw["object_topk_entity"] = args.object_topk["mention"], args.object_topk["entity"]
    for k in ("gcn_edge_enabled", "metrics_topk", "acc_correction"):   # lists there, tuples here (frozen dataclass)
        if hasattr(args, k):
            kw[k] = tuple(getattr(args, k))
    cfg = DrinConfig(**kw)
    if hasattr(args, "num_candidates_model") and args.num_candidates_model != cfg.num_candidates_model:
        raise ValueError("common.args.num_candidates_model != num_candidates_data + 1 (args.py:101)")
    cfg.validate()
    return cfg


def default_config() -> DrinConfig:
    """What `Model()` without arguments builds.  Inside the reference's driver - `train.py:2-3` has imported `common.args`
    before it builds `model_module.Model()` (`train.py:136`), and that module carries the reference's settings
    (`model_type`, `gcn_edge_type`, `num_candidates_data`) - its configuration; anywhere else the reference's WikiDiverse
    defaults.  Nothing is imported by name here (an unrelated package called `common` on `sys.path` can never change the
    model): only a module the caller's process has ALREADY loaded is read; `reference_shim.Model` is the explicit form."""
    import logging
    import sys
    args = sys.modules.get("common.args")
    if args is not None and all(hasattr(args, k) for k in ("model_type", "gcn_edge_type", "num_candidates_data")):
        logging.getLogger("drin_amd").info("Model(): configuration read from the loaded module common.args (%s)",
                                           getattr(args, "__file__", "?"))
        return config_from_reference_args(args)
    return DrinConfig()


def wikidiverse_config(**kw) -> DrinConfig:
    """Reference defaults for WikiDiverse (args.py:92-100,119-126)."""
    return DrinConfig(**kw)


def wikimel_config(**kw) -> DrinConfig:
    """Reference defaults for WikiMEL (args.py:82-91,113-118)."""
    base = dict(
        dataset_name="wikimel",
        num_candidates_data=100,
        max_entity_attr_token_len=64,
        metrics_topk=(1, 5, 10, 20, 50),
        acc_correction=(0.0, 0.0, 0.0),
    )
    base.update(kw)
    return DrinConfig(**base)
