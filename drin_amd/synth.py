"""Seeded synthetic batches and weights in the reference's shapes.

The reference ships no data and its real inputs (WikiMEL / WikiDiverse `.npy`) are not
available offline, so every test and the benchmark run on synthetic tensors of the shapes
`drin/data.py:110-126` hands to `Model.forward` (SURVEY.md §8a row D').  Everything is
drawn from numpy's counter-based Philox generator so that this container (where the
reference can be imported) and the GPU box regenerate bit-identical inputs from a seed.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from .config import DrinConfig

BATCH_FIELDS = (
    "mention_text_feature",
    "mention_text_mask",
    "mention_start_pos",
    "mention_end_pos",
    "mention_image_feature",
    "mention_object_feature",
    "mention_object_score",
    "entity_text_feature",
    "entity_text_mask",
    "entity_image_feature",
    "entity_object_feature",
    "entity_object_score",
    "miet_similarity",
    "mtei_similarity",
    "answer",
)  # order of drin/data.py:110-126


def _rng(seed: int, stream: int) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[seed, stream]))


def _normal(g: np.random.Generator, shape) -> np.ndarray:
    return g.standard_normal(size=shape, dtype=np.float32)


def make_batch(
    cfg: DrinConfig,
    batch: int,
    seed: int = 1,
    *,
    min_span: int = 1,
    max_span: int = 5,
    min_tokens: int = 4,
    as_torch: bool = True,
) -> List:
    """One 15-sequence batch, laid out exactly as `MELData.__getitem__` + default collate.

    Value distributions follow SURVEY.md §8(d): features ~ N(0,1), object scores ~ U(0,1),
    CLIP similarities ~ N(20,5), span start ~ U{1..} / length ~ U{min_span..max_span}
    (positions already carry the +1 CLS shift of drin/data.py:113-114), token counts
    ~ U{min_tokens..T}, answer ~ U{0..num_candidates_data} (the last value selects the
    all-zero row of drin/data.py:159-161).
    """
    B, N = batch, cfg.num_candidates_model
    D, R, L, P = cfg.bert_embed_dim, cfg.resnet_embed_dim, cfg.max_mention_sentence_len, cfg.resnet_num_region
    Km, Ke, T = cfg.object_topk_mention, cfg.object_topk_entity, cfg.max_entity_attr_token_len
    g = _rng(seed, 0)
    out: Dict[str, np.ndarray] = {}
    out["mention_text_feature"] = _normal(g, (B, L, D))
    span = g.integers(min_span, max_span + 1, size=B)
    span = np.minimum(span, L - 1)
    start = g.integers(1, np.maximum(2, np.minimum(20, L - span + 1)), size=B)
    out["mention_start_pos"] = start.astype(np.int64)
    out["mention_end_pos"] = (start + span).astype(np.int64)
    mask = (np.arange(L)[None, :] < np.minimum(L, start + span + 3)[:, None]).astype(np.int64)
    out["mention_text_mask"] = mask
    out["mention_image_feature"] = _normal(g, (B, P, R))
    out["mention_object_feature"] = _normal(g, (B, Km, 1, R))
    out["mention_object_score"] = g.random(size=(B, Km), dtype=np.float32)
    if cfg.token_level_entities:
        out["entity_text_feature"] = _normal(g, (B, N, T, D))
        ntok = g.integers(min(min_tokens, T), T + 1, size=(B, N))
        out["entity_text_mask"] = (np.arange(T)[None, None, :] < ntok[:, :, None]).astype(np.int64)
        out["entity_image_feature"] = _normal(g, (B, N, 1, R))
        out["entity_object_feature"] = _normal(g, (B, N, Ke, 1, R))
    else:
        out["entity_text_feature"] = _normal(g, (B, N, D))
        out["entity_text_mask"] = np.zeros((B,), dtype=np.int64)  # collated `0` of drin/data.py:86
        out["entity_image_feature"] = _normal(g, (B, N, R))
        out["entity_object_feature"] = _normal(g, (B, N, Ke, R))
    out["entity_object_score"] = g.random(size=(B, N, Ke), dtype=np.float32)
    out["miet_similarity"] = (20.0 + 5.0 * _normal(g, (B, N))).astype(np.float32)
    out["mtei_similarity"] = (20.0 + 5.0 * _normal(g, (B, N))).astype(np.float32)
    onehot = np.concatenate([np.eye(N - 1, dtype=np.uint8), np.zeros((1, N - 1), dtype=np.uint8)], 0)
    answer = g.integers(0, N, size=B)
    out["answer"] = onehot[answer]
    seq = [out[k] for k in BATCH_FIELDS]
    if as_torch:
        seq = [torch.from_numpy(np.ascontiguousarray(x)) for x in seq]
    return seq


def plant_gold_signal(cfg: DrinConfig, seq: List, strength: float) -> List:
    """In place on a `make_batch` 15-sequence (numpy arrays or CPU tensors): makes the gold candidate findable, as a stand-in
    task for training checks (the real datasets are not available offline).  The gold candidate's text rows become the
    mention's span mean (`ghmfc.py:54-60`) plus `1 / strength` times their original noise - WikiDiverse layout: its pooled
    row; WikiMEL layout: its token-0 row (what the text-text edge reads, `model.py:73-75`) and the tokens the pooling of
    `ghmfc.py:245-249` averages.  Mentions whose answer row is all zero (gold not among the candidates) stay as drawn."""
    as_np = [t.numpy() if torch.is_tensor(t) else t for t in seq]      # views: the edits land in the caller's tensors
    text, start, end, etext, emask, answer = as_np[0], as_np[2], as_np[3], as_np[7], as_np[8], as_np[14]
    gold = np.where(answer.any(1), answer.argmax(1), -1)
    for i in np.nonzero(gold >= 0)[0]:
        span = text[i, start[i]: end[i]].mean(0)
        g = gold[i]
        if cfg.token_level_entities:
            ntok = int(emask[i, g].sum())
            rows = [0] + list(range(1, max(1, ntok - 1)))
            etext[i, g, rows] = span[None, :] + etext[i, g, rows] / strength
        else:
            etext[i, g] = span + etext[i, g] / strength
    return seq


def make_learnable_batch(cfg: DrinConfig, batch: int, seed: int, strength: float = 2.0, **kw) -> List[torch.Tensor]:
    """`make_batch` with the gold candidate planted (`plant_gold_signal`)."""
    return plant_gold_signal(cfg, make_batch(cfg, batch, seed, **kw), strength)


STATE_DICT_SHAPES = lambda D, R, layers, vector=False: (  # noqa: E731 - SURVEY.md §8(b) state_dict contract
    [
        ("vertex_encoder.mention_text_encoder.final_layer.linear.weight", (D, D)),
        ("vertex_encoder.mention_text_encoder.final_layer.linear.bias", (D,)),
        ("vertex_encoder.entity_text_encoder.final_layer.weight", (D, D)),
        ("vertex_encoder.entity_text_encoder.final_layer.bias", (D,)),
        ("vertex_encoder.mention_image_linear.weight", (D, R)),
        ("vertex_encoder.mention_image_linear.bias", (D,)),
        ("vertex_encoder.entity_image_linear.weight", (D, R)),
        ("vertex_encoder.entity_image_linear.bias", (D,)),
    ]
    + [
        (f"gcn_layers.{l}.{name}", shape)
        for l in range(layers)
        for name, shape in (
            [("w_h.weight", (D, D)), ("w_h.bias", (D,))]
            + ([("w_m.weight", (D, D)), ("w_m.bias", (D,))] if vector else [])   # model.py:112
            + [("w_u.weight", (D // 2 if vector else D, D)), ("w_u.bias", (D // 2 if vector else D,)),
               ("w_v.weight", (D // 2 if vector else D, D)), ("w_v.bias", (D // 2 if vector else D,)),
               ("layer_norm.weight", (D,)), ("layer_norm.bias", (D,))]
        )
    ]
)


def make_state_dict(cfg: DrinConfig, seed: int = 7, as_torch: bool = True) -> Dict[str, "torch.Tensor"]:
    """Portable weights with the reference's key names and torch-default scale.

    Linear weights/biases ~ U(+-1/sqrt(fan_in)) like `nn.Linear`'s kaiming-uniform(a=sqrt(5));
    LayerNorm affine is perturbed away from (1, 0) so that parity tests exercise it.
    """
    g = _rng(seed, 1)
    sd: Dict[str, np.ndarray] = {}
    fan_in = None
    for key, shape in STATE_DICT_SHAPES(cfg.gcn_embed_dim, cfg.resnet_embed_dim, cfg.num_gcn_layers,
                                        cfg.gcn_edge_feature == "vector"):
        if "layer_norm.weight" in key:
            v = 1.0 + 0.1 * _normal(g, shape)
        elif "layer_norm.bias" in key:
            v = 0.1 * _normal(g, shape)
        else:
            if len(shape) == 2:
                fan_in = shape[1]
            bound = 1.0 / np.sqrt(fan_in)
            v = (g.random(size=shape, dtype=np.float32) * 2.0 - 1.0) * bound
        sd[key] = v.astype(np.float32)
    if as_torch:
        return {k: torch.from_numpy(v) for k, v in sd.items()}
    return sd


def make_device_batch(cfg: DrinConfig, batch: int, seed: int, device, dtype=torch.float32,
                      min_tokens: int = 4, generator: Optional[torch.Generator] = None) -> List[torch.Tensor]:
    """Benchmark-sized batch drawn directly on `device` (no host staging).

    Same shapes / distributions as `make_batch`, but from torch's device generator: used
    where the batch is far too large to draw on the host (bench.py, full-size property
    tests).  Parity against the oracle at these sizes is checked on slices copied back.
    """
    B, N = batch, cfg.num_candidates_model
    D, R, L, P = cfg.bert_embed_dim, cfg.resnet_embed_dim, cfg.max_mention_sentence_len, cfg.resnet_num_region
    Km, Ke, T = cfg.object_topk_mention, cfg.object_topk_entity, cfg.max_entity_attr_token_len
    g = generator or torch.Generator(device=device)
    if generator is None:
        g.manual_seed(seed)
    rn = lambda *s: torch.randn(*s, device=device, dtype=dtype, generator=g)  # noqa: E731
    ru = lambda *s: torch.rand(*s, device=device, dtype=torch.float32, generator=g)  # noqa: E731
    span = torch.randint(1, 6, (B,), device=device, generator=g).clamp_(max=L - 1)
    start = torch.randint(1, 20, (B,), device=device, generator=g)
    start = torch.minimum(start, L - span)
    ar = torch.arange(L, device=device)
    mask = (ar[None] < (start + span + 3).clamp(max=L)[:, None]).to(torch.int64)
    seq = [rn(B, L, D), mask, start.to(torch.int64), (start + span).to(torch.int64), rn(B, P, R), rn(B, Km, 1, R), ru(B, Km)]
    if cfg.token_level_entities:
        ntok = torch.randint(min(min_tokens, T), T + 1, (B, N), device=device, generator=g)
        emask = (torch.arange(T, device=device)[None, None] < ntok[..., None]).to(torch.int64)
        seq += [rn(B, N, T, D), emask, rn(B, N, 1, R), rn(B, N, Ke, 1, R)]
    else:
        seq += [rn(B, N, D), torch.zeros(B, dtype=torch.int64, device=device), rn(B, N, R), rn(B, N, Ke, R)]
    seq += [ru(B, N, Ke), 20.0 + 5.0 * rn(B, N).float(), 20.0 + 5.0 * rn(B, N).float()]
    onehot = torch.cat([torch.eye(N - 1, dtype=torch.uint8, device=device),
                        torch.zeros(1, N - 1, dtype=torch.uint8, device=device)], 0)
    seq.append(onehot[torch.randint(0, N, (B,), device=device, generator=g)])
    return seq
