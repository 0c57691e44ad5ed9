"""Drop-in `Model` for the reference's `drin/model.py:156-209`, running on hand-written HIP kernels.

Same constructor side effects (parameter creation order, hence same-seed initial weights,
`train.py:134-136`), same `state_dict` keys, same `forward(batch)` signature
(the 14-sequence of `drin/data.py:110-126` already on the device, `train.py:33`) and the same
`[B, N]` fp32 result.  The arithmetic itself happens in `libdrin_hip.so` behind the C ABI of
`include/drin_hip.h`; PyTorch only owns memory, the stream and the autograd graph edge.

There is no eager/PyTorch fallback: if the library is missing or the tensors are not on an
AMD GPU, `forward` raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

import torch
from torch import nn

from . import _lib
from .config import DrinConfig, default_config


# ---- parameter containers mirroring the reference's module tree (keys of SURVEY.md §8b) -------------
class _AvgLinear(nn.Module):  # baselines/ghmfc.py:63-69
    def __init__(self, in_dim: int, out_dim: int):
        super().__init__()
        self.linear = nn.Linear(in_dim, out_dim)


class _MentionEncoder(nn.Module):  # baselines/ghmfc.py:152-165 (linear / offline branch)
    def __init__(self, dim: int):
        super().__init__()
        self.final_layer = _AvgLinear(dim, dim)


class _EntityEncoder(nn.Module):  # baselines/ghmfc.py:202-214 (linear / offline branch)
    def __init__(self, dim: int):
        super().__init__()
        self.final_layer = nn.Linear(dim, dim)


class VertexEncoder(nn.Module):  # drin/model.py:13-24
    def __init__(self, cfg: DrinConfig):
        super().__init__()
        self.mention_text_encoder = _MentionEncoder(cfg.bert_embed_dim)
        self.entity_text_encoder = _EntityEncoder(cfg.bert_embed_dim)
        self.mention_image_linear = nn.Linear(cfg.resnet_embed_dim, cfg.gcn_embed_dim)
        self.entity_image_linear = nn.Linear(cfg.resnet_embed_dim, cfg.gcn_embed_dim)


class GCNLayer(nn.Module):  # drin/model.py:109-119 (scaler edges: w_m is Identity and owns nothing)
    def __init__(self, cfg: DrinConfig):
        super().__init__()
        d = cfg.gcn_embed_dim
        vector = cfg.gcn_edge_feature == "vector"
        self.w_h = nn.Linear(d, d)
        if vector:
            self.w_m = nn.Linear(d, d)                                      # model.py:112
        self.w_u, self.w_v = [nn.Linear(d, d // 2 if vector else d) for _ in range(2)]   # model.py:113-116
        self.layer_norm = nn.LayerNorm(d)


# precisions that are modes of the fused inference path only: whatever else they meet (training, the per-entity cache,
# geometries off the fused path, traced forwards) runs split-bf16
_FUSED_ONLY = (_lib.PREC_BF16X3_IF16,)
_PLANES = (_lib.PREC_BF16X3, _lib.PREC_BF16X3_ALL, _lib.PREC_BF16X3_IF16)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _param_list(model: "Model") -> List[torch.Tensor]:
    ve = model.vertex_encoder
    ps = [
        ve.mention_text_encoder.final_layer.linear.weight, ve.mention_text_encoder.final_layer.linear.bias,
        ve.entity_text_encoder.final_layer.weight, ve.entity_text_encoder.final_layer.bias,
        ve.mention_image_linear.weight, ve.mention_image_linear.bias,
        ve.entity_image_linear.weight, ve.entity_image_linear.bias,
    ]
    for layer in model.gcn_layers:
        ps += [layer.w_h.weight, layer.w_h.bias, layer.w_u.weight, layer.w_u.bias, layer.w_v.weight, layer.w_v.bias,
               layer.layer_norm.weight, layer.layer_norm.bias]
        if hasattr(layer, "w_m"):
            ps += [layer.w_m.weight, layer.w_m.bias]
    return ps


def _fill_params(struct, tensors: Sequence[Optional[torch.Tensor]], per_layer: int = 8) -> None:
    """`per_layer`: 8 tensors per GCN layer, 10 with vector edges (w_m, b_m appended)."""
    names = ("w_mention_text", "b_mention_text", "w_entity_text", "b_entity_text",
             "w_mention_image", "b_mention_image", "w_entity_image", "b_entity_image")
    for n, t in zip(names, tensors[:8]):
        setattr(struct, n, _ptr(t))
    lnames = ("w_h", "b_h", "w_u", "b_u", "w_v", "b_v", "ln_weight", "ln_bias", "w_m", "b_m")[:per_layer]
    for l in range((len(tensors) - 8) // per_layer):
        for j, n in enumerate(lnames):
            setattr(struct.layer[l], n, _ptr(tensors[8 + per_layer * l + j]))


@torch.no_grad()
def _pool_tokens(text: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """Masked token mean of `ghmfc.py:245-249` for `text [..., T, D]` (fp32 or bf16, read in place) and
    `mask [..., T]` -> `[..., D]` fp32, by the library's pooling kernel (`drin_pool_fwd`)."""
    lib = _lib.load()
    lead, (T, D) = text.shape[:-2], text.shape[-2:]
    text = text.contiguous().view(-1, T, D)
    if text.dtype not in (torch.float32, torch.bfloat16):
        text = text.to(torch.float32)
    mask = mask.to(torch.int64).contiguous().view(-1, T)
    E = text.shape[0]
    pooled = torch.empty(E, D, dtype=torch.float32, device=text.device)
    c = _lib.DrinConfigC()
    _lib.check(lib.drin_default_config(C.byref(c)))
    c.num_candidates, c.embed_dim, c.entity_tokens = 1, D, T
    c.feature_dtype = _lib.FEAT_BF16 if text.dtype == torch.bfloat16 else _lib.FEAT_F32
    stream = torch.cuda.current_stream(text.device).cuda_stream
    step = 1 << 20
    for e0 in range(0, E, step):                                      # one launch per 2^20 rows
        e1 = min(E, e0 + step)
        c.batch = e1 - e0
        b = _lib.DrinBatchC()
        b.entity_text, b.entity_text_mask = text[e0:e1].data_ptr(), mask[e0:e1].data_ptr()
        _lib.check(lib.drin_pool_fwd(C.byref(c), C.byref(b), pooled[e0:e1].data_ptr(), None, None, stream))
    return pooled.view(*lead, D)


class EntityTable:
    """The WikiMEL entity tables of `drin/data.py:163-175`, resident on the device: text features
    `[E, T, D]` (+ mask `[E, T]`) or pooled `[E, D]`, image `[E, (1,) R]`, object `[E, Ke, (1,) R]`, object
    score `[E, Ke]`.  With it a batch carries candidate INDICES instead of 22 MB of gathered features."""

    def __init__(self, text, mask, image, object, object_score):
        self.text, self.mask, self.image, self.object, self.object_score = text, mask, image, object, object_score
        self.cache_enabled = False
        self.cache_format = "f32"
        self.cache_format_forced = False
        self.cache_format_used: Optional[str] = None          # row format of the cache that is built right now
        self._scale_scan = None
        self._fmt_key = None
        self._cache: Optional[torch.Tensor] = None
        self._cache_key = None
        self._pooled = None

    # a row whose largest |x| exceeds this many times the median row's can dominate a mention's mean_n(ii ei): the mixed-f16
    # format is then not used for the table (see enable_cache)
    MIXED_F16_MAX_ROW_RATIO = 8.0

    def enable_cache(self, on: bool = True, format: str = "f32", force: bool = False) -> "EntityTable":
        """Let inference calls score from a per-entity precompute cache (`drin_build_entity_cache`):
        23.5 KB per entity at D=768 / R=2048, rebuilt by the first inference call after any weight change.
        `format="mixed_f16"` (`DRIN_CACHE_MIXED_F16`): the operands of per-pair scalars - the two edge-update rows and the object
        row - are stored as fp16 under a power-of-two scale per row and field, the vertex contractions and the CLS row stay
        fp32: 16.4 KB per entity, scores within ~2e-7 of the fp32 rows' (`oracle/precision_emulation.py`).
        The format holds a per-pair scalar (edge logit, image-image edge) to ~1e-5 and relies on the mention aggregates
        `mean_n(edge x vertex)` (`model.py:143-144`) averaging that over the candidates.  ONE candidate whose image row is orders of
        magnitude larger than the others' dominates the mean instead (measured: 1e-5 .. 7e-5 on the scores with rows x 1e6 - inside
        the 1e-4 bar, outside the 1e-5 guard).  So the format is used only for tables it is safe for BY CONSTRUCTION: the cache
        build scans the image table once per table version, and when any row's largest |x| exceeds `MIXED_F16_MAX_ROW_RATIO` (8) x
        the median row's - a share of at most 8 / (8 + N - 1) of a mention's aggregate: <= 6e-6 at N = 101 - the table gets
        fp32 rows, with a `UserWarning` saying so (`cache_format_used` tells which format a built cache has).
        `force=True` keeps the fp16 fields whatever the scan finds (the emulation tests that pin the format's limit use it)."""
        if format not in _lib.CACHE_FORMATS:
            raise ValueError(f"cache format {format!r}: one of {sorted(_lib.CACHE_FORMATS)}")
        self.cache_enabled = on
        if not on or format != self.cache_format or force != self.cache_format_forced:
            self._cache, self._cache_key = None, None
        self.cache_format, self.cache_format_forced = format, force
        return self

    def _rows_far_off_scale(self):
        """`(far, rows)`: how many image rows have a largest |x| above MIXED_F16_MAX_ROW_RATIO x the median row's.  One pass over
        the image table per table VERSION (no copy of it: max(amax, -amin); one host read-back), not per cache build."""
        img = self.image
        key = (img.data_ptr(), img._version, tuple(img.shape))
        if self._scale_scan is None or self._scale_scan[0] != key:
            flat = img.reshape(img.shape[0], -1)
            m = torch.maximum(flat.amax(1), -flat.amin(1)).float()
            finite = m[torch.isfinite(m) & (m > 0)]
            far = int((finite > self.MIXED_F16_MAX_ROW_RATIO * finite.median()).sum()) if finite.numel() else 0
            self._scale_scan = (key, far, int(m.numel()))
        return self._scale_scan[1], self._scale_scan[2]

    def _effective_cache_format(self) -> str:
        if self.cache_format != "mixed_f16":
            return self.cache_format
        far, rows = self._rows_far_off_scale()
        if far == 0:
            return "mixed_f16"
        import warnings
        if self.cache_format_forced:
            warnings.warn(f"EntityTable cache format mixed_f16 (forced): {far} of {rows} entity image rows are more than "
                          f"{self.MIXED_F16_MAX_ROW_RATIO:g} x the median row's magnitude; mentions that list such an entity may score up to "
                          f"~1e-4 from the fp32 rows", UserWarning, stacklevel=4)
            return "mixed_f16"
        warnings.warn(f"EntityTable cache format mixed_f16: {far} of {rows} entity image rows are more than {self.MIXED_F16_MAX_ROW_RATIO:g} x "
                      f"the median row's magnitude - such a row can dominate a mention's aggregate and carry its one fp16 edge's rounding "
                      f"(up to ~1e-4) into the scores; this table gets fp32 cache rows instead (format='f32')", UserWarning, stacklevel=4)
        return "f32"

    def invalidate(self) -> "EntityTable":
        """Drop the per-entity cache and the pooled-text copy.  Needed only after the tables were edited through a path
        PyTorch's version counters do not see (`t.data.copy_(...)`, an external kernel writing the storage): in-place torch
        ops on the tensors and replaced tensors are detected by themselves."""
        self._cache, self._cache_key, self._pooled, self._scale_scan = None, None, None, None
        return self

    def _table_key(self):
        return tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in (self.text, self.mask, self.image, self.object, self.object_score)
                     if t is not None)

    def _get_cache(self, call: "_Call", key, pc, prepared: torch.Tensor) -> torch.Tensor:
        fmt = self.cache_format_used if (self._cache is not None and self._fmt_key == (self.cache_format, self.cache_format_forced, self._table_key())) \
            else self._effective_cache_format()
        self._fmt_key = (self.cache_format, self.cache_format_forced, self._table_key())
        self.cache_format_used = fmt
        call.cfg.cache_format = _lib.CACHE_FORMATS[fmt]
        key = (key, call.cfg.precision, call.cfg.num_entities, call.cfg.cache_format, self._table_key())
        if key != self._cache_key or self._cache is None:
            lib = _lib.load()
            n = lib.drin_entity_cache_bytes(C.byref(call.cfg))
            if n == 0:
                raise _lib.DrinError(_lib.E_UNSUPPORTED, lib.drin_last_error().decode())
            self._cache = None                                        # release the stale one before allocating
            cache = torch.empty(n, dtype=torch.uint8, device=call.device)
            ws = torch.empty(max(lib.drin_entity_cache_build_workspace_bytes(C.byref(call.cfg)), 16), dtype=torch.uint8,
                             device=call.device)
            stream = torch.cuda.current_stream(call.device).cuda_stream
            _lib.check(lib.drin_build_entity_cache(C.byref(call.cfg), C.byref(call.batch), C.byref(pc), prepared.data_ptr(),
                                                   cache.data_ptr(), n, ws.data_ptr(), ws.numel(), stream))
            self._cache, self._cache_key = cache, key
        return self._cache

    @property
    def num_entities(self) -> int:
        return self.text.shape[0]

    @torch.no_grad()
    def pooled_text(self, cfg: DrinConfig):
        """`(pooled [E, D], cls [E, D])` of a token-level table: the masked token mean of `ghmfc.py:245-249` and the
        token-0 row of `model.py:73-75`, per ENTITY.  Neither depends on the weights, so training in table form pools
        every entity once (library kernel, same arithmetic as the per-pair pooling of a gathered batch: identical
        values) instead of gathering and pooling 197 KB of tokens per candidate per step."""
        if self.text.dim() != 3:
            raise ValueError("pooled_text: the table already holds pooled text [E, D]")
        key = (self.text.data_ptr(), self.text._version, self.mask.data_ptr(), self.mask._version)
        if self._pooled is None or self._pooled[0] != key:
            text = self.text.contiguous()
            self._pooled = (key, _pool_tokens(text, self.mask), text[:, 0, :].contiguous())   # [E, D] each
        return self._pooled[1], self._pooled[2]

    def to(self, device) -> "EntityTable":
        mv = lambda t: None if t is None else t.to(device)  # noqa: E731
        moved = EntityTable(mv(self.text), mv(self.mask), mv(self.image), mv(self.object), mv(self.object_score))
        moved.cache_enabled, moved.cache_format, moved.cache_format_forced = self.cache_enabled, self.cache_format, self.cache_format_forced
        return moved

    def gather(self, index: torch.Tensor):
        """Per-pair tensors exactly as `MELData.__getitem__` + collate would deliver them (data.py:87-93)."""
        mask = self.mask[index] if self.mask is not None else torch.zeros(index.shape[0], dtype=torch.int64, device=index.device)
        return [self.text[index], mask, self.image[index], self.object[index], self.object_score[index]]


class IndexedBatch:
    """A batch in table form: the seven mention-side tensors of the 14-sequence, the device-resident
    `EntityTable`, `candidates [B, N]` int64 rows of it, the two CLIP similarity matrices `[B, N]`."""

    def __init__(self, mention: Sequence[torch.Tensor], table: EntityTable, candidates: torch.Tensor,
                 miet_similarity: torch.Tensor, mtei_similarity: torch.Tensor):
        if len(mention) != 7:
            raise ValueError("mention part must be the first 7 tensors of the 14-sequence (drin/data.py:110-117)")
        self.mention, self.table, self.candidates = list(mention), table, candidates
        self.miet_similarity, self.mtei_similarity = miet_similarity, mtei_similarity

    def gathered(self) -> List[torch.Tensor]:
        """The equivalent 14-sequence (entity rows materialised with torch indexing)."""
        return self.mention + self.table.gather(self.candidates) + [self.miet_similarity, self.mtei_similarity]

    def gathered_pooled(self, cfg: DrinConfig):
        """`(14-sequence with pooled entity text [B, N, D], cls rows [B, N, D])`: what a training step needs of a
        token-level table - 22 KB per candidate instead of 219 KB, and no per-step token pooling."""
        t, idx = self.table, self.candidates
        pooled, cls = t.pooled_text(cfg)
        dummy = torch.zeros(idx.shape[0], dtype=torch.int64, device=idx.device)
        seq = self.mention + [pooled[idx], dummy, t.image[idx], t.object[idx], t.object_score[idx],
                              self.miet_similarity, self.mtei_similarity]
        return seq, cls[idx]


def _split_mentions(batch, step: int):
    """Slices of at most `step` mentions of a 14-sequence or an `IndexedBatch` (views, nothing is copied)."""
    if isinstance(batch, IndexedBatch):
        B = batch.candidates.shape[0]
        for b0 in range(0, B, step):
            sl = slice(b0, min(B, b0 + step))
            yield IndexedBatch([t[sl] for t in batch.mention], batch.table, batch.candidates[sl],
                               batch.miet_similarity[sl], batch.mtei_similarity[sl])
    else:
        B = batch[0].shape[0]
        for b0 in range(0, B, step):
            sl = slice(b0, min(B, b0 + step))
            yield [t[sl] if torch.is_tensor(t) and t.dim() > 0 and t.shape[0] == B else t for t in batch]


class _Call:
    """One forward's C structs; keeps the tensors they point into alive."""

    def __init__(self, cfg: DrinConfig, batch: Sequence[torch.Tensor], precision: int,
                 entity_index: Optional[torch.Tensor] = None, keep_bf16: bool = False,
                 entity_text_cls: Optional[torch.Tensor] = None, index_status: Optional[torch.Tensor] = None):
        """`keep_bf16`: the caller will take a path that reads bf16-stored features in place (fused inference);
        otherwise bf16 features are widened to fp32 here (exact) - e.g. for training.
        `entity_text_cls` `[B, N, D]`: the token-0 rows of the text-text edge when `entity_text_feature` holds token
        means pooled ahead of time (`EntityTable.pooled_text`)."""
        if len(batch) not in (14, 15):
            raise ValueError(f"batch must be the 14-sequence of drin/data.py:110-126 (got {len(batch)} items)")
        (mtf, _mask, start, end, mimg, mobj, mscore, etf, emask, eimg, eobj, escore, miet, mtei) = batch[:14]
        dev = mtf.device
        if dev.type != "cuda":
            raise RuntimeError("drin_amd.Model runs on an AMD GPU only (no CPU / eager fallback); move the batch to 'cuda'")

        def f32(t):
            if t.device != dev:
                raise RuntimeError("all batch tensors must be on the same device")
            return t.to(torch.float32).contiguous()

        # bf16 feature storage (BASELINE configs 2-3): all six feature tensors, or none
        feats = (mtf, mimg, mobj, etf, eimg, eobj)
        n_bf16 = sum(t.dtype == torch.bfloat16 for t in feats)
        bf16 = keep_bf16 and n_bf16 == len(feats)
        if keep_bf16 and 0 < n_bf16 < len(feats):
            raise ValueError("bf16 feature storage: give all six feature tensors (mention text / image / object, entity "
                             "text / image / object) as bfloat16, or none")

        def feat(t):
            if t.device != dev:
                raise RuntimeError("all batch tensors must be on the same device")
            return t.contiguous() if bf16 else t.to(torch.float32).contiguous()

        def i64(t):
            return t.to(device=dev, dtype=torch.int64).contiguous()

        mtf, mimg, mobj, etf, eimg, eobj = map(feat, feats)
        mscore, escore, miet, mtei = map(f32, (mscore, escore, miet, mtei))
        start, end = i64(start), i64(end)
        B, L, D = mtf.shape
        N = cfg.num_candidates_model
        R = mimg.shape[-1]
        table = entity_index is not None
        if table:
            # entity tensors are tables [E, ...]: give them the per-pair ranks by viewing E as (E, 1)
            E = etf.shape[0]
            entity_index = i64(entity_index)
            if tuple(entity_index.shape) != (B, N):
                raise ValueError(f"candidates has shape {tuple(entity_index.shape)}, expected {(B, N)}")
            etf, eimg, eobj, escore = (t.unsqueeze(1) for t in (etf, eimg, eobj, escore))
            if emask is not None and torch.is_tensor(emask) and emask.dim() >= 2:
                emask = emask.unsqueeze(1)
            lead = (E, 1)
        else:
            lead = (B, N)
        token_level = etf.dim() == 4                                  # model.py:73-75
        if D != cfg.bert_embed_dim or R != cfg.resnet_embed_dim:
            raise ValueError(f"feature dims ({D}, {R}) do not match the config ({cfg.bert_embed_dim}, {cfg.resnet_embed_dim})")
        if tuple(etf.shape[:2]) != lead:
            raise ValueError(f"entity_text_feature leads with {tuple(etf.shape[:2])}, expected {lead}")
        if mobj.dim() not in (3, 4) or eobj.dim() not in (4, 5) or eimg.dim() not in (3, 4):
            raise ValueError("unexpected rank for object / image features (model.py:43-44,78-83)")
        Km = mobj.shape[1]
        Ke = eobj.shape[2]
        for name, t, shape in (
            ("mention_image_feature", mimg, (B, mimg.shape[1], R)),
            ("mention_object_score", mscore, (B, Km)),
            ("entity_object_score", escore, lead + (Ke,)),
            ("miet_similarity", miet, (B, N)),
            ("mtei_similarity", mtei, (B, N)),
            ("mention_start_pos", start, (B,)),
            ("mention_end_pos", end, (B,)),
        ):
            if tuple(t.shape) != shape:
                raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {shape}")
        if tuple(eimg.shape[0:2]) != lead or tuple(eobj.shape[0:2]) != lead or eimg.shape[-1] != R or eobj.shape[-1] != R or mobj.shape[-1] != R:
            raise ValueError("entity/mention image or object feature shape mismatch")
        if token_level:
            emask = i64(emask)
            if tuple(emask.shape) != tuple(etf.shape[:3]):
                raise ValueError(f"entity_text_mask has shape {tuple(emask.shape)}, expected {tuple(etf.shape[:3])}")
        else:
            emask = None
        if entity_text_cls is not None:
            if token_level:
                raise ValueError("entity_text_cls goes with pooled entity text")
            entity_text_cls = f32(entity_text_cls)
            want = (etf.shape[0], D) if table else (B, N, D)       # a table of token-0 rows, or per-pair rows
            if tuple(entity_text_cls.shape) != want:
                raise ValueError(f"entity_text_cls has shape {tuple(entity_text_cls.shape)}, expected {want}")
        if index_status is not None and (index_status.dtype != torch.int32 or index_status.numel() < 4 or index_status.device != dev):
            raise ValueError("index_status: int32[4] on the batch's device")
        self.keep = [mtf, start, end, mimg, mobj, mscore, etf, emask, eimg, eobj, escore, miet, mtei, entity_index,
                     entity_text_cls, index_status if table else None]
        self.device = dev
        self.B, self.N, self.D = B, N, D
        c = _lib.DrinConfigC()
        _lib.check(_lib.load().drin_default_config(C.byref(c)))
        c.batch, c.num_candidates, c.embed_dim, c.image_dim = B, N, D, R
        c.mention_tokens, c.image_regions = L, mimg.shape[1]
        c.mention_objects, c.entity_objects = Km, Ke
        c.entity_tokens = etf.shape[2] if token_level else 0
        c.mention_object_inner = mobj.shape[2] if mobj.dim() == 4 else 0
        c.entity_image_inner = eimg.shape[2] if eimg.dim() == 4 else 0
        c.entity_object_inner = eobj.shape[3] if eobj.dim() == 5 else 0
        c.num_layers = cfg.num_gcn_layers
        c.dynamic_edges = 1 if cfg.gcn_edge_type == "dynamic" else 0
        for k in range(4):
            c.edge_enabled[k] = float(cfg.gcn_edge_enabled[k])
        c.layer_norm_eps, c.cosine_eps, c.miei_eps, c.clip_scale = (
            cfg.layer_norm_eps, cfg.cosine_eps, cfg.miei_eps, cfg.clip_logit_scale)
        c.precision = precision
        c.num_entities = etf.shape[0] if table else 0
        c.vector_edges = 1 if cfg.gcn_edge_feature == "vector" else 0
        c.feature_dtype = _lib.FEAT_BF16 if bf16 else _lib.FEAT_F32
        c.vertex_activation = _lib.ACTIVATIONS[cfg.gcn_vertex_activation]
        c.edge_activation = _lib.ACTIVATIONS[cfg.gcn_edge_activation]
        self.per_layer = 10 if c.vector_edges else 8
        self.cfg = c
        b = _lib.DrinBatchC()
        for name, t in zip(("mention_text", "mention_start", "mention_end", "mention_image", "mention_object",
                            "mention_object_score", "entity_text", "entity_text_mask", "entity_image",
                            "entity_object", "entity_object_score", "miet_similarity", "mtei_similarity",
                            "entity_index", "entity_text_cls", "index_status"), self.keep):
            setattr(b, name, _ptr(t))
        self.batch = b

    def workspace(self, training: bool) -> torch.Tensor:
        n = _lib.load().drin_workspace_bytes(C.byref(self.cfg), 1 if training else 0)
        if n == 0 and self.B > 0:
            raise _lib.DrinError(_lib.E_SHAPE, _lib.load().drin_last_error().decode())
        return torch.empty(max(n, 16), dtype=torch.uint8, device=self.device)


class _Prepared:
    """Weight-only products of the fused inference path (drin_prepare), cached per weight version."""

    def __init__(self):
        self.key = None
        self.buf: Optional[torch.Tensor] = None
        self.generation = 0          # bumped by every rebuild: what caches derived from the folds are keyed on

    def invalidate(self) -> None:
        self.key = None

    def get(self, call: "_Call", params: Sequence[torch.Tensor], pc) -> torch.Tensor:
        key = (call.cfg.embed_dim, call.cfg.image_dim, call.cfg.dynamic_edges) + tuple((p.data_ptr(), p._version) for p in params)
        if key != self.key or self.buf is None or self.buf.device != call.device:
            lib = _lib.load()
            n = lib.drin_prepared_bytes(C.byref(call.cfg))
            self.buf = torch.empty(n, dtype=torch.uint8, device=call.device)
            stream = torch.cuda.current_stream(call.device).cuda_stream
            _lib.check(lib.drin_prepare(C.byref(call.cfg), C.byref(pc), self.buf.data_ptr(), n, stream))
            self.key = key
            self.generation += 1
        return self.buf


def _dead_param_indices(n_params: int, per_layer: int, dynamic: bool) -> List[int]:
    """Positions in `_param_list` order of the parameters the score does not depend on (`.grad is None` in the reference)."""
    nl = (n_params - 8) // per_layer
    dead_layers = range(nl - 1, nl) if dynamic else range(nl)
    return [8 + per_layer * l + j for l in dead_layers for j in (2, 3, 4, 5) + ((8, 9) if per_layer == 10 else ())]


def _bucket_layout(params: Sequence[torch.Tensor], per_layer: int, dynamic: bool):
    """`(offsets, live_floats, total_floats)`: where each parameter of `_param_list` order sits in a flat fp32 bucket - the
    parameters that receive a gradient first (the all-reduced / Adam-stepped prefix), the dead ones behind them; every
    slot starts on a 256-byte boundary."""
    dead = set(_dead_param_indices(len(params), per_layer, dynamic))
    offsets, off = [0] * len(params), 0
    for want_dead in (False, True):
        for i, p in enumerate(params):
            if (i in dead) == want_dead:
                offsets[i] = off
                off += (p.numel() + 63) & ~63
        if not want_dead:
            live = off
    return offsets, live, off


class _DrinScore(torch.autograd.Function):
    """Autograd edge around drin_forward / drin_backward (loss.backward() of train.py:33-34)."""

    @staticmethod
    def forward(ctx, call: _Call, prepared: Optional[_Prepared], training: bool, *params: torch.Tensor):
        lib = _lib.load()
        versions = params
        params = tuple(p.detach().contiguous() for p in params)
        pc = _lib.DrinParamsC()
        _fill_params(pc, params, call.per_layer)
        if not training and prepared is not None and lib.drin_fused_supported(C.byref(call.cfg)) == _lib.OK:
            # inference: fused two-layer path on weights folded once per weight version
            pbuf = prepared.get(call, versions, pc)
            n = lib.drin_fused_workspace_bytes(C.byref(call.cfg))
            ws = torch.empty(max(n, 16), dtype=torch.uint8, device=call.device)
            scores = torch.empty(call.B, call.N, dtype=torch.float32, device=call.device)
            stream = torch.cuda.current_stream(call.device).cuda_stream
            _lib.check(lib.drin_forward_prepared(C.byref(call.cfg), C.byref(call.batch), C.byref(pc), pbuf.data_ptr(),
                                                 ws.data_ptr(), ws.numel(), scores.data_ptr(), stream))
            return scores
        ws = call.workspace(training)
        scores = torch.empty(call.B, call.N, dtype=torch.float32, device=call.device)
        stream = torch.cuda.current_stream(call.device).cuda_stream
        ready = getattr(call, "params_ready", None)       # train.OverlappedStep: the previous update is still running on a side stream
        if ready is not None:
            # the pooling passes and static edges of THIS step run under it; the stream waits before the first weight is read
            _lib.check(lib.drin_forward_staged(C.byref(call.cfg), C.byref(call.batch), C.byref(pc), ws.data_ptr(), ws.numel(),
                                               scores.data_ptr(), 1 if training else 0, None, ready.cuda_event, stream))
        else:
            _lib.check(lib.drin_forward(C.byref(call.cfg), C.byref(call.batch), C.byref(pc), ws.data_ptr(), ws.numel(),
                                        scores.data_ptr(), 1 if training else 0, None, stream))
        ctx.call, ctx.ws, ctx.pc, ctx.params = call, ws, pc, params
        return scores

    @staticmethod
    def backward(ctx, grad_scores: torch.Tensor):
        lib = _lib.load()
        call, params = ctx.call, ctx.params
        owner = getattr(call, "owner", None)
        if owner is not None and owner.grad_bucket_enabled:
            # every gradient is a view of ONE flat fp32 bucket, zeroed with one memset: autograd's AccumulateGrad adopts the
            # views as .grad, so the data-parallel all-reduce and the one-launch Adam run on the bucket itself, copy-free
            grads = owner._bucket_grads(params)
        else:
            grads = [torch.empty_like(p) for p in params]
            torch._foreach_zero_(grads)                               # one multi-tensor launch instead of 24 fills
        gc = _lib.DrinParamGradsC()
        _fill_params(gc, grads, call.per_layer)
        g = grad_scores.to(torch.float32).contiguous()
        stream = torch.cuda.current_stream(call.device).cuda_stream
        # data-parallel overlap (drin_backward_staged): the library records `ready` once the GCN layers' gradients are
        # complete; the hook - GradBucket's - starts their all-reduce behind it, under the vertex encoders' dW products
        hook = getattr(owner, "_layers_ready_hook", None) if owner is not None and owner.grad_bucket_enabled else None
        staged = hook is not None and owner._grad_flat is not None and grads[0].data_ptr() == owner._grad_flat.data_ptr()
        if staged:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(call.device))        # creates the hipEvent_t; the library records it again
            _lib.check(lib.drin_backward_staged(C.byref(call.cfg), C.byref(call.batch), C.byref(ctx.pc), ctx.ws.data_ptr(),
                                                ctx.ws.numel(), g.data_ptr(), C.byref(gc), ready.cuda_event, stream))
            offsets, live, _total = owner.bucket_layout()
            hook(owner._grad_flat[:live], offsets[8], ready)             # [vertex encoders | GCN layers | dead]: the layers start at slot 8
        else:
            _lib.check(lib.drin_backward(C.byref(call.cfg), C.byref(call.batch), C.byref(ctx.pc), ctx.ws.data_ptr(),
                                         ctx.ws.numel(), g.data_ptr(), C.byref(gc), stream))
        out = list(grads)
        # parameters the score does not depend on get no gradient at all in the reference (.grad is None):
        # the last layer's edge update is dead (model.py:130-134), and static edges never use w_u / w_v (/ w_m)
        for i in _dead_param_indices(len(params), call.per_layer, bool(call.cfg.dynamic_edges)):
            out[i] = None
        return (None, None, None, *out)


def _invalidate_after_load(module, _incompatible_keys) -> None:
    module.invalidate()


class Model(nn.Module):
    """`model_module.Model()` of `train.py:136` (drin/model.py:156-162)."""

    # mentions per library call: larger batches are scored in slices of this many (mentions are independent; the candidate
    # grouping of the per-mention sums depends on the size of the call - csrc/fused_forward.hip: 16 / 48 / whole-list
    # workgroups - so a mention's scores in calls of different sizes agree to fp32 re-association, <= 5e-6, and are the same
    # bits every run within one call size); the one-workgroup-per-(mention, chunk) grids of the row kernels stop at 65 535 mentions
    MAX_CALL_MENTIONS = 32768

    def __init__(self, cfg: Optional[DrinConfig] = None, precision: str = "bf16x3", fused: bool = True,
                 grad_bucket: bool = True):
        """`cfg`: None reads an importable reference `common.args` (the no-argument `Model()` of `train.py:136`), else the
        WikiDiverse defaults (`config.default_config`).
        `precision`: "bf16x3" (default: split-bf16 MFMA, fp32-equivalent - measured <= 1.4e-6 on the scores against the
        reference's fp32 forward, <= 6e-6 on trained weights, bar 1e-4; 2x the rate of exact fp32), "f32" (exact fp32 MFMA) or
        "bf16x3_if16" (precision by contraction: split-bf16 except the folded entity-image contraction, which runs ONE pass of the
        FP16 matrix instruction on the image rows the stream kernel hands over as fp16 under a power-of-two scale per row, where the
        candidate list is long enough - N >= 64 - for the mean over candidates behind it to average its rounding noise down:
        <= 4e-6 on the scores at N = 101 with freshly initialised weights, <= 2e-5 with trained ones - inside the bar either way;
        for per-pair (not table-form) fp32-stored image rows of large inference calls at D = 768 / R = 2048; shorter lists, small
        calls, bf16-stored features and every other path run "bf16x3" bit for bit);
        `fused`: let inference calls (no parameter needs a gradient) take the folded two-layer path;
        `grad_bucket`: backward writes every gradient into one flat bucket the `.grad`s are views of (like DDP's
        `gradient_as_bucket_view`: a `.grad` kept across `zero_grad(set_to_none=True)` + `backward()` is overwritten)."""
        super().__init__()
        self.cfg = cfg or default_config()
        self.cfg.validate()
        self._prepared = _Prepared() if fused else None
        self.grad_bucket_enabled = grad_bucket
        self._grad_flat: Optional[torch.Tensor] = None
        self._bucket_in_flight = False                       # handed out by a backward pass that is still running
        self._layers_ready_hook = None                       # set by train.GradBucket(overlap=True): see _DrinScore.backward
        self._params_ready = None                            # set by train.OverlappedStep: event behind an update still in flight
        self._param_flat: Optional[torch.Tensor] = None
        self._layout = None
        # table-form calls: where the kernels report a candidate row outside the entity tables (drin_batch.index_status)
        self.validate_indices = os.environ.get("DRIN_VALIDATE", "") not in ("", "0")
        self._index_status: Dict[torch.device, torch.Tensor] = {}
        self._index_watch: List = []                          # (host copy, event): async read-backs of calls still in flight
        self._index_pool: List = []                           # landed (pinned host words, event) pairs, re-used
        self.register_load_state_dict_post_hook(_invalidate_after_load)
        modes = {"f32": _lib.PREC_F32, "bf16x3": _lib.PREC_BF16X3, "bf16x3_all": _lib.PREC_BF16X3_ALL, "bf16x3_if16": _lib.PREC_BF16X3_IF16}
        if precision not in modes:
            # ("bf16" - every contraction in one bf16 pass, 5e-4 - and "bf16x3_i1" - the image contraction in one bf16 pass, 1.6e-4 on
            #  trained weights - were outside the path's 1e-4 bar and were removed in round 5)
            raise ValueError(f"precision {precision!r}: one of {sorted(modes)}")
        self.precision = modes[precision]
        self.vertex_encoder = VertexEncoder(self.cfg)          # model.py:159 (RNG order: ghmfc.py:165,211; model.py:23-24)
        self.gcn_layers = nn.ModuleList([GCNLayer(self.cfg) for _ in range(self.cfg.num_gcn_layers)])  # model.py:161

    def invalidate(self) -> "Model":
        """Forget the folded weights of the fused inference path (and with them every `EntityTable` cache keyed on them).
        The folds are keyed on `(data_ptr, _version)` of the parameters, so optimiser steps, `load_state_dict` and in-place
        torch ops are detected by themselves; writes PyTorch's version counter does not see - `p.data.copy_(...)`,
        `p.data.mul_(...)`, EMA / weight swapping through `.data`, an external kernel - need this call."""
        if self._prepared is not None:
            self._prepared.invalidate()
        return self

    # ---- flat buckets (one all-reduce per step; train.py:55-56: one Adam launch) -------------------------
    def bucket_layout(self):
        params = _param_list(self)
        key = tuple(p.numel() for p in params)
        if self._layout is None or self._layout[0] != key:
            pl = 10 if self.cfg.gcn_edge_feature == "vector" else 8
            self._layout = (key,) + _bucket_layout(params, pl, self.cfg.gcn_edge_type == "dynamic")
        return self._layout[1:]

    def _bucket_grads(self, params: Sequence[torch.Tensor]) -> List[Optional[torch.Tensor]]:
        """Views of the flat gradient bucket for `params` (`_param_list` order; the dead ones get their own slots behind
        the live prefix and are never returned to autograd), zeroed with one memset.  A bucket some `.grad` still aliases
        (gradient accumulation over several backward passes) is left alone and a fresh one is taken - and so is a bucket
        an EARLIER node of the backward pass that is running now has handed out (two scoring calls in one graph:
        `loss(model(b1)) + loss(model(b2))`, or a training batch above MAX_CALL_MENTIONS): its views sit in autograd's
        input buffers while every `.grad` is still None, and the engine sums the two nodes' gradients itself."""
        offsets, live, total = self.bucket_layout()
        dev = params[0].device
        flat = self._grad_flat
        if flat is not None and (flat.device != dev or flat.numel() != total):
            flat = None
        if flat is not None:
            base = flat.untyped_storage().data_ptr()
            if any(p.grad is not None and p.grad.untyped_storage().data_ptr() == base for p in self.parameters()):
                flat = None
        views = lambda f: [f[o:o + p.numel()].view(p.shape) for o, p in zip(offsets, params)]
        if self._bucket_in_flight:
            extra = torch.empty(total, dtype=torch.float32, device=dev)   # not the model's bucket: the sums will be new tensors
            extra[:live].zero_()
            return views(extra)
        if flat is None:
            flat = torch.empty(total, dtype=torch.float32, device=dev)
        self._grad_flat = flat
        flat[:live].zero_()
        self._bucket_in_flight = True
        try:                                                  # cleared when the engine finishes this backward pass
            torch.autograd.Variable._execution_engine.queue_callback(self._bucket_landed)
        except RuntimeError:                                  # not inside a backward pass (a direct call): nothing in flight
            self._bucket_in_flight = False
        return views(flat)

    def _bucket_landed(self) -> None:
        self._bucket_in_flight = False

    def grad_bucket(self) -> Optional[torch.Tensor]:
        """The live prefix of the flat gradient bucket when every current `.grad` is a view of it, else None."""
        flat = self._grad_flat
        if flat is None:
            return None
        offsets, live, _total = self.bucket_layout()
        base, seen = flat.data_ptr(), False
        for o, p in zip(offsets, _param_list(self)):
            if p.grad is None:
                continue
            if p.grad.data_ptr() != base + 4 * o or not p.grad.is_contiguous():
                return None
            seen = True
        return flat[:live] if seen else None

    def flatten_parameters(self) -> torch.Tensor:
        """Move every parameter into one flat fp32 bucket laid out like the gradient bucket (`p.data` become views; values,
        `state_dict` keys and `load_state_dict` are unaffected).  Returns the bucket; idempotent; redo after `.to(...)`."""
        params = _param_list(self)
        offsets, _live, total = self.bucket_layout()
        flat = self._param_flat
        ok = (flat is not None and flat.device == params[0].device and flat.numel() == total
              and all(p.data_ptr() == flat.data_ptr() + 4 * o and p.is_contiguous() for o, p in zip(offsets, params)))
        if not ok:
            flat = torch.zeros(total, dtype=torch.float32, device=params[0].device)
            with torch.no_grad():
                for o, p in zip(offsets, params):
                    view = flat[o:o + p.numel()].view(p.shape)
                    view.copy_(p.data)
                    p.data = view
            self._param_flat = flat
            self.invalidate()
        return flat

    def __getstate__(self):
        # copy.deepcopy / pickle of the Module: the index watch holds HIP events and pinned buffers of calls in flight - per-process
        # bookkeeping, not state (a copy starts with none; its first table-form call makes its own status words)
        state = self.__dict__.copy()
        state["_index_watch"], state["_index_pool"], state["_index_status"] = [], [], {}
        return state

    # ---- candidate rows outside the entity tables (drin/data.py:87-93 raises IndexError) ------------------------------
    def _status_words(self, device: torch.device) -> torch.Tensor:
        t = self._index_status.get(device)
        if t is None:
            t = self._index_status[device] = torch.zeros(4, dtype=torch.int32, device=device)
        return t

    @staticmethod
    def _raise_bad_index(words) -> None:
        w = [int(x) for x in words]
        value = ((w[3] & 0xFFFFFFFF) << 32 | (w[2] & 0xFFFFFFFF))
        value -= (1 << 64) if value >= (1 << 63) else 0
        what = f"row {value} at pair {w[1]} (b * N + n)" if (w[2] or w[3]) else "a row of a training batch"
        raise IndexError(f"IndexedBatch.candidates: {what} is outside the entity tables (it was clamped: the scores of that call are of the "
                         f"WRONG entity there); the reference's fancy index, drin/data.py:87-93, raises here too")

    def _watch_indices(self, device: torch.device) -> None:
        """After a table-form call: raise at once when validating eagerly, else start an asynchronous read-back of the status
        words which `forward` / `check_indices` look at once it has landed (no synchronisation is added to the step)."""
        if torch.cuda.is_current_stream_capturing():          # inside a graph capture: no read-back, no event (the status words are still
            return                                            # written by the replayed kernels: check_indices() after a replay sees them)
        if self.validate_indices:
            self.check_indices()
            return
        host, ev = self._index_pool.pop() if self._index_pool else (torch.empty(4, dtype=torch.int32, pin_memory=True), torch.cuda.Event())
        host.copy_(self._status_words(device), non_blocking=True)
        ev.record(torch.cuda.current_stream(device))
        self._index_watch.append((host, ev))

    def check_indices(self, wait: bool = True) -> None:
        """Raise `IndexError` if any table-form call since the last check met a candidate row outside `[0, E - 1]`.  `wait=True`
        synchronises with the device (call it where the loop synchronises anyway: `MELRunner.run_epoch` does, at the epoch's
        end); `wait=False` only looks at read-backs that have already landed.  `Model(...).validate_indices = True` (or
        `DRIN_VALIDATE=1`) checks after every call instead - the reference's behaviour, at one synchronisation per call."""
        bad = None
        if wait:
            self._index_watch.clear()
            for t in self._index_status.values():
                w = t.cpu()
                if int(w[0]) != 0 and bad is None:
                    bad = w.tolist()
                if int(w[0]) != 0:
                    t.zero_()
        else:
            while self._index_watch and self._index_watch[0][1].query():
                host, ev = self._index_watch.pop(0)
                if int(host[0]) != 0 and bad is None:
                    bad = host.tolist()
                self._index_pool.append((host, ev))               # pinned words and event are re-used by the next call
            if bad is not None:
                self._index_watch.clear()
                for t in self._index_status.values():
                    t.zero_()
        if bad is not None:
            self._raise_bad_index(bad)

    def forward(self, batch) -> torch.Tensor:
        if self._index_watch and not torch.cuda.is_current_stream_capturing():
            self.check_indices(wait=False)
        B = batch.candidates.shape[0] if isinstance(batch, IndexedBatch) else batch[0].shape[0]
        if B > self.MAX_CALL_MENTIONS:
            return torch.cat([self._forward(part) for part in _split_mentions(batch, self.MAX_CALL_MENTIONS)], 0)
        return self._forward(batch)

    def _forward(self, batch) -> torch.Tensor:
        params = _param_list(self)
        cls = None
        if isinstance(batch, IndexedBatch):
            # table form: inference gathers inside the stream kernel; everything else (training,
            # exact-fp32 precision, geometries off the fused path) gathers with torch indexing first
            inference = not (torch.is_grad_enabled() and any(p.requires_grad for p in params))
            planes = self.precision in _PLANES
            t = batch.table
            if inference and self._prepared is not None and self.cfg.num_gcn_layers == 2 and (planes or t.cache_enabled):
                seq = batch.mention + [t.text, t.mask, t.image, t.object, t.object_score,
                                       batch.miet_similarity, batch.mtei_similarity]
                if t.cache_enabled and t.text.dtype == torch.bfloat16:
                    raise ValueError("the per-entity cache is built from fp32 tables; give EntityTable fp32 features")
                # bf16-stored features are read in place by the fused path (never widened: the table is large)
                prec = _lib.PREC_BF16X3 if (t.cache_enabled and self.precision in _FUSED_ONLY) else self.precision
                call = _Call(self.cfg, seq, prec, entity_index=batch.candidates, keep_bf16=planes and not t.cache_enabled,
                             index_status=self._status_words(batch.candidates.device))
                if _lib.load().drin_fused_supported(C.byref(call.cfg)) == _lib.OK:
                    if t.cache_enabled:                                # per-entity precompute cache
                        out = self._forward_cached(call, t, params)
                        self._watch_indices(call.device)
                        return out
                    if planes:
                        out = self._score(call, self._prepared, False, *params)
                        self._watch_indices(call.device)
                        return out
            if not inference and t.text.dim() == 3:
                # training on a token-level table: every entity's tokens pooled once; the step then reads the pooled /
                # token-0 / image / object tables through the candidate index inside the kernels, or gathers those rows
                call = self._indexed_training_call(batch, planes)
                if call is not None:
                    if call.B == 0:
                        return torch.zeros(0, call.N, dtype=torch.float32, device=call.device)
                    out = self._score(call, None, True, *params)
                    self._watch_indices(call.device)
                    return out
                batch, cls = self._clamped(batch).gathered_pooled(self.cfg)
                self._watch_indices(batch[0].device)
            else:
                batch = self._clamped(batch).gathered()
                self._watch_indices(batch[0].device)
        # grad mode is already off inside Function.forward (and needs_input_grad ignores no_grad), so the
        # caller's mode is read here
        training = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        # features stored as bf16 are read in place by the fused inference path in split-bf16 precision; every
        # other path (training, exact fp32, geometries off the fused path) gets them widened to fp32 - exact
        in_place = (not training and self._prepared is not None and self.cfg.num_gcn_layers == 2
                    and self.precision in _PLANES)
        # "bf16x3_if16" is a mode of the fused inference path; anything else it meets runs split-bf16
        prec = self.precision if (in_place or self.precision not in _FUSED_ONLY) else _lib.PREC_BF16X3
        if (cls is None and training and len(batch) >= 14 and batch[7].dtype == torch.bfloat16 and batch[7].dim() == 4
                and batch[7].is_cuda and batch[7].shape[0] > 0):
            # a training step on bf16-stored token blocks: pooled in place by the library - half
            # the bytes, no widened copy of the 197 KB per candidate - then the pooled-ahead form below
            etf = batch[7]
            batch = list(batch[:7]) + [_pool_tokens(etf, batch[8]), torch.zeros(etf.shape[0], dtype=torch.int64, device=etf.device)] \
                + list(batch[9:])
            cls = etf[:, :, 0, :]
        if cls is not None:
            # pooled-ahead batch: the layer-by-layer entry points (the fused path folds the pooling into its one pass)
            call = _Call(self.cfg, batch, _lib.PREC_BF16X3 if prec in _FUSED_ONLY else prec, entity_text_cls=cls)
            if call.B == 0:
                return torch.zeros(0, call.N, dtype=torch.float32, device=call.device)
            return self._score(call, None, training, *params)
        call = _Call(self.cfg, batch, prec, keep_bf16=in_place)
        if ((call.cfg.feature_dtype != _lib.FEAT_F32 or prec in _FUSED_ONLY)
                and _lib.load().drin_fused_supported(C.byref(call.cfg)) != _lib.OK):
            call = _Call(self.cfg, batch, _lib.PREC_BF16X3 if prec in _FUSED_ONLY else prec)
        if call.B == 0:
            return torch.zeros(0, call.N, dtype=torch.float32, device=call.device)
        return self._score(call, self._prepared, training, *params)

    def wait_for_parameters(self) -> None:
        """Make the current stream wait for an optimiser update `train.OverlappedStep` left running on its side stream (no-op
        otherwise).  Every path that reads the parameters calls it - or, the layer-by-layer forward, hands the event to
        `drin_forward_staged` so that its parameter-free head runs under the update."""
        ev, self._params_ready = self._params_ready, None
        if ev is not None:
            torch.cuda.current_stream(next(self.parameters()).device).wait_event(ev)

    def _score(self, call: _Call, prepared, training: bool, *params):
        call.owner = self
        ev = self._params_ready
        if ev is not None:
            if training or prepared is None:                 # the layer-by-layer entry point: staged
                call.params_ready, self._params_ready = ev, None
            else:
                self.wait_for_parameters()
        try:
            return _DrinScore.apply(call, prepared, training, *params)
        except BaseException:
            # a staged call that failed before the library enqueued its wait (validation / workspace error) must not lose the
            # ordering behind the update still running on the side stream: the current stream waits here (harmless if the
            # library already did), so that whatever reads the parameters next reads them after Adam has written them
            if ev is not None:
                torch.cuda.current_stream(call.device).wait_event(ev)
            raise

    def _clamped(self, batch: "IndexedBatch") -> "IndexedBatch":
        """The batch with its candidate rows clamped into the tables and any clamped row reported in the status words: the torch
        gathers of the paths off the fused kernels must never see a row >= E (a device-side assert), and a negative row wraps
        silently there."""
        cand = batch.candidates
        safe = cand.clamp(0, batch.table.num_entities - 1)
        self._status_words(cand.device)[0:1].bitwise_or_((safe != cand).any().view(1).to(torch.int32))
        return IndexedBatch(batch.mention, batch.table, safe, batch.miet_similarity, batch.mtei_similarity)

    def _indexed_training_call(self, batch: "IndexedBatch", planes: bool) -> Optional[_Call]:
        """The table form of `drin_forward` / `drin_backward` (`drin_batch.entity_index` over tables pooled ahead of time),
        when the library builds it for this geometry (`indexed_supported` in csrc/api.hip); None: gather the rows."""
        t, cand = batch.table, batch.candidates
        D, R = self.cfg.bert_embed_dim, self.cfg.resnet_embed_dim
        ok = (planes and self.cfg.gcn_edge_feature != "vector"
              and t.text.dtype == torch.float32 and t.image.dtype == torch.float32 and t.object.dtype == torch.float32
              and cand.numel() >= 1024 and cand.shape[0] <= 65535 and D % 32 == 0 and R % 32 == 0 and D >= 128
              and 128 <= R <= 2048 and t.object.shape[1] == 1 and t.image.dim() in (2, 3) and t.object.dim() in (3, 4)
              and (t.image.dim() == 2 or t.image.shape[1] == 1) and (t.object.dim() == 3 or t.object.shape[2] == 1))
        if not ok:
            return None
        pooled, cls = t.pooled_text(self.cfg)
        dummy = torch.zeros(cand.shape[0], dtype=torch.int64, device=cand.device)
        seq = batch.mention + [pooled, dummy, t.image, t.object, t.object_score, batch.miet_similarity, batch.mtei_similarity]
        # the layer-by-layer kernels trust the index (the stream kernel of the fused path clamps it): clamp here, so that a
        # bad candidate row can never become an out-of-bounds read on the device - and report it like the kernels do
        # (four small launches, no synchronisation: word 0 of the status words becomes 1, the value words stay 0)
        prec = _lib.PREC_BF16X3 if self.precision in _FUSED_ONLY else self.precision
        safe = cand.clamp(0, t.num_entities - 1)
        self._status_words(cand.device)[0:1].bitwise_or_((safe != cand).any().view(1).to(torch.int32))
        return _Call(self.cfg, seq, prec, entity_index=safe, entity_text_cls=cls)

    @torch.no_grad()
    def _forward_cached(self, call: _Call, table: EntityTable, params) -> torch.Tensor:
        """Table-form inference from the per-entity cache (`drin_forward_cached`)."""
        lib = _lib.load()
        if call.B == 0:
            return torch.zeros(0, call.N, dtype=torch.float32, device=call.device)
        self.wait_for_parameters()
        det = tuple(p.detach().contiguous() for p in params)
        pc = _lib.DrinParamsC()
        _fill_params(pc, det, call.per_layer)
        pbuf = self._prepared.get(call, params, pc)
        cache = table._get_cache(call, (id(self._prepared), self._prepared.generation), pc, pbuf)
        n = lib.drin_cached_workspace_bytes(C.byref(call.cfg))
        ws = torch.empty(max(n, 16), dtype=torch.uint8, device=call.device)
        scores = torch.empty(call.B, call.N, dtype=torch.float32, device=call.device)
        stream = torch.cuda.current_stream(call.device).cuda_stream
        _lib.check(lib.drin_forward_cached(C.byref(call.cfg), C.byref(call.batch), C.byref(pc), pbuf.data_ptr(),
                                           cache.data_ptr(), ws.data_ptr(), ws.numel(), scores.data_ptr(), stream))
        return scores

    @torch.no_grad()
    def forward_traced(self, batch: Sequence[torch.Tensor]) -> Dict[str, torch.Tensor]:
        """Scores plus every stage's vertices and edges (tests / debugging)."""
        lib = _lib.load()
        self.wait_for_parameters()
        call = _Call(self.cfg, batch, _lib.PREC_BF16X3 if self.precision in _FUSED_ONLY else self.precision)
        params = tuple(p.detach().contiguous() for p in _param_list(self))
        pc = _lib.DrinParamsC()
        _fill_params(pc, params, call.per_layer)
        B, N, D, nl, dev = call.B, call.N, call.D, self.cfg.num_gcn_layers, call.device
        edge_shape = (4, B, N, D) if call.cfg.vector_edges else (4, B, N)
        tr = _lib.DrinTraceC()
        out: Dict[str, torch.Tensor] = {}
        for l in range(nl + 1):
            for field, key, shape in (("mention_text_vertex", f"mt{l}", (B, D)), ("mention_image_vertex", f"mi{l}", (B, D)),
                                      ("entity_text_vertex", f"et{l}", (B, N, D)), ("entity_image_vertex", f"ei{l}", (B, N, D)),
                                      ("edges", f"edges{l}", edge_shape)):
                t = torch.full(shape, float("nan"), dtype=torch.float32, device=dev)
                out[key] = t
                getattr(tr, field)[l] = t.data_ptr()
        ws = call.workspace(False)
        scores = torch.empty(B, N, dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(lib.drin_forward(C.byref(call.cfg), C.byref(call.batch), C.byref(pc), ws.data_ptr(), ws.numel(),
                                    scores.data_ptr(), 0, C.byref(tr), stream))
        out["scores"] = scores
        return out
