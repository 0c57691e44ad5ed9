"""CPU oracle for the DRIN scoring path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A vectorised CPU restatement (plain torch tensor ops on the host) of the reference's
`Model.forward` and of the caller-side loss/metric, each function citing the reference
file:line it follows (paths relative to /root/reference).  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module; the
product (`drin_amd/`) never does and fails loudly when its HIP library is missing.

Parity pin: `oracle/gen_golden.py` imports the *unmodified* reference in the build
container and writes `tests/golden/*.npz`; `tests/test_oracle_golden.py` checks this
restatement against those vectors (<= 2e-6 abs on scores).  `TripletLoss` / `TopkAccuracy`
(`common/utils.py`) cannot be imported there (torchmetrics / lightning are not installed), so
they are restated from the source text and pinned by hand-computed known-answer cases.

All arithmetic is fp32 by default like the reference; pass `dtype=torch.float64` to get a
higher-precision yard-stick (used to show whose rounding error is whose).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

# vertex / edge topology of drin/model.py:105,107 (u <- [(edge, neighbour)], edge -> (u, v))
VERTEX_GRAPH = [[[0, 2], [1, 3]], [[2, 2], [3, 3]], [[0, 0], [2, 1]], [[1, 0], [3, 1]]]
EDGE_GRAPH = [[0, 2], [0, 3], [1, 2], [1, 3]]


def span_mean(seq: torch.Tensor, begin: torch.Tensor, end: torch.Tensor) -> torch.Tensor:
    """`Avg.avg` (baselines/ghmfc.py:54-60): mean of seq[i, begin[i]:end[i]] over tokens.

    Python slice semantics are kept (end clipped to L); an empty span is 0/0 = NaN exactly
    like `torch.mean` of an empty slice.
    """
    L = seq.shape[1]
    idx = torch.arange(L)
    m = (idx[None, :] >= begin[:, None]) & (idx[None, :] < end[:, None])
    total = torch.where(m[..., None], seq, torch.zeros((), dtype=seq.dtype)).sum(1)
    return total / m.sum(1).to(seq.dtype)[:, None]


def entity_token_mean(feat: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """WikiMEL entity pooling (baselines/ghmfc.py:245-249 with AvgPool(dim=0) :38-44,257-258):
    x[b,n] = mean(feat[b,n,1:ntok-1]), ntok = sum(mask[b,n]).  Slice semantics kept:
    ntok in {1,2} -> empty -> NaN; ntok == 0 -> stop index -1 -> tokens 1..T-2."""
    T = feat.shape[2]
    ntok = mask.sum(-1)
    stop = ntok - 1
    stop = torch.where(stop < 0, stop + T, stop).clamp(min=0, max=T)
    idx = torch.arange(T)
    m = (idx[None, None, :] >= 1) & (idx[None, None, :] < stop[..., None])
    total = torch.where(m[..., None], feat, torch.zeros((), dtype=feat.dtype)).sum(2)
    return total / m.sum(-1).to(feat.dtype)[..., None]


def cosine(x: torch.Tensor, y: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    """`nn.CosineSimilarity(dim=-1, eps=1e-8)` as torch>=2 computes it:
    each operand is divided by its own norm clamped at eps, then the products are summed."""
    xn = x / torch.linalg.vector_norm(x, dim=-1, keepdim=True).clamp_min(eps)
    yn = y / torch.linalg.vector_norm(y, dim=-1, keepdim=True).clamp_min(eps)
    return (xn * yn).sum(-1)


def vertex_encoder(p: Dict[str, torch.Tensor], batch: Sequence[torch.Tensor], token_level: bool) -> List[torch.Tensor]:
    """`VertexEncoder.forward` (drin/model.py:26-46) -> [mt, mi, et, ei]."""
    (mtf, _mask, start, end, mimg, _mobj, _ms, etf, emask, eimg, _eobj, _es, _miet, _mtei) = batch[:14]
    pre = "vertex_encoder."
    # mention text: AvgLinear (baselines/ghmfc.py:63-69, wired :163-165)
    mt = F.linear(span_mean(mtf, start, end), p[pre + "mention_text_encoder.final_layer.linear.weight"],
                  p[pre + "mention_text_encoder.final_layer.linear.bias"])
    # entity text: offline branch (baselines/ghmfc.py:237-251)
    x_et = entity_token_mean(etf, emask) if token_level else etf
    et = F.linear(x_et, p[pre + "entity_text_encoder.final_layer.weight"], p[pre + "entity_text_encoder.final_layer.bias"])
    # mention image: mean over regions then Linear (drin/model.py:41-42)
    mi = F.linear(mimg.mean(-2), p[pre + "mention_image_linear.weight"], p[pre + "mention_image_linear.bias"])
    # entity image (drin/model.py:43-45)
    if eimg.dim() == 4:
        eimg = eimg.mean(-2)
    ei = F.linear(eimg, p[pre + "entity_image_linear.weight"], p[pre + "entity_image_linear.bias"])
    return [mt, mi, et, ei]


def edge_encoder(batch: Sequence[torch.Tensor], cos_eps: float = 1e-8, miei_eps: float = 1e-9) -> List[torch.Tensor]:
    """`EdgeEncoder.forward` (drin/model.py:60-94) -> (mtet, miei), both [B, N]."""
    (mtf, _mask, start, end, _mimg, mobj, ms, etf, _emask, _eimg, eobj, es, _miet, _mtei) = batch[:14]
    m = span_mean(mtf, start, end)                                   # :71
    et_raw = etf[:, :, 0] if etf.dim() == 4 else etf                   # :73-75  entity CLS / pooler
    mtet = cosine(m[:, None, :], et_raw, cos_eps)                      # :76
    if mobj.dim() == 4:                                                # :78-79
        mobj = mobj.mean(-2)
    if eobj.dim() == 5:                                                # :82-83
        eobj = eobj.mean(-2)
    sim = cosine(mobj[:, None, :, None, :], eobj[:, :, None, :, :], cos_eps)   # [B,N,Km,Ke]  :88
    w = ms[:, None, :, None] * es[:, :, None, :]                               # :89
    # the reference accumulates i-major, j-minor (:86-91); keep that order for the sums
    similarity = torch.zeros(sim.shape[:2], dtype=sim.dtype)
    scores = torch.zeros(sim.shape[:2], dtype=sim.dtype)
    for i in range(sim.shape[2]):
        for j in range(sim.shape[3]):
            similarity = similarity + sim[:, :, i, j] * w[:, :, i, j]
            scores = scores + w[:, :, i, j]
    miei = similarity / (scores + miei_eps)                            # :92
    return [mtet, miei]


def gcn_layer(p: Dict[str, torch.Tensor], l: int, vertexes: List[torch.Tensor], edges: List[torch.Tensor],
              edge_enabled: Sequence[float], dynamic: bool, ln_eps: float = 1e-5, vector: bool = False,
              vertex_activation: str = "gelu", edge_activation: str = "sigmoid"):
    """`GCNLayer.forward` (drin/model.py:121-153).  `vector`: gcn_edge_feature == "vector" - edges are
    [B, N, D] (model.py:140-141 skipped), w_u / w_v map to D/2 and are concatenated (:151-152), w_m is a Linear (:112)."""
    pre = f"gcn_layers.{l}."
    act_v, act_e = getattr(F, vertex_activation), getattr(F, edge_activation)   # model.py:117-118 (args.py:35-36)
    D = vertexes[0].shape[-1]
    edges = [e * m for e, m in zip(edges, edge_enabled)]               # :122
    new_v = []
    for u, nb in zip(vertexes, VERTEX_GRAPH):                          # :124-129
        acc = torch.zeros_like(u)
        for ei, vi in nb:
            e, v = (edges[ei] if vector else edges[ei][..., None]), vertexes[vi]
            if v.dim() == 3:
                acc = acc + (e * v).mean(1)                            # mention <- entity  :143-144
            else:
                acc = acc + e * v[:, None, :]                          # entity <- mention  :146
        h = F.linear(acc + u, p[pre + "w_h.weight"], p[pre + "w_h.bias"])
        h = F.layer_norm(h, (D,), p[pre + "layer_norm.weight"], p[pre + "layer_norm.bias"], ln_eps)
        new_v.append(act_v(h))                                         # :128; default exact-erf gelu (args.py:35)
    if dynamic:                                                        # :130-134
        new_e = []
        for e, (ui, vi) in zip(edges, EDGE_GRAPH):
            fu = F.linear(vertexes[ui], p[pre + "w_u.weight"], p[pre + "w_u.bias"])
            fv = F.linear(vertexes[vi], p[pre + "w_v.weight"], p[pre + "w_v.bias"])
            if vector:                                                 # :151-152 then w_m Linear (:112,133)
                cat = torch.cat([fu[:, None, :].expand(-1, fv.shape[1], -1), fv], dim=-1)
                new_e.append(act_e(F.linear(cat + e, p[pre + "w_m.weight"], p[pre + "w_m.bias"])))
                continue
            new_e.append(act_e((fu[:, None, :] * fv).mean(-1) + e))    # :148-153,133; w_m = Identity :112; default sigmoid
    else:
        new_e = edges                                                  # :136
    return new_v, new_e


def forward(p: Dict[str, torch.Tensor], batch: Sequence[torch.Tensor], *, token_level: Optional[bool] = None,
            num_layers: int = 2, edge_enabled: Sequence[float] = (1, 1, 1, 1), dynamic: bool = True,
            dtype: torch.dtype = torch.float32, trace: Optional[dict] = None, vector: bool = False,
            vertex_activation: str = "gelu", edge_activation: str = "sigmoid") -> torch.Tensor:
    """`Model.forward` (drin/model.py:164-209) -> scores [B, N]."""
    batch = [t.to(dtype) if t.is_floating_point() else t for t in batch[:14]]
    p = {k: v.to(dtype) for k, v in p.items()}
    if token_level is None:
        token_level = batch[7].dim() == 4
    vertexes = vertex_encoder(p, batch, token_level)                   # :181-190
    mtet, miei = edge_encoder(batch)                                   # :191-200
    edges = [mtet, batch[13] / 100, batch[12] / 100, miei]             # :201-204  (tt, ti, it, ii)
    if vector:                                                         # :202
        D = vertexes[0].shape[-1]
        edges = [e.unsqueeze(-1).expand(-1, -1, D) for e in edges]
    if trace is not None:
        trace["vertex0"] = [v.clone() for v in vertexes]
        trace["edge0"] = [e.clone() for e in edges]
    for l in range(num_layers):                                        # :205-206
        vertexes, edges = gcn_layer(p, l, vertexes, edges, edge_enabled, dynamic, vector=vector,
                                    vertex_activation=vertex_activation, edge_activation=edge_activation)
        if trace is not None:
            trace[f"vertex{l + 1}"] = [v.clone() for v in vertexes]
            trace[f"edge{l + 1}"] = [e.clone() for e in edges]
    return cosine(vertexes[0][:, None, :], vertexes[2])                # :207-209


def config_kwargs(cfg) -> dict:
    """The keyword arguments of `forward` a `drin_amd.config.DrinConfig` (the names of common/args.py:24-36) stands for."""
    return dict(num_layers=cfg.num_gcn_layers, edge_enabled=cfg.gcn_edge_enabled, dynamic=cfg.gcn_edge_type == "dynamic",
                vector=cfg.gcn_edge_feature == "vector", vertex_activation=cfg.gcn_vertex_activation,
                edge_activation=cfg.gcn_edge_activation)


def triplet_loss(y_true: torch.Tensor, y_pred: torch.Tensor, margin: float = 0.25) -> torch.Tensor:
    """`TripletLoss.__call__` (common/utils.py:35-43).  Each mention's positive distance is
    compared with the WHOLE batch's [B, N-1] matrix (:42), gold column included."""
    if y_pred.shape[1] != y_true.shape[1]:
        y_pred = y_pred[:, :-1]
    y_pred = -y_pred
    pos = (y_pred * y_true).sum(-1)
    per = torch.clamp(pos[:, None, None] - y_pred[None, :, :] + margin, min=0).mean((1, 2))
    return per.sum() / y_true.shape[0]


def topk_counts(y_pred: torch.Tensor, y_true: torch.Tensor, k: int):
    """`TopkAccuracy.update` (common/utils.py:60-66): (correct, total) increments; ties with
    the k-th largest score count as correct."""
    if y_pred.shape[1] != y_true.shape[1]:
        y_pred = y_pred[:, :-1]
    lb = torch.topk(y_pred, k)[0][:, -1:]
    return int((y_true * (y_pred >= lb)).sum()), int(y_true.shape[0])


def reference_style_forward(p, batch, **kw):
    """Same result as `forward`, but keeping the reference's Python loops over mentions and
    candidates (baselines/ghmfc.py:58-59,246-249) to expose its own cost profile in the
    cpu_baseline leg of bench.py."""
    batch = list(batch[:14])
    mtf, start, end = batch[0], batch[2], batch[3]
    res = torch.empty(mtf.shape[0], mtf.shape[-1])
    for i in range(mtf.shape[0]):
        res[i] = torch.mean(mtf[i, start[i]: end[i]], dim=0)
    if batch[7].dim() == 4:
        etf, emask = batch[7], batch[8]
        enc = torch.empty(etf.shape[0], etf.shape[1], etf.shape[-1])
        for i in range(etf.shape[0]):
            ntok = emask[i].sum(-1)
            for j in range(etf.shape[1]):
                enc[i, j] = torch.mean(etf[i, j, 1: ntok[j] - 1, :], dim=0)
    return forward(p, batch, **kw)


def flops_per_pair(D: int = 768, R: int = 2048, layers: int = 2) -> float:
    """Reference-faithful forward FLOPs per (mention, candidate) pair (SURVEY.md §8a tail)."""
    return 2.0 * D * D + 2.0 * R * D + layers * (2 * 2.0 * D * D + 2 * 2.0 * D * D)


assert math.isclose(flops_per_pair(), 13.76e6, rel_tol=1e-3)
