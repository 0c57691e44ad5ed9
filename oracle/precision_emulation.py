"""Test infrastructure (never imported by the product): what rounding the operands of ONE pair-sized contraction of the folded
scoring path to fewer bits costs on the final scores - the evidence behind precision BY CONTRACTION (`precision="bf16x3_if16"`, DESIGN.md
section 4.3; round 4's bf16 form of it, `bf16x3_i1`, was removed in round 5) and behind the mixed-f16 cache rows (section 6.1).

The folded inference path (csrc/fused_forward.hip) leaves three contractions per (mention, candidate) pair:
    image  x_i C_i^T,  C_i = W_h1 W_ei   (D x R: 57 % of the path's FLOPs)  -> pre-LayerNorm value of the layer-1 entity IMAGE vertex
    text   x_t C_t^T,  C_t = W_h1 W_et   (D x D)                            -> pre-LayerNorm value of the layer-1 entity TEXT vertex
    wh2    et' W_h2^T                    (D x D)                            -> pre-LayerNorm value of the layer-2 entity text vertex
`ei'` reaches the score only through `mean_n(ti' ei')` in the layer-2 mention-text vertex (`drin/model.py:124-129,143-144`,
vertex graph `:105`): an average over the N candidates, so its rounding noise is divided by ~sqrt(N) before it meets the score;
`et'` and `et''` enter the final cosine (`model.py:207-209`) directly.

`scores_with_rounded_contraction` replays `oracle.drin_oracle.forward` in fp64 (default geometry: two layers, scalar dynamic
edges) and adds, to the pre-LayerNorm value the chosen contraction produces, exactly the error of evaluating that contraction on
bf16-rounded operands: `mode` "one" = both operands rounded (one MFMA pass), "two_a" = activations split hi + lo, weights
rounded (two passes), "two_w" = activations rounded, weights split.  Everything else stays exact, so the score difference
against the plain fp64 forward is that contraction's contribution alone."""
from __future__ import annotations

from typing import Dict, Sequence

import torch
import torch.nn.functional as F

from . import drin_oracle as O


def _r(x: torch.Tensor) -> torch.Tensor:            # round to bf16 (nearest even), back in fp64
    return x.to(torch.float32).to(torch.bfloat16).to(torch.float64)


def _rounded_product_error(a: torch.Tensor, w: torch.Tensor, mode: str) -> torch.Tensor:
    """(what the MFMA passes of `mode` compute) - a w^T, in fp64; a [..., K], w [D, K]."""
    exact = a @ w.T
    if mode == "one":
        return _r(a) @ _r(w).T - exact
    if mode == "two_a":                              # (hi + lo) of the activations against rounded weights
        hi = _r(a)
        return (hi + _r(a - hi)) @ _r(w).T - exact
    if mode == "two_w":
        hi = _r(w)
        return _r(a) @ (hi + _r(w - hi)).T - exact
    if mode == "one_f16":                            # one fp16 pass: 11-bit operands; activation rows scaled by a power of two
        # (max |row| into [0.5, 1): exact, keeps every row inside fp16's range whatever the features' scale), weights as they are
        sc = torch.exp2(torch.ceil(torch.log2(a.abs().amax(-1, keepdim=True).clamp_min(1e-30))))
        h = lambda x: x.to(torch.float32).to(torch.float16).to(torch.float64)   # noqa: E731
        return (h(a / sc) @ h(w).T) * sc - exact
    if mode == "x3":                                 # the default split product: hi hi + hi lo + lo hi (lo lo dropped)
        ah, wh = _r(a), _r(w)
        al, wl = _r(a - ah), _r(w - wh)
        return ah @ wh.T + ah @ wl.T + al @ wh.T - exact
    raise ValueError(mode)


def scores_with_rounded_contraction(p: Dict[str, torch.Tensor], batch: Sequence[torch.Tensor], which: str = "none",
                                    mode: str = "one") -> torch.Tensor:
    f64 = torch.float64
    batch = [t.to(f64) if t.is_floating_point() else t for t in batch[:14]]
    p = {k: v.to(f64) for k, v in p.items()}
    token_level = batch[7].dim() == 4
    v = O.vertex_encoder(p, batch, token_level)                          # [mt, mi, et, ei]
    mtet, miei = O.edge_encoder(batch)
    edges = [mtet, batch[13] / 100, batch[12] / 100, miei]
    x_t = O.entity_token_mean(batch[7], batch[8]) if token_level else batch[7]
    x_i = batch[9].mean(-2) if batch[9].dim() == 4 else batch[9]
    D = v[0].shape[-1]
    for l in range(2):
        pre = f"gcn_layers.{l}."
        W, b = p[pre + "w_h.weight"], p[pre + "w_h.bias"]
        new_v = []
        for idx, (u, nb) in enumerate(zip(v, O.VERTEX_GRAPH)):
            acc = torch.zeros_like(u)
            for ei, vi in nb:
                e, nv = edges[ei][..., None], v[vi]
                acc = acc + ((e * nv).mean(1) if nv.dim() == 3 else e * nv[:, None, :])
            h = F.linear(acc + u, W, b)
            if l == 0 and idx == 3 and which == "image":
                h = h + _rounded_product_error(x_i, W @ p["vertex_encoder.entity_image_linear.weight"], mode)
            if l == 0 and idx == 2 and which == "text":
                h = h + _rounded_product_error(x_t, W @ p["vertex_encoder.entity_text_encoder.final_layer.weight"], mode)
            if l == 1 and idx == 2 and which == "wh2":
                h = h + _rounded_product_error(v[2], W, mode)
            h = F.layer_norm(h, (D,), p[pre + "layer_norm.weight"], p[pre + "layer_norm.bias"], 1e-5)
            new_v.append(F.gelu(h))
        new_e = []
        for e, (ui, vi) in zip(edges, O.EDGE_GRAPH):
            fu = F.linear(v[ui], p[pre + "w_u.weight"], p[pre + "w_u.bias"])
            fv = F.linear(v[vi], p[pre + "w_v.weight"], p[pre + "w_v.bias"])
            new_e.append(torch.sigmoid((fu[:, None, :] * fv).mean(-1) + e))
        v, edges = new_v, new_e
    return O.cosine(v[0][:, None, :], v[2])


def contraction_errors(p, batch, cases=(("image", "one"), ("image", "two_a"), ("image", "two_w"), ("text", "one"), ("wh2", "one"),
                                        ("image", "x3"))) -> Dict[str, dict]:
    exact = scores_with_rounded_contraction(p, batch)
    out = {}
    for which, mode in cases:
        s = scores_with_rounded_contraction(p, batch, which, mode)
        d = (s - exact).abs()
        flips = int((s[:, :-1].argmax(1) != exact[:, :-1].argmax(1)).sum())
        out[f"{which}:{mode}"] = {"max": float(d.max()), "rms": float((d ** 2).mean().sqrt()), "top1_flips": flips, "scores": int(d.numel())}
    return out


# ---- precision by STORAGE: the per-entity cache's DRIN_CACHE_MIXED_F16 rows (csrc/entity_cache.hip) -------------------------
CACHE_FIELDS = ("h_t", "h_i", "fv_t", "fv_i", "chat", "ohat")
MIXED_F16_FIELDS = ("fv_t", "fv_i", "ohat")                 # what DRIN_CACHE_MIXED_F16 stores as scaled fp16


def _scaled_f16(x: torch.Tensor) -> torch.Tensor:
    """fp16 under one power-of-two scale per row (`cache_field_scale`: 2^ceil(log2 max|row|)), back in fp64."""
    sc = torch.exp2(torch.ceil(torch.log2(x.abs().amax(-1, keepdim=True).clamp_min(1e-30))))
    return (x / sc).to(torch.float32).to(torch.float16).to(torch.float64) * sc


def scores_with_rounded_cache_fields(p: Dict[str, torch.Tensor], batch: Sequence[torch.Tensor], fields: Sequence[str] = (),
                                     rnd=_scaled_f16) -> torch.Tensor:
    """`oracle.drin_oracle.forward` replayed in fp64 with the named fields of the per-entity cache row (`drin_hip.h`,
    drin_cache_format) passed through `rnd` where the cached path reads them, everything else exact:
      h_t  = x_t (W_h1 W_et)^T, h_i = x_i (W_h1 W_ei)^T: the layer-1 entity contractions (`model.py:128,146`); the cached path
             also forms the layer-1 MENTION aggregates from them (mean_n(e h), `model.py:143-144` through W_h1);
      fv_t = W_v1(et0), fv_i = W_v1(ei0): the edge-update operands (`model.py:148-153`);
      chat = the normalised CLS / pooler row of the text-text edge (`model.py:71-76`);
      ohat = sum_j es_j obj_j / |obj_j|: the entity side of the image-image edge (`model.py:84-92`)."""
    f64 = torch.float64
    batch = [t.to(f64) if t.is_floating_point() else t for t in batch[:14]]
    p = {k: v.to(f64) for k, v in p.items()}
    (mtf, _mask, start, end, _mimg, mobj, ms, etf, emask, eimg, eobj, es, miet, mtei) = batch
    token_level = etf.dim() == 4
    v = O.vertex_encoder(p, batch, token_level)
    unit = lambda x: x / torch.linalg.vector_norm(x, dim=-1, keepdim=True).clamp_min(1e-8)   # noqa: E731
    chat = unit(etf[:, :, 0] if token_level else etf)
    if "chat" in fields:
        chat = rnd(chat)
    mtet = (unit(O.span_mean(mtf, start, end))[:, None, :] * chat).sum(-1)
    if mobj.dim() == 4:
        mobj = mobj.mean(-2)
    if eobj.dim() == 5:
        eobj = eobj.mean(-2)
    ohat = (es[..., None] * unit(eobj)).sum(-2)
    if "ohat" in fields:
        ohat = rnd(ohat)
    mu = (ms[..., None] * unit(mobj)).sum(-2)
    miei = (mu[:, None, :] * ohat).sum(-1) / (ms.sum(-1)[:, None] * es.sum(-1) + 1e-9)
    edges = [mtet, mtei / 100, miet / 100, miei]
    x_t = O.entity_token_mean(etf, emask) if token_level else etf
    x_i = eimg.mean(-2) if eimg.dim() == 4 else eimg
    D = v[0].shape[-1]
    for l in range(2):
        pre = f"gcn_layers.{l}."
        W, b = p[pre + "w_h.weight"], p[pre + "w_h.bias"]
        d_h = {}
        if l == 0:
            for name, x, w in (("h_t", x_t, "vertex_encoder.entity_text_encoder.final_layer.weight"),
                               ("h_i", x_i, "vertex_encoder.entity_image_linear.weight")):
                if name in fields:
                    h = x @ (W @ p[w]).T
                    d_h[name] = rnd(h) - h
        new_v = []
        for idx, (u, nb) in enumerate(zip(v, O.VERTEX_GRAPH)):
            acc = torch.zeros_like(u)
            for ei, vi in nb:
                e, nv = edges[ei][..., None], v[vi]
                acc = acc + ((e * nv).mean(1) if nv.dim() == 3 else e * nv[:, None, :])
            h = F.linear(acc + u, W, b)
            if l == 0 and idx == 2 and "h_t" in d_h:
                h = h + d_h["h_t"]
            if l == 0 and idx == 3 and "h_i" in d_h:
                h = h + d_h["h_i"]
            if l == 0 and idx < 2:
                for ei, vi in nb:
                    name = "h_t" if vi == 2 else "h_i"
                    if name in d_h:
                        h = h + (edges[ei][..., None] * d_h[name]).mean(1)
            h = F.layer_norm(h, (D,), p[pre + "layer_norm.weight"], p[pre + "layer_norm.bias"], 1e-5)
            new_v.append(F.gelu(h))
        new_e = []
        for e, (ui, vi) in zip(edges, O.EDGE_GRAPH):
            fu = F.linear(v[ui], p[pre + "w_u.weight"], p[pre + "w_u.bias"])
            fv = F.linear(v[vi], p[pre + "w_v.weight"], p[pre + "w_v.bias"])
            if l == 0 and (("fv_t" in fields and vi == 2) or ("fv_i" in fields and vi == 3)):
                fv = rnd(fv)
            new_e.append(torch.sigmoid((fu[:, None, :] * fv).mean(-1) + e))
        v, edges = new_v, new_e
    return O.cosine(v[0][:, None, :], v[2])


def cache_field_errors(p, batch, cases=(("h_i",), ("fv_t",), ("fv_i",), ("ohat",), MIXED_F16_FIELDS, ("chat",), ("h_t",))) -> Dict[str, dict]:
    exact = scores_with_rounded_cache_fields(p, batch)
    out = {}
    for fields in cases:
        s = scores_with_rounded_cache_fields(p, batch, fields)
        d = (s - exact).abs()
        flips = int((s[:, :-1].argmax(1) != exact[:, :-1].argmax(1)).sum())
        out["+".join(fields)] = {"max": float(d.max()), "rms": float((d ** 2).mean().sqrt()), "top1_flips": flips, "scores": int(d.numel())}
    return out
