"""Generate tests/golden/*.npz by running the UNMODIFIED reference forward on CPU.

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference); the GPU box
only ever sees the committed fixtures.  Recipe: put /root/reference on
sys.path, set `common.args.use_device = "cpu"` before importing `drin.model`, then patch
the star-imported module globals of `drin.model` / `baselines.ghmfc` per case (the classes
read them at call time).  Inputs and weights are NOT stored: both sides regenerate them
from `drin_amd.synth` seeds.  Stored: scores, every stage's edges and mention vertices,
entity vertices (full for tiny cases, mention 0 only for full-width cases), gradients of a
fixed linear functional of the scores and of the triplet loss, and one Adam step.

`TripletLoss` lives in `common/utils.py`, whose module import fails on `torchmetrics`; the
class itself is pure torch, so its source is taken from that file with `ast` and executed
as is (no stand-in for torchmetrics is written).  `TopkAccuracy` subclasses
`torchmetrics.Metric`, so the CLASS cannot be built - but its `update` and `compute` bodies
(`common/utils.py:60-69`) are pure torch on `self.top_k / self.correct / self.total`: the two
function definitions are lifted with `ast` the same way and called, unmodified, on a plain
namespace carrying those three attributes (`tests/golden/topk.npz`).

usage:  python oracle/gen_golden.py            (writes tests/golden/)
"""
from __future__ import annotations

import ast
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

import common.args as ref_args  # noqa: E402

ref_args.use_device = "cpu"
from baselines import ghmfc as ref_ghmfc  # noqa: E402
from drin import model as ref_model  # noqa: E402

from drin_amd import synth  # noqa: E402
from drin_amd.config import DrinConfig  # noqa: E402


def _reference_triplet_loss():
    src = open(os.path.join(REF, "common/utils.py")).read()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "TripletLoss")
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[cls], type_ignores=[]), "common/utils.py", "exec"), ns)
    return ns["TripletLoss"]


RefTripletLoss = _reference_triplet_loss()


def _reference_topk_methods():
    """`(update, compute)` of the reference's `TopkAccuracy` (`common/utils.py:60-69`) as plain functions of `self`."""
    tree = ast.parse(open(os.path.join(REF, "common/utils.py")).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "TopkAccuracy")
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in ("update", "compute")]
    assert [f.name for f in fns] == ["update", "compute"], [f.name for f in fns]
    ns = {"torch": torch, "Tensor": torch.Tensor}
    exec(compile(ast.Module(body=fns, type_ignores=[]), "common/utils.py", "exec"), ns)
    return ns["update"], ns["compute"]


ref_topk_update, ref_topk_compute = _reference_topk_methods()


def _patch(cfg: DrinConfig) -> None:
    vals = dict(
        dataset_name=cfg.dataset_name,
        num_candidates_data=cfg.num_candidates_data,
        num_candidates_model=cfg.num_candidates_model,
        bert_embed_dim=cfg.bert_embed_dim,
        resnet_embed_dim=cfg.resnet_embed_dim,
        gcn_embed_dim=cfg.gcn_embed_dim,
        mention_final_output_dim=cfg.gcn_embed_dim,
        entity_final_output_dim=cfg.gcn_embed_dim,
        num_gcn_layers=cfg.num_gcn_layers,
        gcn_edge_type=cfg.gcn_edge_type,
        gcn_edge_feature=cfg.gcn_edge_feature,
        gcn_edge_enabled=list(cfg.gcn_edge_enabled),
        gcn_vertex_activation=cfg.gcn_vertex_activation,
        gcn_edge_activation=cfg.gcn_edge_activation,
        use_device="cpu",
    )
    for mod in (ref_args, ref_model, ref_ghmfc):
        for k, v in vals.items():
            setattr(mod, k, v)


from oracle.cases import CASES, TINY, edge_case_batches as _edge_case_batches  # noqa: E402


def run_case(name):
    cfg, B, dseed, wseed, full_ev, with_grads = CASES[name]
    _patch(cfg)
    torch.manual_seed(0)
    model = ref_model.Model()
    sd = synth.make_state_dict(cfg, wseed)
    assert list(sd.keys()) == list(model.state_dict().keys()), "state_dict key/order contract broken"
    model.load_state_dict(sd)
    batch = _edge_case_batches(name, cfg, synth.make_batch(cfg, B, dseed))
    inputs, answer = batch[:-1], batch[-1]
    out = {}

    # stage capture through forward hooks on the reference's own modules
    cap = {}
    model.vertex_encoder.register_forward_hook(lambda m, i, o: cap.__setitem__("vertex0", [t.detach().clone() for t in o]))
    model.edge_encoder.register_forward_hook(lambda m, i, o: cap.__setitem__("edge_enc", [t.detach().clone() for t in o]))
    for l, layer in enumerate(model.gcn_layers):
        layer.register_forward_hook(
            lambda m, i, o, l=l: cap.__setitem__(f"layer{l + 1}", ([t.detach().clone() for t in o[0]], [t.detach().clone() for t in o[1]]))
        )
    scores = model(inputs)
    out["scores"] = scores.detach().numpy()
    out["mtet"], out["miei"] = (t.numpy() for t in cap["edge_enc"])
    stages = [("0", cap["vertex0"], None)] + [(str(l + 1),) + cap[f"layer{l + 1}"] for l in range(cfg.num_gcn_layers)]
    for tag, vs, es in stages:
        out[f"mt{tag}"], out[f"mi{tag}"] = vs[0].numpy(), vs[1].numpy()
        for nm, v in (("et", vs[2]), ("ei", vs[3])):
            out[f"{nm}{tag}"] = v.numpy() if full_ev else v[0].numpy()
            out[f"{nm}{tag}_sum"] = np.float64(v.double().sum().item())
            out[f"{nm}{tag}_l2"] = np.float64(v.double().norm().item())
        if es is not None:
            out[f"edges{tag}"] = torch.stack(es).numpy()

    if with_grads:
        # (1) gradient of a fixed linear functional of the scores: pins d(scores)/d(theta)
        g = np.random.Generator(np.random.Philox(key=[dseed, 99]))
        w = torch.from_numpy(g.standard_normal(size=tuple(scores.shape), dtype=np.float32))
        out["functional_weights_seed"] = np.int64(dseed)
        model.zero_grad()
        (scores * w).sum().backward()
        none_names = []
        for k, prm in model.named_parameters():
            if prm.grad is None:
                none_names.append(k)
                continue
            gr = prm.grad
            out[f"lin_grad/{k}"] = gr.numpy().copy() if full_ev else gr.flatten()[:16].numpy().copy()
            out[f"lin_grad_sum/{k}"] = np.float64(gr.double().sum().item())
            out[f"lin_grad_l2/{k}"] = np.float64(gr.double().norm().item())
        out["grad_none"] = np.array(none_names)
        # (2) reference TripletLoss (common/utils.py:26-43, executed from source) + its grads + one Adam step
        model.zero_grad()
        opt = torch.optim.Adam(model.parameters(), lr=cfg.learning_rate)   # train.py:55-56
        loss = RefTripletLoss(cfg.triplet_margin)(answer, model(inputs))
        out["triplet_loss"] = np.float32(loss.item())
        loss.backward()
        for k, prm in model.named_parameters():
            if prm.grad is not None:
                out[f"loss_grad_l2/{k}"] = np.float64(prm.grad.double().norm().item())
                if full_ev:
                    out[f"loss_grad/{k}"] = prm.grad.numpy().copy()
        opt.step()
        for k, prm in model.named_parameters():
            out[f"adam_l2/{k}"] = np.float64(prm.detach().double().norm().item())
            out[f"adam_head/{k}"] = prm.detach().flatten()[:8].numpy().copy()
    return out


def nan_case():
    """Empty span -> NaN row (baselines/ghmfc.py:59); other mentions are unaffected."""
    cfg = DrinConfig(**TINY)
    _patch(cfg)
    model = ref_model.Model()
    model.load_state_dict(synth.make_state_dict(cfg, 8))
    batch = synth.make_batch(cfg, 3, 11)
    batch[3][1] = batch[2][1]  # end == start
    return {"scores": model(batch[:-1]).detach().numpy()}


def init_order_case():
    """Same-seed construction (train.py:134-136): checksum of every tensor after manual_seed(0)."""
    cfg = DrinConfig()
    _patch(cfg)
    torch.manual_seed(0)
    model = ref_model.Model()
    out = {}
    for k, v in model.state_dict().items():
        out[f"l2/{k}"] = np.float64(v.double().norm().item())
        out[f"head/{k}"] = v.flatten()[:8].numpy().copy()
    return out


def triplet_cases():
    """The reference TripletLoss on a few seeded [B, N-1] / [B, N] pairs."""
    out = {}
    g = np.random.Generator(np.random.Philox(key=[5, 5]))
    for i, (B, N) in enumerate([(2, 4), (7, 11), (64, 101)]):
        yhat = torch.from_numpy((g.random(size=(B, N), dtype=np.float32) * 2 - 1))
        ans = g.integers(0, N, size=B)
        onehot = np.concatenate([np.eye(N - 1, dtype=np.uint8), np.zeros((1, N - 1), dtype=np.uint8)], 0)
        y = torch.from_numpy(onehot[ans])
        out[f"yhat{i}"], out[f"y{i}"] = yhat.numpy(), y.numpy()
        out[f"loss{i}"] = np.float32(RefTripletLoss(0.25)(y, yhat).item())
    return out


def topk_cases():
    """The reference `TopkAccuracy.update` / `.compute` (executed from source on a namespace with `top_k`, `correct`, `total`) on
    seeded `[B, N]` scores / `[B, N-1]` one-hot answers: plain rows, rows full of TIES (scores rounded to multiples of 0.25: the
    `>=` rule counts every candidate equal to the k-th largest), all-zero answer rows (the gold entity is not among the
    candidates: `answer == N - 1`), the last column dropped (`utils.py:61-62`), two updates accumulated, ks {1, 5, 10, 20, 50}."""
    from types import SimpleNamespace
    out = {}
    g = np.random.Generator(np.random.Philox(key=[6, 6]))
    shapes = [(2, 4), (7, 11), (64, 101), (33, 101)]
    out["ks"] = np.array([1, 5, 10, 20, 50], dtype=np.int64)
    for i, (B, N) in enumerate(shapes):
        for ties in (0, 1):
            yhat = g.random(size=(B, N), dtype=np.float32) * 2 - 1
            if ties:
                yhat = np.round(yhat * 4) / 4
            ans = g.integers(0, N, size=B)
            ans[0] = N - 1                                            # an all-zero answer row in every case
            onehot = np.concatenate([np.eye(N - 1, dtype=np.uint8), np.zeros((1, N - 1), dtype=np.uint8)], 0)
            y = onehot[ans]
            tag = f"{i}_{ties}"
            out[f"yhat{tag}"], out[f"y{tag}"] = yhat.astype(np.float32), y
            for k in out["ks"]:
                if k > N - 1:
                    continue
                m = SimpleNamespace(top_k=int(k), correct=torch.tensor(0), total=torch.tensor(0))
                ref_topk_update(m, torch.from_numpy(yhat.astype(np.float32)), torch.from_numpy(y))
                out[f"correct{tag}_k{k}"], out[f"total{tag}_k{k}"] = np.int64(int(m.correct)), np.int64(int(m.total))
                # a second update on the first half of the rows accumulates; compute() = correct / total
                ref_topk_update(m, torch.from_numpy(yhat[: (B + 1) // 2].astype(np.float32)), torch.from_numpy(y[: (B + 1) // 2]))
                out[f"correct2{tag}_k{k}"], out[f"total2{tag}_k{k}"] = np.int64(int(m.correct)), np.int64(int(m.total))
                out[f"acc2{tag}_k{k}"] = np.float64(float(ref_topk_compute(m)))
    # y_pred already without the answer slot (shape[1] == y_true.shape[1]): nothing is dropped (utils.py:61)
    yhat = g.random(size=(5, 10), dtype=np.float32)
    y = np.eye(10, dtype=np.uint8)[g.integers(0, 10, size=5)]
    m = SimpleNamespace(top_k=3, correct=torch.tensor(0), total=torch.tensor(0))
    ref_topk_update(m, torch.from_numpy(yhat), torch.from_numpy(y))
    out["yhat_same_width"], out["y_same_width"], out["correct_same_width_k3"] = yhat, y, np.int64(int(m.correct))
    return out


def main():
    dst = os.path.join(REPO, "tests", "golden")
    os.makedirs(dst, exist_ok=True)
    torch.set_num_threads(8)
    only = sys.argv[1:]
    if only == ["topk"]:
        np.savez_compressed(os.path.join(dst, "topk.npz"), **topk_cases())
        print("wrote topk")
        return
    if only:  # regenerate just the named cases
        for name in only:
            np.savez_compressed(os.path.join(dst, f"{name}.npz"), **run_case(name))
            print("wrote", name)
        return
    for name in CASES:
        np.savez_compressed(os.path.join(dst, f"{name}.npz"), **run_case(name))
        print("wrote", name)
    np.savez_compressed(os.path.join(dst, "tiny_wd_nan.npz"), **nan_case())
    np.savez_compressed(os.path.join(dst, "init_order.npz"), **init_order_case())
    np.savez_compressed(os.path.join(dst, "triplet.npz"), **triplet_cases())
    np.savez_compressed(os.path.join(dst, "topk.npz"), **topk_cases())
    print("done")


if __name__ == "__main__":
    main()
