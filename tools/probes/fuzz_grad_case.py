"""A case of tests/test_gpu_fuzz.py whose gradient check is tight: whose rounding is it?  Gradients of the loss from the
oracle in fp64 (the yardstick), from the oracle in fp32, and from the library in f32 / bf16x3 / bf16x3_all."""
import importlib.util, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from drin_amd import synth
from drin_amd.metrics import TripletLoss
from drin_amd.model import Model
from oracle import drin_oracle as O
spec = importlib.util.spec_from_file_location("fz", os.path.join(os.path.dirname(__file__), "..", "..", "tests", "test_gpu_fuzz.py"))
fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
for i in [int(x) for x in sys.argv[1:]]:
    cfg, B, precision, seed = fz._draw(i)
    sd = synth.make_state_dict(cfg, 3 + i)
    T = cfg.max_entity_attr_token_len
    batch = synth.make_batch(cfg, B, seed % 100000, min_span=1, max_span=3, min_tokens=min(3, T))
    if cfg.token_level_entities and T >= 3:
        batch[8][0, 0, :] = 1; batch[8][-1, -1, :] = 0; batch[8][-1, -1, :3] = 1
    def oracle(dtype):
        p = {k: v.clone().to(dtype).requires_grad_(True) for k, v in sd.items()}
        out = O.forward(p, batch[:14], dtype=dtype, **O.config_kwargs(cfg))
        loss = O.triplet_loss(batch[14].to(dtype), out, cfg.triplet_margin)
        return out.detach(), dict(zip(p, torch.autograd.grad(loss, list(p.values()), allow_unused=True)))
    s64, g64 = oracle(torch.float64)
    s32, g32 = oracle(torch.float32)
    print(f"case {i}: {cfg.dataset_name} N={cfg.num_candidates_model} D={cfg.gcn_embed_dim} L={cfg.num_gcn_layers} {cfg.gcn_vertex_activation}/{cfg.gcn_edge_activation} B={B}; "
          f"scores: spread {float(s64.max() - s64.min()):.3e}, oracle fp32 vs fp64 {float((s32 - s64).abs().max()):.1e}")
    rows = {"oracle fp32": g32}
    dev = [t.to("cuda") for t in batch]
    for prec in ("f32", "bf16x3", "bf16x3_all"):
        m = Model(cfg, precision=prec).to("cuda"); m.load_state_dict(sd); m.train()
        TripletLoss(cfg.triplet_margin)(dev[14], m(dev[:14])).backward()
        rows["library " + prec] = {k: (p.grad.cpu() if p.grad is not None else None) for k, p in m.named_parameters()}
    for name, g in rows.items():
        worst, wk = 0.0, ""
        for k, r in g64.items():
            if r is None or r.norm() < 1e-10: continue
            rel = float((g[k].double() - r).norm() / r.norm())
            if rel > worst: worst, wk = rel, k
        print(f"   {name:20s} worst relative gradient error vs the fp64 oracle {worst:.2e}  ({wk})")
