"""Weight gradients of one training step (WikiMEL-shaped, D = 768, R = 2048, N = 101, B = 64, T = 8) against autograd through the
fp64 oracle: the split product (3 passes) and the one-pass experiment (drin_set_weight_gradient_passes(1)), per parameter."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from drin_amd import _lib, synth
from drin_amd.config import wikimel_config
from drin_amd.metrics import TripletLoss
from drin_amd.model import Model
from oracle import drin_oracle as O
torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
cfg = wikimel_config(max_entity_attr_token_len=8, batch_size=64)
sd = synth.make_state_dict(cfg, 7)
batch = synth.plant_gold_signal(cfg, synth.make_device_batch(cfg, 64, 50, "cpu"), 0.15)
p = {k: v.clone().double().requires_grad_(True) for k, v in sd.items()}
loss = O.triplet_loss(batch[14].double(), O.forward(p, batch[:14], dtype=torch.float64), cfg.triplet_margin)
g64 = dict(zip(p, torch.autograd.grad(loss, list(p.values()), allow_unused=True)))
dev = [t.to("cuda") for t in batch]
lib = _lib.load()
for passes in (3, 1):
    _lib.check(lib.drin_set_weight_gradient_passes(passes))
    m = Model(cfg).to("cuda"); m.load_state_dict(sd); m.train()
    TripletLoss(cfg.triplet_margin)(dev[14], m(dev[:14])).backward()
    rows = []
    for k, q in m.named_parameters():
        if g64[k] is None or q.grad is None: continue
        rows.append(((q.grad.cpu().double() - g64[k]).norm().item() / g64[k].norm().item(), k))
    rows.sort(reverse=True)
    print(f"weight-gradient passes = {passes}: worst relative gradient error vs the fp64 oracle " + ", ".join(f"{k.split('.')[-2]}.{k.split('.')[-1]} {e:.1e}" for e, k in rows[:4]) +
          f"; median {sorted(e for e, _ in rows)[len(rows) // 2]:.1e}")
lib.drin_set_weight_gradient_passes(-1)
