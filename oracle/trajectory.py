"""Test infrastructure (never imported by the product): a training TRAJECTORY of the HIP `Model` checked against the same
loop driven by the CPU oracle.

`train.py:30-56` of the reference is: scores = Model.forward(batch) -> TripletLoss -> loss.backward() -> Adam(lr).step(), at
batch 64 (`common/args.py:118`).  `trajectory` runs that loop twice from the same seed-0 initial weights over the same
learnable synthetic stream (`drin_amd.synth.plant_gold_signal`; the datasets are not available offline) - once with the HIP
`Model` in the arithmetic under test + the library's one-launch Adam on the GPU, once with `oracle.drin_oracle.forward`
(fp32, torch autograd) + `torch.optim.Adam` on the CPU - and reports the per-step losses of both and their held-out loss and
top-k counts (`common/utils.py:26-73`).  Used by tests/test_gpu_round4.py and by bench.py's `train_parity` (the checker of
the train_step leg, outside every timed region)."""
from __future__ import annotations

from typing import Callable, Optional

import torch

from drin_amd import synth
from drin_amd.config import DrinConfig

from . import drin_oracle as O


def trajectory(cfg: DrinConfig, steps: int, strength: float, dev, precision: str = "bf16x3", held_out: int = 128, seed0: int = 50,
               log: Optional[Callable[[str], None]] = None) -> dict:
    """A fresh batch of `cfg.batch_size` mentions every step (the curve is generalisation, not memorised mentions).
    Returns {"curve": [(hip loss, oracle loss)...], "hip": {...}, "oracle": {...}, "max_abs_held_out_score_diff": ...}."""
    from drin_amd.metrics import TripletLoss
    from drin_amd.model import Model
    from drin_amd.train import make_adam

    B = cfg.batch_size
    torch.manual_seed(0)
    hip = Model(cfg, precision=precision).to(dev)                       # seed-0 init in the reference's constructor order
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in hip.state_dict().items()}
    kw = O.config_kwargs(cfg)
    o_hip, o_ora = make_adam(hip, cfg.learning_rate), torch.optim.Adam(list(p.values()), lr=cfg.learning_rate)
    loss_fn = TripletLoss(cfg.triplet_margin)
    held = synth.plant_gold_signal(cfg, synth.make_device_batch(cfg, held_out, 999, "cpu"), strength)
    curve = []
    for i in range(steps):
        b = synth.plant_gold_signal(cfg, synth.make_device_batch(cfg, B, seed0 + i, "cpu"), strength)
        bd = [t.to(dev) for t in b]
        o_hip.zero_grad(set_to_none=True)
        lh = loss_fn(bd[14], hip(bd[:14]))
        lh.backward()
        o_hip.step()
        o_ora.zero_grad(set_to_none=True)
        lo = O.triplet_loss(b[14], O.forward(p, b[:14], **kw), cfg.triplet_margin)
        lo.backward()
        o_ora.step()
        curve.append((float(lh.detach()), float(lo.detach())))
        if log:
            log(f"step {i}: loss hip {curve[-1][0]:.6f} oracle {curve[-1][1]:.6f} diff {curve[-1][0] - curve[-1][1]:+.2e}")

    def summary(scores, y):
        return {"loss": float(O.triplet_loss(y, scores, cfg.triplet_margin)),
                "topk": {k: int(O.topk_counts(scores, y, k)[0]) for k in (1, 5)}}

    with torch.no_grad():
        s_hip = hip.eval()([t.to(dev) for t in held[:14]]).cpu()
        s_ora = O.forward(p, held[:14], **kw)
    return {"curve": curve, "hip": summary(s_hip, held[14]), "oracle": summary(s_ora, held[14]),
            "max_abs_held_out_score_diff": float((s_hip - s_ora).abs().max()), "held_out_mentions": held_out,
            "optimizer": type(o_hip).__name__}
