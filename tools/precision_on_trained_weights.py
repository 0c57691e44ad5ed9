"""How the precision modes' score errors move as the weights are TRAINED (every other precision measurement uses freshly initialised
weights): the reference's loop (train.py:30-56: TripletLoss, Adam lr 1e-3, batch 64) on the learnable synthetic stream at the
reference's width (D = 768, R = 2048, N = 101, T = 8); after 0 / 20 / 40 / 100 / 200 / 400 steps, 512 held-out mentions are scored in
the exact-fp32 arithmetic and in each mode with the SAME weights.   python tools/precision_on_trained_weights.py [strength] [wikimel|wikidiverse] [steps,steps,...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drin_amd import synth
from drin_amd.config import wikimel_config
from drin_amd.metrics import TripletLoss
from drin_amd.model import Model
from drin_amd.train import make_adam
from oracle import drin_oracle as O

DEV = "cuda"
strength = float(sys.argv[1]) if len(sys.argv) > 1 else 0.15
dataset = sys.argv[2] if len(sys.argv) > 2 else "wikimel"
if dataset == "wikidiverse":
    from drin_amd.config import DrinConfig
    cfg = DrinConfig(batch_size=64)
else:
    cfg = wikimel_config(max_entity_attr_token_len=8, batch_size=64)
HELD = 512 if dataset == "wikimel" else 2048          # enough pairs for the one-pass kernel's whole 256 x 256 grids
torch.manual_seed(0)
model = Model(cfg).to(DEV)
opt = make_adam(model, cfg.learning_rate)
loss_fn = TripletLoss(cfg.triplet_margin)
held = [t.to(DEV) for t in synth.plant_gold_signal(cfg, synth.make_device_batch(cfg, HELD, 999, "cpu"), strength)]
y = held[14].cpu()
modes = ("bf16x3", "bf16x3_if16")
print(f"gold signal strength {strength}; columns: max |score - exact fp32| over the held-out scores (top-1 agreement with the exact path)")
step, loss = 0, float("nan")
CHECKPOINTS = tuple(int(x) for x in sys.argv[3].split(",")) if len(sys.argv) > 3 else (0, 20, 40, 100, 200, 400)
for upto in CHECKPOINTS:
    while step < upto:
        b = [t.to(DEV) for t in synth.plant_gold_signal(cfg, synth.make_device_batch(cfg, 64, 50 + step, "cpu"), strength)]
        opt.zero_grad(set_to_none=True)
        l = loss_fn(b[14], model(b[:14]))
        l.backward()
        opt.step()
        loss, step = float(l.detach()), step + 1
        if step % 100 == 0:
            print(f"  ... step {step}, train loss {loss:.4f}", flush=True)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    out = {}
    for prec in ("f32",) + modes:
        m = Model(cfg, precision=prec).to(DEV).eval()
        m.load_state_dict(sd)
        with torch.no_grad():
            out[prec] = m(held[:14]).cpu()
    # the per-entity cache's two row formats on the same weights: the held-out pairs' entity rows as a table (one entity per pair)
    from drin_amd.model import EntityTable, IndexedBatch
    E = HELD * cfg.num_candidates_model
    table = EntityTable(held[7].reshape(E, *held[7].shape[2:]), (held[8].reshape(E, -1) if cfg.token_level_entities else None), held[9].reshape(E, *held[9].shape[2:]),
                        held[10].reshape(E, *held[10].shape[2:]), held[11].reshape(E, -1))
    ib = IndexedBatch(held[:7], table, torch.arange(E, device=DEV).view(HELD, -1), held[12], held[13])
    m = Model(cfg).to(DEV).eval()
    m.load_state_dict(sd)
    with torch.no_grad():
        table.enable_cache(True)
        c32 = m(ib).cpu()
        table.enable_cache(True, format="mixed_f16")
        c16 = m(ib).cpu()
    table.enable_cache(False)
    del table, ib
    ref = out["f32"]
    cache_cells = f"cache fp32 rows {float((c32 - ref).abs().max()):.2e}  mixed-f16 rows {float((c16 - ref).abs().max()):.2e} (vs fp32 rows {float((c16 - c32).abs().max()):.2e})"
    top1 = O.topk_counts(ref, y, 1)[0]
    spread = float(ref[:, :-1].max(1).values.sub(ref[:, :-1].median(1).values).mean())
    wn = float(torch.cat([v.flatten() for k, v in sd.items() if k.endswith("weight") and v.dim() == 2]).norm())
    cells = "  ".join(f"{p} {float((out[p] - ref).abs().max()):.2e} ({float((out[p][:, :-1].argmax(1) == ref[:, :-1].argmax(1)).float().mean()):.4f})" for p in modes)
    print(f"after {step:3d} steps (train loss {loss:.4f}, held-out top-1 {top1}/{HELD}, mean top-minus-median score {spread:.3f}, |W| {wn:.1f}):  {cells}  {cache_cells}", flush=True)
