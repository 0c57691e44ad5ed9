"""Per-parameter gradient differences of one fuzz case between two library precisions (and the fp64 oracle): where do they sit?"""
import importlib.util, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from drin_amd import synth
from drin_amd.metrics import TripletLoss
from drin_amd.model import Model
from oracle import drin_oracle as O
spec = importlib.util.spec_from_file_location("fz", os.path.join(os.path.dirname(__file__), "..", "..", "tests", "test_gpu_fuzz.py"))
fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
i = int(sys.argv[1])
act = sys.argv[2] if len(sys.argv) > 2 else None
cfg, B, precision, seed = fz._draw(i)
if act:
    cfg = cfg.with_(gcn_vertex_activation=act)
sd = synth.make_state_dict(cfg, 3 + i)
T = cfg.max_entity_attr_token_len
batch = synth.make_batch(cfg, B, seed % 100000, min_span=1, max_span=3, min_tokens=min(3, T))
p = {k: v.clone().double().requires_grad_(True) for k, v in sd.items()}
out = O.forward(p, batch[:14], dtype=torch.float64, **O.config_kwargs(cfg))
loss = O.triplet_loss(batch[14].double(), out, cfg.triplet_margin)
g64 = dict(zip(p, torch.autograd.grad(loss, list(p.values()), allow_unused=True)))
dev = [t.to("cuda") for t in batch]
res = {}
for prec in ("bf16x3", "bf16x3_all"):
    m = Model(cfg, precision=prec).to("cuda"); m.load_state_dict(sd); m.train()
    s = m(dev[:14])
    TripletLoss(cfg.triplet_margin)(dev[14], s).backward()
    res[prec] = ({k: (q.grad.cpu().double() if q.grad is not None else None) for k, q in m.named_parameters()}, s.detach().cpu())
print(f"case {i} vertex act {cfg.gcn_vertex_activation}: scores bf16x3 vs bf16x3_all {float((res['bf16x3'][1] - res['bf16x3_all'][1]).abs().max()):.2e}; vs oracle {float((res['bf16x3_all'][1].double() - out.detach()).abs().max()):.2e}")
for k, r in g64.items():
    if r is None: continue
    a, b = res["bf16x3"][0][k], res["bf16x3_all"][0][k]
    d = (b - r).abs()
    print(f"  {k:62s} |g| {float(r.norm()):.2e}  x3 {float((a - r).norm() / (r.norm() + 1e-30)):.1e}  all {float((b - r).norm() / (r.norm() + 1e-30)):.1e}  max elem diff {float(d.max()):.2e} at {tuple(int(x) for x in torch.unravel_index(d.argmax(), d.shape))}  n(|d|>1e-3 max|g|) {int((d > 1e-3 * r.abs().max()).sum())}/{d.numel()}")
