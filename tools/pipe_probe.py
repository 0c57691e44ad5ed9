"""Side-by-side phases of the fused forward (drin_set_pipeline): ms / step and score differences against the one-stream
schedule, one process, one resident batch.  python tools/pipe_probe.py [B] [dataset] [settings ...]   setting = cus[:pairs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drin_amd import _lib, synth
from drin_amd.config import wikimel_config, wikidiverse_config
from drin_amd.model import Model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dataset = sys.argv[2] if len(sys.argv) > 2 else "wikimel"
settings = sys.argv[3:] or ["0", "96", "112", "128", "144", "128:103424"]
dev = "cuda"
cfg = wikimel_config() if dataset == "wikimel" else wikidiverse_config()
model = Model(cfg, precision="bf16x3").to(dev).eval()
batch = synth.make_device_batch(cfg, B, 5, dev)[:14]
lib = _lib.load()


def run(n):
    with torch.no_grad():
        for _ in range(n):
            out = model(batch)
    return out


base = None
for s in settings:
    cus, _, pairs = s.partition(":")
    _lib.check(lib.drin_set_pipeline(int(cus), int(pairs) if pairs else -1))
    run(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run(8)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 8 * 1e3
    _lib.profile_begin()
    run(1)
    torch.cuda.synchronize()
    prof = _lib.profile_end()
    sc = out[0] if isinstance(out, (tuple, list)) else out
    if base is None:
        base = sc.clone()
    diff = (sc - base).abs().max().item()
    kern = "  ".join(f"{k} {v[0]:.2f}" for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:5])
    print(f"{s:>12}: {ms:7.3f} ms/step  {B * cfg.num_candidates_model / ms / 1e3:6.2f} M pairs/s  max|score - one-stream| {diff:.2e}   {kern}", flush=True)
