"""Does the stream kernel's time depend on where the driver put the batch?  One process: for each spacer size, allocate the
spacer FIRST (it takes the physical pages the batch would have got), draw the headline batch behind it, time the folded
scoring call's kernel classes (library profiler), free everything (empty_cache -> hipFree).  GPU only.
usage: python tools/probes/placement_probe.py [spacer GiB ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from drin_amd import _lib, synth
from drin_amd.config import wikimel_config
from drin_amd.model import Model

dev = torch.device("cuda:0")
cfg = wikimel_config()
model = Model(cfg).to(dev).eval()
B = 4096
sizes = [int(a) for a in sys.argv[1:]] or [0, 0, 32, 64, 96, 128, 160, 0]
for gib in sizes:
    spacer = torch.empty(gib << 30, dtype=torch.uint8, device=dev) if gib else None
    batch = synth.make_device_batch(cfg, B, 100, dev)[:14]
    with torch.no_grad():
        for _ in range(2):
            model(batch)
        _lib.profile_begin(1 << 12)
        for _ in range(4):
            model(batch)
        prof = _lib.profile_end()
    torch.cuda.synchronize()
    ms = {k: round(v[0] / 4, 3) for k, v in prof.items() if v[0] > 0}
    print(f"spacer {gib:3d} GiB: entity_text at {batch[7].data_ptr():#x}  {ms}", flush=True)
    del batch, spacer
    torch.cuda.empty_cache()
