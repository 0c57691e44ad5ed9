"""Does the stream kernel's time depend on where the driver put the batch?  One process: draw the headline batch, time the
folded scoring call's kernel classes (library profiler), free everything (empty_cache -> hipFree), optionally leave a
spacer allocation behind, draw again.  GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from drin_amd import _lib, synth
from drin_amd.config import DrinConfig
from drin_amd.model import Model

dev = torch.device("cuda:0")
cfg = DrinConfig(dataset_name="wikimel", num_candidates_data=100)
model = Model(cfg).to(dev).eval()
B = 4096
spacers = []
for trial in range(6):
    batch = synth.make_device_batch(cfg, B, 100, dev)[:14]
    ptr = batch[7].data_ptr()
    with torch.no_grad():
        for _ in range(2):
            model(batch)
        _lib.profile_begin(1 << 12)
        for _ in range(4):
            model(batch)
        prof = _lib.profile_end()
    torch.cuda.synchronize()
    ms = {k: round(v[0] / 4, 3) for k, v in prof.items() if v[0] > 0}
    print(f"trial {trial}: entity_text at {ptr:#x} (mod 2 MiB {ptr % (1 << 21):#x}, mod 1 GiB {ptr % (1 << 30):#x})  {ms}", flush=True)
    del batch
    torch.cuda.empty_cache()
    if trial % 2 == 1:   # leave a spacer behind so that the next draw cannot land on the same pages
        spacers.append(torch.empty((1 << 30) + trial * (37 << 20), dtype=torch.uint8, device=dev))
