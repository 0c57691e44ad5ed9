"""The stream kernel's two speed modes between fresh processes (profiles/r5_run_to_run.txt: ~8.15 and ~8.8 ms): does the mode follow the ALLOCATION?  One process,
the headline batch allocated K times over (freed, the caching allocator emptied, a pad of varying size allocated first); per allocation: the device addresses of the
big tensors and the stream kernel's ms.   python tools/probes/placement_realloc_probe.py [K]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from drin_amd import _lib, synth  # noqa: E402
from drin_amd.config import wikimel_config  # noqa: E402
from drin_amd.model import Model  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda", 0)
cfg = wikimel_config()
sd = synth.make_state_dict(cfg, 7)
model = Model(cfg).to(dev).eval()
model.load_state_dict(sd)
pads = []
with torch.no_grad():
    for k in range(K):
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        if k % 2 == 1:
            pads.append(torch.empty((k * 997 + 123) << 20, dtype=torch.uint8, device=dev))
        batch = synth.make_device_batch(cfg, 4096, 100 + k, dev)[:14]
        for _ in range(3):
            model(batch)
        _lib.profile_begin(1 << 12)
        for _ in range(6):
            model(batch)
        prof = _lib.profile_end()
        torch.cuda.synchronize()
        et, ei, eo = batch[7].data_ptr(), batch[9].data_ptr(), batch[10].data_ptr()
        print(f"alloc {k}: stream {prof['stream'][0] / 6:.3f} ms | entity_text @ {et:#x} (mod 1 GiB {et % (1 << 30):#x}, mod 2 MiB {et % (1 << 21):#x}) image @ {ei:#x} object @ {eo:#x}", flush=True)
        del batch
