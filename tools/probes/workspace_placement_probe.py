"""Does k_entity_stream's +-6 % "placement band" follow the INPUT batch or the call's own WORKSPACE (the planes it writes)?
One process, one resident headline batch (4 096 mentions, 92 GB); the scoring call repeated with its workspace block forced to a new
address every trial (the caching allocator emptied, a pad of another size allocated first).  Prints the stream kernel's ms per trial.
    python tools/probes/workspace_placement_probe.py [trials]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from drin_amd import _lib, synth  # noqa: E402
from drin_amd.config import wikimel_config  # noqa: E402
from drin_amd.model import Model  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
cfg = wikimel_config()
sd = synth.make_state_dict(cfg, 7)
model = Model(cfg).to(dev).eval()
model.load_state_dict(sd)
batch = synth.make_device_batch(cfg, 4096, 100, dev)[:14]
pads = []
with torch.no_grad():
    model([t[:8] for t in batch])
    for t in range(trials):
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        pads.append(torch.empty((t * 53 + 17) << 20, dtype=torch.uint8, device=dev))     # keeps the previous address ranges busy
        for _ in range(3):
            model(batch)
        _lib.profile_begin(1 << 12)
        for _ in range(6):
            out = model(batch)
        prof = _lib.profile_end()
        torch.cuda.synchronize()
        print(f"trial {t}: stream {prof['stream'][0] / 6:.3f} ms  gemm_x3 {prof['gemm_x3'][0] / 6:.3f}  gemm_planes {prof['gemm_planes'][0] / 6:.3f}  gcn {prof['gcn'][0] / 6:.3f}", flush=True)
    # the same workspace again and again: how much does ONE placement move by itself?
    for t in range(3):
        _lib.profile_begin(1 << 12)
        for _ in range(6):
            model(batch)
        prof = _lib.profile_end()
        print(f"same workspace, repeat {t}: stream {prof['stream'][0] / 6:.3f} ms", flush=True)
