"""One fresh process: draw the headline batch, print where its big tensors landed and the stream kernel's time (see
placement_probe.py for the in-process version).  GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from drin_amd import _lib, synth
from drin_amd.config import wikimel_config
from drin_amd.model import Model

dev = torch.device("cuda:0")
cfg = wikimel_config()
model = Model(cfg).to(dev).eval()
batch = synth.make_device_batch(cfg, 4096, 100, dev)[:14]
with torch.no_grad():
    for _ in range(2):
        model(batch)
    _lib.profile_begin(1 << 12)
    for _ in range(4):
        model(batch)
    prof = _lib.profile_end()
torch.cuda.synchronize()
p = [batch[i].data_ptr() for i in (7, 9, 10, 0, 4)]
print("stream", round(prof["stream"][0] / 4, 3), "entity_text %#x image %#x object %#x mention_text %#x mention_image %#x" % tuple(p),
      "text mod 1GiB %#x" % (p[0] % (1 << 30)), flush=True)
