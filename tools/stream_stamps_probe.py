"""Cycle stamps of k_entity_stream (variant build -DDRIN_STREAM_STAMPS: python -m drin_amd.build --variant stamps with
DRIN_EXTRA_FLAGS=-DDRIN_STREAM_STAMPS; run with DRIN_LIB_PATH=drin_amd/libdrin_hip_stamps.so): where does a workgroup's time go
at WikiDiverse's 11 candidates per mention, and at WikiMEL's 101?   usage: stream_stamps_probe.py [wikidiverse|wikimel] [B]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from drin_amd import _lib, synth
from drin_amd.config import DrinConfig, wikimel_config
from drin_amd.model import Model
which = sys.argv[1] if len(sys.argv) > 1 else "wikidiverse"
cfg = DrinConfig() if which == "wikidiverse" else wikimel_config()
B = int(sys.argv[2]) if len(sys.argv) > 2 else (16384 if which == "wikidiverse" else 4096)
raw = C.CDLL(_lib.LIB_PATH)
model = Model(cfg).to("cuda").eval()
batch = synth.make_device_batch(cfg, B, 100, "cuda")[:14]
with torch.no_grad():
    for _ in range(3):
        model(batch)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (64 * 8))()
assert raw.drin_debug_stream_stamps(buf) == 0
s = np.array(buf, dtype=np.int64).reshape(64, 8)
names = ["mention vectors -> LDS (to the first barrier)", "norms / kappa dots", "second barrier", "candidate loop", "cross-wave reduction (4 rounds)", "write-out"]
d = np.stack([s[:, i + 1] - s[:, i] for i in range(6)], 1)
ok = (d > 0).all(1) & (d < 10_000_000).all(1)
d = d[ok]
if not len(d):
    raise SystemExit(f"{which} B = {B}: no workgroup stamped")
print(f"{which} B = {B}: {ok.sum()} workgroups stamped (wave 0), cycles at the 100 MHz-free shader clock counter; median (10 % .. 90 %)")
tot = d.sum(1)
for i, n in enumerate(names):
    print(f"  {n:48s} {np.median(d[:, i]):9.0f}  ({np.percentile(d[:, i], 10):.0f} .. {np.percentile(d[:, i], 90):.0f})   {100 * np.median(d[:, i]) / np.median(tot):5.1f} %")
print(f"  {'whole workgroup':48s} {np.median(tot):9.0f}")
