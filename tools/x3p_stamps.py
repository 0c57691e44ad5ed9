"""Diagnostic (X3_STAMPS build only): cycle shares of one wave of the LDS-DMA split-bf16 GEMM."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drin_amd import _lib
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
raw.drin_debug_launch = None
m, n = 25856, 768
st = torch.cuda.current_stream().cuda_stream
lg = raw._ZN4drin21launch_gemm_x3_planesEPKvS1_lS1_S1_lPKfPfllii if False else None
from drin_amd.config import wikimel_config
from drin_amd import synth
from drin_amd.model import Model
cfg = wikimel_config()
model = Model(cfg, precision="bf16x3").cuda().eval()
batch = synth.make_device_batch(cfg, 256, 3, "cuda")[:14]
with torch.no_grad():
    for _ in range(3):
        model(batch)
torch.cuda.synchronize()
out = (C.c_ulonglong * 8)()
print("rc", raw.drin_debug_x3p_stamps(out), "(last planes GEMM of the forward: et1 x W_h2, K = 768)")
names = ["issue LDS-DMA", "k16 step 0", "k16 step 1", "vmcnt(0)", "barrier"]
tot = sum(out[:5])
for nme, v in zip(names, out[:5]):
    print(f"{nme:16s} {v:10d} cycles {100.0 * v / tot:5.1f}%  per K-block {v / 24:8.1f}")
print("total per K-block", tot / 24)
