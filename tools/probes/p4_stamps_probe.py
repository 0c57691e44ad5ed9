"""Per-phase cycle stamps of the two four-phase GEMM kernels inside ONE headline scoring call (library built with -DDRIN_P4_STAMPS,
selected by DRIN_LIB_PATH): for one workgroup (block 121) and one wave of each wave group, per K-block and phase
  S = phase start, A = after the first barrier (the MFMAs may start), M = after the MFMAs
-> reading half + wait for the first barrier (A - S), MFMA half (M - A), wait for the second barrier (S' - M).
    DRIN_LIB_PATH=$PWD/drin_amd/libdrin_hip_p4stamps.so python tools/probes/p4_stamps_probe.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from drin_amd import _lib, synth  # noqa: E402
from drin_amd.config import wikimel_config  # noqa: E402
from drin_amd.model import Model  # noqa: E402

dev = torch.device("cuda", 0)
cfg = wikimel_config()
model = Model(cfg).to(dev).eval()
batch = synth.make_device_batch(cfg, 4096, 100, dev)[:14]
with torch.no_grad():
    for _ in range(3):
        model(batch)
torch.cuda.synchronize()
raw = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * (2 * 2 * 32 * 4 * 3))()
assert raw.drin_debug_p4_stamps(buf) == 0
s = np.array(buf, dtype=np.int64).reshape(2, 2, 32, 4, 3)
for kern, name, nkb in ((0, "k_gemm_x3_planes_p4 (all planes by LDS-DMA; the last launch of the call: et' W_h2^T, K = 768)", 24),
                        (1, "k_gemm_bf16x3_p4 (fp32 A through registers, split in the MFMA half; x_i C_i^T, K = 2048)", 32)):
    print(name)
    for g in range(2):
        a = s[kern, g, 2:nkb - 1]                       # skip the first K-blocks (pipeline fill)
        nxt = s[kern, g, 3:nkb, 0, 0]                   # next K-block's phase-0 start
        line = f"  wave group {g}:"
        tot = 0.0
        for ph in range(4):
            S, A, M = a[:, ph, 0], a[:, ph, 1], a[:, ph, 2]
            Sn = a[:, ph + 1, 0] if ph < 3 else nxt
            read, mma, wait2 = np.median(A - S), np.median(M - A), np.median(Sn - M)
            tot += read + mma + wait2
            line += f"  phase {ph}: to-MFMA {read:5.0f} MFMA {mma:5.0f} barrier {wait2:4.0f} |"
        print(line + f"  K-block {tot:6.0f} cycles")
