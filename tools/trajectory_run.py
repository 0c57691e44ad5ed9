"""A longer training-trajectory comparison than the test suite affords: BASELINE config 3's own shape (WikiMEL-shaped, T = 64 token
blocks, N = 101, D = 768, R = 2048, batch 64), HIP `Model` (default split-bf16 arithmetic) + LibraryAdam against the CPU oracle's fp32
autograd + torch.optim.Adam, same seed-0 weights, same learnable synthetic stream (oracle/trajectory.py).
    python tools/trajectory_run.py [steps] [tokens] [precision] > gpurun_out/trajectory.txt
(precision "f32": the exact-fp32 MFMA kernels - the control that shows how far two fp32 loops that differ in summation order alone
drift apart over the same steps)
Prints one line per step (so a long run is never silent) and a summary."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from drin_amd.config import wikimel_config  # noqa: E402
from oracle.trajectory import trajectory  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
tokens = int(sys.argv[2]) if len(sys.argv) > 2 else 64
precision = sys.argv[3] if len(sys.argv) > 3 else "bf16x3"
torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
cfg = wikimel_config(max_entity_attr_token_len=tokens, batch_size=64)
t0 = time.time()
res = trajectory(cfg, steps, 0.15, "cuda", precision=precision, held_out=128, log=lambda s: print(f"[{time.time() - t0:6.1f} s] {s}", flush=True))
curve = res["curve"]
print(f"{precision}: {steps} steps at T = {tokens}: loss {curve[0][1]:.5f} -> {curve[-1][1]:.5f} (oracle), worst per-step |hip - oracle| "
      f"{max(abs(a - b) for a, b in curve):.2e}; held-out (128 mentions) hip {res['hip']} oracle {res['oracle']}; "
      f"max |held-out score diff| {res['max_abs_held_out_score_diff']:.2e}; {time.time() - t0:.0f} s")
