"""What the vendor bf16 GEMM reaches on the shapes of the MFMA phase (a ceiling to read the hand-written split-bf16 kernels
against, nothing the library calls): torch.matmul in bf16 on [M, K'] x [K', N] with K' = K (one pass) and 3 K (the three
passes of the split product laid side by side along K).  GPU only."""
import torch

dev = "cuda"
for (m, n, k) in ((413696, 768, 2048), (413696, 768, 6144), (413696, 768, 768), (413696, 768, 2304), (12928, 768, 768), (12928, 768, 2304)):
    a = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    b = torch.randn(n, k, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        y = a @ b.t()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    iters = 10
    for _ in range(iters):
        y = a @ b.t()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"bf16 {m:7d} x {n:4d} x {k:5d}: {us:9.1f} us  {2.0 * m * n * k / us / 1e6:8.1f} TFLOP/s executed", flush=True)
    del a, b, y
