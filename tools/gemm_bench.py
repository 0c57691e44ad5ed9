"""Times drin_linear_fwd (the GEMM kernels) on the pair-sized shapes of the path.  GPU only.
usage: python tools/gemm_bench.py [rows]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drin_amd import _lib

lib = _lib.load()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 25856
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
for prec, name in ((0, "f32"), (3, "bf16x3")):
    for (m, n, k) in ((rows, 768, 768), (rows, 768, 2048), (2 * rows, 768, 768), (512, 768, 768), (512, 1536, 768), (256, 768, 2048)):
        x = torch.randn(m, k, device=dev)
        w = torch.randn(n, k, device=dev) / k ** 0.5
        b = torch.randn(n, device=dev)
        y = torch.empty(m, n, device=dev)
        for _ in range(3):
            _lib.check(lib.drin_linear_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), m, n, k, prec, st))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        iters = 20
        for _ in range(iters):
            lib.drin_linear_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), m, n, k, prec, st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        tf = 2.0 * m * n * k / us / 1e6
        print(f"{name:7s} {m:6d}x{n:4d}x{k:4d}: {us:8.1f} us  {tf:7.1f} TFLOP/s (algorithmic)")

# LDS-DMA split-bf16 GEMM on pre-split planes
for (m, n, k) in ((rows, 768, 768), (rows, 768, 2048), (4 * rows, 768, 768), (4 * rows, 768, 2048)):
    x = torch.randn(m, k, device=dev)
    w = torch.randn(n, k, device=dev) / k ** 0.5
    y = torch.empty(m, n, device=dev)
    pl = [torch.empty(t.shape, dtype=torch.bfloat16, device=dev) for t in (x, x, w, w)]
    lib.drin_split_planes(x.data_ptr(), pl[0].data_ptr(), pl[1].data_ptr(), x.numel(), st)
    lib.drin_split_planes(w.data_ptr(), pl[2].data_ptr(), pl[3].data_ptr(), w.numel(), st)
    args = [p.data_ptr() for p in pl] + [None, y.data_ptr(), m, n, k, st]
    for _ in range(3):
        _lib.check(lib.drin_linear_planes_fwd(*args))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.drin_linear_planes_fwd(*args)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"planes  {m:6d}x{n:4d}x{k:4d}: {us:8.1f} us  {2.0 * m * n * k / us / 1e6:7.1f} TFLOP/s (algorithmic)")
