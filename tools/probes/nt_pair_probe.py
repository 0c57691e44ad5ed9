"""Upper bound of what grouping two independent mid-sized split-bf16 NT products into one launch could give: the same two
products back to back on one stream against side by side on two streams.  GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from drin_amd import _lib

lib = _lib.load()
dev = "cuda"
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for (m, n, k) in ((12928, 768, 768), (6464, 768, 768), (6464, 768, 2048)):
    xs = [torch.randn(m, k, device=dev) for _ in range(2)]
    ws = [torch.randn(n, k, device=dev) / k ** 0.5 for _ in range(2)]
    ys = [torch.empty(m, n, device=dev) for _ in range(2)]
    torch.cuda.synchronize()

    def run(streams, iters):
        for _ in range(iters):
            for i in range(2):
                _lib.check(lib.drin_linear_fwd(xs[i].data_ptr(), ws[i].data_ptr(), None, ys[i].data_ptr(), m, n, k, 3, streams[i].cuda_stream))

    for name, streams in (("one stream ", (s1, s1)), ("two streams", (s1, s2))):
        run(streams, 3)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s1)
        s2.wait_event(e0)
        run(streams, 20)
        ev = torch.cuda.Event()
        ev.record(s2)
        s1.wait_event(ev)
        e1.record(s1)
        torch.cuda.synchronize()
        print(f"{m} x {n} x {k}  {name}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us per pair of products", flush=True)
