"""Is the placement spread particular to k_entity_stream's access pattern?  A plain linear read (torch's own reduction kernel
over an 81 GB fp32 tensor) over freshly allocated tensors, several draws in one process.  GPU only."""
import torch

dev = torch.device("cuda:0")
n = 4096 * 101 * 64 * 768
for trial in range(8):
    x = torch.empty(n, dtype=torch.float32, device=dev)
    x.zero_()
    v = x.view(-1, 1 << 20)
    for _ in range(2):
        s = v.sum(dim=1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        s = v.sum(dim=1)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 4
    print(f"trial {trial}: at {x.data_ptr():#x}  {ms:.3f} ms  {4 * n / ms / 1e9:.2f} TB/s", flush=True)
    del x, v, s
    torch.cuda.empty_cache()
    if trial % 2 == 1:
        keep = torch.empty((trial + 1) << 33, dtype=torch.uint8, device=dev)   # 16, 32, 48 GiB: shift the next draw
