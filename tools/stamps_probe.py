"""Per-wave cycle stamps of the K-loop of k_gemm_x3_planes (build with -DDRIN_STAMPS): where does a K-block's time go?"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from drin_amd import _lib
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
m, n, k = 103424, 768, int(sys.argv[1]) if len(sys.argv) > 1 else 768
dev = "cuda"; st = torch.cuda.current_stream().cuda_stream
x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) / k ** 0.5; y = torch.empty(m, n, device=dev)
pl = [torch.empty(t.shape, dtype=torch.bfloat16, device=dev) for t in (x, x, w, w)]
lib.drin_split_planes(x.data_ptr(), pl[0].data_ptr(), pl[1].data_ptr(), x.numel(), st)
lib.drin_split_planes(w.data_ptr(), pl[2].data_ptr(), pl[3].data_ptr(), w.numel(), st)
args = [p.data_ptr() for p in pl] + [None, y.data_ptr(), m, n, k, st]
for _ in range(3): _lib.check(lib.drin_linear_planes_fwd(*args))
torch.cuda.synchronize()
buf = (C.c_ulonglong * (8 * 64 * 4 + 64))()
assert raw.drin_debug_stamps(buf) == 0
allv = np.array(buf, dtype=np.int64)
s = allv[:8 * 64 * 4].reshape(8, 64, 4)[:, : k // 32]
outer = allv[8 * 64 * 4:8 * 64 * 4 + 32].reshape(8, 4)
for wv in range(8):
    a = s[wv]
    frag = a[:, 1] - a[:, 0]; mfma = a[:, 2] - a[:, 1]; bar = a[:, 3] - a[:, 2]; tot = a[1:, 0] - a[:-1, 0]
    print(f"wave {wv}: per K-block cycles (median)  fragment reads {np.median(frag):6.0f}  MFMA issue phase {np.median(mfma):6.0f}  barrier wait {np.median(bar):6.0f}  total {np.median(tot):6.0f}")
for wv in range(8):
    o = outer[wv]
    print(f"wave {wv}: prologue {o[1] - o[0]:7d}  K-loop {o[2] - o[1]:8d}  epilogue {o[3] - o[2]:7d} cycles")
