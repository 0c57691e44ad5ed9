"""Diagnostic (X3_STAMPS build only): where one wave of the bf16x3 GEMM spends its cycles."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drin_amd import _lib
lib = _lib.load()
m, n, k = 25856, 768, 768
x = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda"); y = torch.empty(m, n, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    lib.drin_linear_fwd(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), m, n, k, 3, st)
torch.cuda.synchronize()
out = (C.c_ulonglong * 8)()
raw = C.CDLL(_lib.LIB_PATH)
print("rc", raw.drin_debug_x3_stamps(out))
names = ["k16 step 0", "split + LDS write", "issue global loads", "k16 step 1", "barrier"]
tot = sum(out[:5])
for nme, v in zip(names, out[:5]):
    print(f"{nme:20s} {v:10d} cycles  {100.0 * v / tot:5.1f}%   per K-block {v / (k // 32):8.1f}")
print("total", tot, "per K-block", tot / (k // 32))
