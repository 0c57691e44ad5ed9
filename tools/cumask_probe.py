"""Do the HBM-bound stream kernel and the MFMA-bound GEMMs overlap when they run on DISJOINT sets of CUs
(hipExtStreamCreateWithCUMask)?  Co-residency on the same CUs does not work (tools/coresidency_probe.py)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drin_amd import _lib, synth
from drin_amd.config import wikimel_config
from drin_amd.model import Model

hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]

def masked_stream(cus):
    """cus: iterable of CU indices (0..255)"""
    words = (C.c_uint32 * 8)()
    for c in cus:
        words[c // 32] |= 1 << (c % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)

dev = "cuda"
torch.cuda.init(); torch.zeros(1, device=dev)
cfg = wikimel_config()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n_stream = int(sys.argv[2]) if len(sys.argv) > 2 else 96
layout = sys.argv[3] if len(sys.argv) > 3 else "block"     # block: CUs [0, n) ; stride: every k-th CU
model = Model(cfg, precision="bf16x3").to(dev).eval()
batch = synth.make_device_batch(cfg, B, 5, dev)[:14]
lib = _lib.load()
M, N, K = 8192, 768, 2048
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); y = torch.empty(M, N, device=dev)
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 40
if layout == "block":
    a_cus = list(range(n_stream)); b_cus = list(range(n_stream, 256))
else:
    a_cus = [c for c in range(256) if (c * n_stream) // 256 != ((c - 1) * n_stream) // 256 or c == 0][:n_stream]
    b_cus = [c for c in range(256) if c not in set(a_cus)]
sa, sb, sall = masked_stream(a_cus), masked_stream(b_cus), torch.cuda.Stream()

def run_model(st):
    with torch.cuda.stream(st), torch.no_grad():
        model(batch)

def run_gemm(st):
    for _ in range(reps):
        _lib.check(lib.drin_linear_fwd(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), M, N, K, _lib.PREC_BF16X3_ALL, st.cuda_stream))

def timed(fns, n=5):
    for f in fns: f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for f in fns: f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

def stream_ms(st, with_gemm=None):
    torch.cuda.synchronize()
    _lib.profile_begin()
    if with_gemm is not None: run_gemm(with_gemm)
    run_model(st)
    torch.cuda.synchronize()
    p = _lib.profile_end()
    return p["stream"][0]

for f in (lambda: run_model(sall), lambda: run_model(sa), lambda: run_gemm(sb)): f()
torch.cuda.synchronize()
print(f"B={B}: stream kernel on all CUs {stream_ms(sall):.2f} ms | on {len(a_cus)} CUs ({layout}) {stream_ms(sa):.2f} ms | "
      f"on {len(a_cus)} CUs while GEMMs run on the other {len(b_cus)}: {stream_ms(sa, sb):.2f} ms")
g_all, g_b = timed([lambda: run_gemm(sall)]), timed([lambda: run_gemm(sb)])
m_a = timed([lambda: run_model(sa)])
both = timed([lambda: run_model(sa), lambda: run_gemm(sb)])
print(f"GEMM loop on all CUs {g_all:.2f} ms | on {len(b_cus)} CUs {g_b:.2f} ms | model on {len(a_cus)} CUs {m_a:.2f} ms | both concurrently {both:.2f} ms "
      f"(sum {m_a + g_b:.2f}, max {max(m_a, g_b):.2f})")
