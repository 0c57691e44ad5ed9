"""Do an HBM-bound kernel (k_entity_stream, held to ONE workgroup per CU by LDS padding) and an MFMA-bound GEMM
(64 x 128 tiles, 48 KB LDS) share CUs when launched on two HIP streams?  Compares each alone with both together.

Measured (MI355X, round 1): B=4096: model alone 18.4-18.8 ms, GEMM loop alone 7.7 ms, both 24.8-25.2 ms (sum 26.1-26.6):
only ~1.3 ms is hidden, with the stream kernel at 2 workgroups/CU (146 KB LDS, no room for the GEMM) AND at 1
workgroup/CU (LDS padded to 82 KB via a one-line patch adding DRIN_STREAM_LDS_PAD bytes to the dynamic LDS size in
launch_stream_t - not in the product).  The co-resident GEMM starves: its tile loads queue behind the ~100 KB per CU the
stream kernel keeps in flight.  Side result: the stream kernel is as fast at 1 workgroup/CU as at 2."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drin_amd import _lib, synth
from drin_amd.config import wikimel_config
from drin_amd.model import Model

dev = "cuda"
cfg = wikimel_config()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
model = Model(cfg, precision="bf16x3").to(dev).eval()
batch = synth.make_device_batch(cfg, B, 5, dev)[:14]
lib = _lib.load()
M, N, K = 8192, 768, 2048
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); y = torch.empty(M, N, device=dev)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

def run_model():
    with torch.cuda.stream(sa), torch.no_grad():
        model(batch)

def run_gemm():
    for _ in range(reps):
        _lib.check(lib.drin_linear_fwd(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), M, N, K, _lib.PREC_BF16X3_ALL, sb.cuda_stream))

def timed(fns, n=5):
    for f in fns: f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for f in fns: f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

a, g = timed([run_model]), timed([run_gemm])
both = timed([run_model, run_gemm])
fl = 2.0 * M * N * K * reps
print(f"LDS pad {os.environ.get('DRIN_STREAM_LDS_PAD', '0')}: model alone {a:.2f} ms | gemm alone {g:.2f} ms ({fl / g / 1e9:.0f} TF/s alg) | both {both:.2f} ms "
      f"(sum {a + g:.2f}, max {max(a, g):.2f})")
