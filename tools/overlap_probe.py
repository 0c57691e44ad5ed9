"""Does running two half-batches on two HIP streams overlap the HBM-bound and MFMA-bound kernels?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from drin_amd import synth
from drin_amd.config import wikimel_config
from drin_amd.model import Model

cfg = wikimel_config()
dev = "cuda"
model = Model(cfg, precision="bf16x3").to(dev).eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nsplit = int(sys.argv[2]) if len(sys.argv) > 2 else 2
batch = synth.make_device_batch(cfg, B, 5, dev)[:14]
parts = [[t[i * (B // nsplit):(i + 1) * (B // nsplit)].contiguous() for t in batch] for i in range(nsplit)]
streams = [torch.cuda.Stream() for _ in range(nsplit)]

def seq():
    for p in parts:
        model(p)

def par():
    for s, p in zip(streams, parts):
        with torch.cuda.stream(s):
            model(p)

def whole():
    model(batch)

with torch.no_grad():
    for name, fn in (("whole batch, 1 stream", whole), ("%d parts sequential" % nsplit, seq), ("%d parts on %d streams" % (nsplit, nsplit), par)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"{name:28s} {dt * 1e3:7.3f} ms  {B * cfg.num_candidates_model / dt / 1e6:6.2f} Mpairs/s")
