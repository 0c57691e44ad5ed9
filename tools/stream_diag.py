import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from drin_amd import synth
from drin_amd.config import DrinConfig
class A: gpus=1; stub=False
ctx = bench.Ctx(A)
dev = ctx.dev
tc = DrinConfig(num_candidates_data=1000)
tsd = synth.make_state_dict(tc, 7)
m = bench.make_model(tc, tsd, dev, "bf16x3")
table, g = bench.build_table(tc, 1_000_000, dev)
table.enable_cache()
with torch.no_grad():
    m(bench.make_table_chunk(tc, table, 64, 99, dev, g))
    ib = bench.make_table_chunk(tc, table, 4096, 1, dev, g)
    torch.cuda.synchronize()
    for name, fn in (("gen", lambda: bench.make_table_chunk(tc, table, 4096, 2, dev, g)), ("score", lambda: m(ib))):
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(5): fn()
            torch.cuda.synchronize()
            print(name, rep, (time.perf_counter() - t0) / 5 * 1e3, "ms")
    for n in (8, 40):
        el, nch, out, last = bench.stream_table(ctx, m, tc, table, g, n * 4096, 4096)
        print("stream", n, el / nch * 1e3, "ms/chunk", torch.cuda.memory_reserved() / 2**30, "GiB reserved")
