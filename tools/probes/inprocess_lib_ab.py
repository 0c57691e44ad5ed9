"""IN-PROCESS A/B of two builds of the library on ONE resident headline batch (fresh processes differ by +-6 % in k_entity_stream through
the batch's physical placement - too noisy for effects of a few per cent): both .so files are dlopen'ed, the scoring call alternates
between them on the same tensors, the kernel classes are timed by each library's own profile.
    python tools/probes/inprocess_lib_ab.py drin_amd/libdrin_hip_prev.so [rounds] [--precision P] [--features bf16] [--workload wikidiverse]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from drin_amd import _lib, synth  # noqa: E402
from drin_amd.config import DrinConfig, wikimel_config  # noqa: E402
from drin_amd.model import Model  # noqa: E402

other = os.path.abspath(sys.argv[1])
rounds = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 4
precision = sys.argv[sys.argv.index("--precision") + 1] if "--precision" in sys.argv else "bf16x3"
bf16 = "--features" in sys.argv and sys.argv[sys.argv.index("--features") + 1] == "bf16"
dev = torch.device("cuda", 0)
wd = "--workload" in sys.argv and sys.argv[sys.argv.index("--workload") + 1] == "wikidiverse"
cfg = DrinConfig() if wd else wikimel_config()
sd = synth.make_state_dict(cfg, 7)
batch = synth.make_device_batch(cfg, 16384 if wd else 4096, 100, dev, dtype=torch.bfloat16 if bf16 else torch.float32)[:14]
libs = {}
shipped_path = _lib.LIB_PATH
for name, path in (("shipped", shipped_path), ("other", other)):
    _lib._lib, _lib.LIB_PATH = None, path
    libs[name] = _lib.load()
outs = {}
with torch.no_grad():
    for r in range(rounds):
        for name in ("shipped", "other"):
            _lib._lib = libs[name]
            model = Model(cfg, precision=precision).to(dev).eval()          # (its folded weights are built by the library in force)
            model.load_state_dict(sd)
            for _ in range(3):
                model(batch)
            _lib.profile_begin(1 << 12)
            for _ in range(8):
                outs[name] = model(batch)
            prof = _lib.profile_end()
            torch.cuda.synchronize()
            print(f"round {r} {name:8s}: " + "  ".join(f"{k} {v[0] / 8:.3f}" for k, v in prof.items() if v[0]), flush=True)
print("max |scores shipped - other|", float((outs["shipped"] - outs["other"]).abs().max()))
