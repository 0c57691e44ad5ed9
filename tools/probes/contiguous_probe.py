"""Does PHYSICALLY CONTIGUOUS device memory (hipExtMallocWithFlags(hipDeviceMallocContiguous)) take the run-to-run spread out of
the stream kernel?  One process, alternating: the 81 GB entity-token tensor of the headline batch in a contiguous allocation /
in torch's ordinary one; the library profiler's kernel classes for each.  GPU only."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from drin_amd import _lib, synth
from drin_amd.config import wikimel_config
from drin_amd.model import Model

hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipFree.argtypes = [C.c_void_p]


class Raw:
    def __init__(self, nbytes, flags):
        p = C.c_void_p()
        err = hip.hipExtMallocWithFlags(C.byref(p), nbytes, flags)
        if err != 0:
            raise RuntimeError(f"hipExtMallocWithFlags({nbytes}, {flags}) -> {err}")
        self.ptr, self.n = p.value, nbytes // 4
        self.__cuda_array_interface__ = {"shape": (self.n,), "typestr": "<f4", "data": (self.ptr, False), "version": 2}

    def free(self):
        hip.hipFree(C.c_void_p(self.ptr))


dev = torch.device("cuda:0")
cfg = wikimel_config()
model = Model(cfg).to(dev).eval()
B = 4096
for trial, flags in enumerate([4, 0, 4, 0, 4, 0]):
    batch = synth.make_device_batch(cfg, B, 100, dev)[:14]
    shape = batch[7].shape
    batch[7] = None
    torch.cuda.empty_cache()
    raw = Raw(4 * shape.numel(), flags)
    t = torch.as_tensor(raw, device=dev).view(shape)
    for i in range(0, B, 256):
        t[i:i + 256].normal_()
    batch[7] = t
    with torch.no_grad():
        for _ in range(2):
            model(batch)
        _lib.profile_begin(1 << 12)
        for _ in range(4):
            model(batch)
        prof = _lib.profile_end()
    torch.cuda.synchronize()
    ms = {k: round(v[0] / 4, 3) for k, v in prof.items() if v[0] > 0}
    print(f"trial {trial} flags {flags} ({'contiguous' if flags == 4 else 'default'}): entity_text at {raw.ptr:#x}  {ms}", flush=True)
    del batch, t
    torch.cuda.empty_cache()
    raw.free()
