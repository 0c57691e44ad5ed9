"""Does torch.cuda.Event.cuda_event hand out a hipEvent_t the HIP runtime accepts?  (probe for drin_backward_staged)"""
import ctypes as C
import torch

hip = C.CDLL("libamdhip64.so")
hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
hip.hipEventRecord.restype = C.c_int
hip.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
hip.hipStreamWaitEvent.restype = C.c_int
torch.zeros(1, device="cuda")
ev = torch.cuda.Event()
print("before record: cuda_event =", ev.cuda_event, flush=True)
ev.record(torch.cuda.current_stream())
h = ev.cuda_event
print("after record: cuda_event =", h, hex(h) if h else None, flush=True)
st = torch.cuda.current_stream().cuda_stream
print("stream handle", st, flush=True)
print("hipEventRecord ->", hip.hipEventRecord(h, st), flush=True)
side = torch.cuda.Stream()
side.wait_event(ev)
torch.cuda.synchronize()
print("ok", flush=True)
