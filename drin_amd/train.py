"""Lightning-free equivalent of the reference driver `train.py` for `model_type = "drin"`.

Reproduces the semantics a `train.py` user observes:
  * `seed_everything(seed)` before data and model construction (`train.py:134-136`);
  * `num_epoch // test_epoch_interval` rounds, each with a NEW Adam(lr) over all parameters - i.e. the
    optimizer state is reset every `test_epoch_interval` epochs (`train.py:55-56,112-122,141-144`) -
    followed by a test pass;
  * per step: `y_hat = model(batch[:-1])`, `TripletLoss(margin)(y, y_hat)`, top-k metrics updated on every
    step and reset at every epoch start, reported divided by `1 - acc_correction[split]`
    (`train.py:30-44,72-77`).
Data-parallel (one process per GPU, `torch.distributed`): every rank steps through its own shard of the
mentions; the only collectives are one all-reduce of the flat fp32 gradient bucket per step and the
metric counters / loss scalar.  Per-rank batch = the reference's batch (64), gradients
averaged; `global_batch_loss=True` instead all-gathers the scores so the TripletLoss couples the whole
global batch exactly like a single-process run on `world * batch` mentions would.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.distributed as dist
from torch import nn

from .config import DrinConfig
from .metrics import DeviceLossMetric, TopkAccuracy, TripletLoss


def _world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _collectives(force: bool) -> bool:
    """Whether the step's collectives run: with more than one rank always; `force`: also in a process group of ONE rank
    (a one-GPU box then drives the real RCCL code path - communicator, `ReduceOp.AVG`, stream ordering - instead of
    skipping it; `bench.py --force-collective`, `tests/test_gpu_round3.py`)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or force


class GradBucket:
    """The gradients of all parameters that receive one (26.8 MB of the 31.5 MB carry gradients),
    all-reduced once per step.

    `drin_amd.model.Model` writes its gradients into ONE flat bucket the `.grad`s are views of: the collective then runs
    on that bucket in place - no gather copy before, no scatter copy after.  Any other set of gradients (a foreign
    module, `grad_bucket=False`, gradients accumulated over several backward passes) is packed into a staging bucket.

    `overlap=True` (with a `drin_amd.model.Model`): the bucket is reduced in TWO pieces.  `drin_backward_staged` records an
    event once the GCN layers' gradients are complete (only the four vertex encoders' weight-gradient products run after
    it); the layers' piece - the bucket's tail, 9.4 of the 26.8 MB - is all-reduced behind that event on a side stream,
    under those products, and `allreduce_mean()` reduces the vertex encoders' piece and joins.  Element for element the
    same collective arithmetic, so the parameters follow the one-piece path bit for bit
    (`tests/test_dist_gloo.py`, `tests/test_gpu_round3.py`)."""

    def __init__(self, params: Sequence[nn.Parameter], force: bool = False, overlap: bool = False, model=None):
        self.params = [p for p in params if p.requires_grad]
        self.flat: Optional[torch.Tensor] = None          # staging bucket of the copy path
        self._view = None                                 # (storage ptr, lo, hi, flat view) of the in-place path
        self.in_place = False                             # what the last call did
        self.force = force
        self.collectives = 0                              # all_reduce calls issued so far
        self.overlapped = 0                               # steps whose layers' piece started inside backward
        self._early = None                                # (bucket ptr, split, work | None) of the piece in flight
        self._comm: Optional[torch.cuda.Stream] = None
        self.model = model if overlap else None
        if self.model is not None:
            if not hasattr(self.model, "_layers_ready_hook"):
                raise ValueError("overlap=True needs a drin_amd.model.Model (drin_backward_staged)")
            self.model._layers_ready_hook = self._layers_ready

    def close(self) -> None:
        """Detach from the model (its backward goes back to the one-stage drin_backward)."""
        if self.model is not None and self.model._layers_ready_hook == self._layers_ready:
            self.model._layers_ready_hook = None
        self.model = None

    def nbytes(self) -> int:
        live = [p for p in self.params if p.grad is not None]
        return 4 * sum(p.numel() for p in live) if live else 4 * sum(p.numel() for p in self.params)

    def _aliased_bucket(self, live) -> Optional[torch.Tensor]:
        """One flat tensor over the storage range the gradients occupy when they all live in a single fp32 storage
        (gaps - 256-byte slot padding - are all-reduced along: they hold zeros)."""
        st = live[0].grad.untyped_storage()
        base = st.data_ptr()
        lo, hi, elems = None, None, 0
        for p in live:
            g = p.grad
            if g.dtype != torch.float32 or not g.is_contiguous() or g.untyped_storage().data_ptr() != base:
                return None
            o = g.storage_offset()
            lo = o if lo is None else min(lo, o)
            hi = o + g.numel() if hi is None else max(hi, o + g.numel())
            elems += g.numel()
        if hi - lo > elems + 64 * len(live):               # not a bucket: unrelated views of one big storage
            return None
        if self._view is None or self._view[:3] != (base, lo, hi):
            flat = torch.empty(0, dtype=torch.float32, device=live[0].grad.device).set_(st, lo, (hi - lo,))
            self._view = (base, lo, hi, flat)
        return self._view[3]

    def _reduce(self, t: torch.Tensor, async_op: bool = False):
        avg = dist.get_backend() == "nccl"                 # RCCL averages inside the collective; gloo has no AVG
        self.collectives += 1
        work = dist.all_reduce(t, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, async_op=async_op)
        return work, (None if avg else dist.get_world_size())

    def _layers_ready(self, live_flat: torch.Tensor, split: int, ready: "torch.cuda.Event") -> None:
        """Called inside `_DrinScore.backward` right after `drin_backward_staged` was enqueued: `live_flat[split:]` (the GCN
        layers' gradients) is final once `ready` fires; the vertex encoders' products behind it write `live_flat[:split]` only."""
        if not _collectives(self.force) or self._early is not None or split <= 0 or split >= live_flat.numel():
            return
        piece = live_flat[split:]
        if live_flat.is_cuda:
            if self._comm is None or self._comm.device != live_flat.device:
                self._comm = torch.cuda.Stream(device=live_flat.device)
            self._comm.wait_event(ready)
            with torch.cuda.stream(self._comm):            # the collective is ordered behind the event, not behind the step's stream
                work, div = self._reduce(piece, async_op=True)
        else:                                              # host tensors (the gloo tests): the piece is final when the hook runs
            work, div = self._reduce(piece, async_op=True)
        self._early = (live_flat.data_ptr(), split, work, div, piece)

    def allreduce_mean(self) -> None:
        if not _collectives(self.force):
            self._early = None
            return
        live = [p for p in self.params if p.grad is not None]
        if not live:
            return
        flat = self._aliased_bucket(live)
        self.in_place = flat is not None
        early, self._early = self._early, None
        if early is not None:
            ptr, split, work, div, piece = early
            if flat is None or flat.data_ptr() != ptr or flat.numel() <= split:
                # the gradients did not end up as views of the bucket the early piece was taken from (accumulation over several
                # backward passes): nothing sound can be salvaged from a collective on memory that is no longer the gradient
                work.wait()
                raise RuntimeError("GradBucket(overlap=True): the gradients are not the flat bucket of this backward pass "
                                   "(gradient accumulation / two scoring calls in one graph): use overlap=False")
            _w, div_b = self._reduce(flat[:split])         # the vertex encoders' piece, on the step's stream
            work.wait()                                    # the step's stream now waits for the layers' piece
            if div is not None:                            # gloo: SUM, then the division - per piece, element for element
                piece.div_(div)
                flat[:split].div_(div_b)
            self.overlapped += 1
            return
        if flat is not None:
            _w, div = self._reduce(flat)
            if div is not None:
                flat.div_(div)
            return
        n = sum(p.numel() for p in live)
        if self.flat is None or self.flat.numel() != n or self.flat.device != live[0].grad.device:
            self.flat = torch.empty(n, dtype=torch.float32, device=live[0].grad.device)
        torch._foreach_copy_(list(self.flat.split([p.numel() for p in live])), [p.grad.reshape(-1) for p in live])
        _w, div = self._reduce(self.flat)
        if div is not None:
            self.flat.div_(div)
        torch._foreach_copy_([p.grad.reshape(-1) for p in live], list(self.flat.split([p.numel() for p in live])))


class OverlappedStep:
    """`bucket.allreduce_mean(); optimizer.step()` of step t on a SIDE stream, under the parameter-free head of step t + 1's
    forward (SURVEY.md 8e asks for no more than "overlap optional"; this is the part that costs nothing).  A WikiMEL step
    opens with 0.2 ms of pooling and static-edge kernels that read the batch alone (`ghmfc.py:54-60,245-249`,
    `model.py:41-45,60-94`); 26.8 MB of all-reduce over xGMI at 8 GPUs is 0.15-0.36 ms.  So after
    `loss.backward()` the step's stream records an event and goes on to the next batch; the side stream waits for it, runs the
    collective and the one-launch Adam, and records `model._params_ready`; `drin_forward_staged` makes the step's stream wait
    for THAT right before its first weight-reading launch.  Same kernels, same arithmetic, same bits as the serial loop
    (`tests/test_gpu_round3.py`, `tests/test_dist_gloo.py`).

    Whoever reads the parameters outside `model(...)` - evaluation code on another stream, `state_dict()` - calls `finish()`
    first (`MELRunner` does at the end of every training epoch).  Host tensors / no GPU: runs inline."""

    def __init__(self, model, bucket: GradBucket, optimizer):
        self.model, self.bucket, self.optimizer = model, bucket, optimizer
        self._comm: Optional[torch.cuda.Stream] = None
        self.steps = 0

    def run(self) -> None:
        """call right after `loss.backward()`"""
        p0 = next(self.model.parameters())
        self.steps += 1
        if not p0.is_cuda or not hasattr(self.model, "_params_ready"):
            self.bucket.allreduce_mean()
            self.optimizer.step()
            return
        main = torch.cuda.current_stream(p0.device)
        if self._comm is None or self._comm.device != p0.device:
            self._comm = torch.cuda.Stream(device=p0.device)
        self.model.wait_for_parameters()                   # (an update nobody consumed: keep the updates ordered)
        done = torch.cuda.Event()
        done.record(main)                                  # the backward pass, and with it the gradient bucket, is complete
        self._comm.wait_event(done)
        with torch.cuda.stream(self._comm):
            self.bucket.allreduce_mean()
            self.optimizer.step()
            ready = torch.cuda.Event()
            ready.record(self._comm)
        self.model._params_ready = ready

    def finish(self) -> None:
        """the current stream waits for the update in flight (if any)"""
        if hasattr(self.model, "wait_for_parameters"):
            self.model.wait_for_parameters()


class LibraryAdam:
    """`torch.optim.Adam(model.parameters(), lr)` of `train.py:55-56` (torch defaults) for a `drin_amd.model.Model`, as ONE
    launch of `drin_adam_step` over the model's flat parameter / gradient / moment buckets instead of torch's nine
    multi-tensor launches.  Same op sequence and per-op fp32 rounding as torch's default implementation, so a loop stepped
    with it follows the reference's loop bit for bit (`tests/test_gpu_round2.py::test_library_adam_matches_torch_adam_bitwise`).
    Like torch's Adam it skips parameters whose `.grad` is None, and like torch's it counts the steps PER PARAMETER
    (`state[p]["step"]`): a parameter that receives its first gradient at the optimiser's k-th step - unfrozen later - is
    bias-corrected for ITS first step.  `t` is the largest count.  `state_dict()` is this class's own (flat moment buckets +
    the per-parameter counts), not interchangeable with `torch.optim.Adam.state_dict()`."""

    def __init__(self, model, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        self.model, self.lr, self.betas, self.eps = model, float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.t = 0
        self.steps: List[int] = []                           # per parameter, `_param_list` order
        self.exp_avg: Optional[torch.Tensor] = None
        self.exp_avg_sq: Optional[torch.Tensor] = None
        self.one_launch_steps = 0
        self.param_groups = [{"params": list(model.parameters()), "lr": self.lr}]   # what callers of torch optimisers inspect

    def describe(self) -> str:
        return "library Adam: one drin_adam_step launch over the flat parameter bucket (torch.optim.Adam arithmetic)"

    def state_dict(self) -> dict:
        """Step count, hyper-parameters and the two moment buckets (laid out like `Model.bucket_layout`)."""
        return {"t": self.t, "steps": list(self.steps), "lr": self.lr, "betas": self.betas, "eps": self.eps,
                "exp_avg": None if self.exp_avg is None else self.exp_avg.clone(),
                "exp_avg_sq": None if self.exp_avg_sq is None else self.exp_avg_sq.clone()}

    def load_state_dict(self, state: dict) -> None:
        self.t, self.lr, self.betas, self.eps = int(state["t"]), float(state["lr"]), tuple(state["betas"]), float(state["eps"])
        self.param_groups[0]["lr"] = self.lr
        self.steps = [int(x) for x in state.get("steps", [])]
        dev = next(self.model.parameters()).device
        self.exp_avg = None if state["exp_avg"] is None else state["exp_avg"].to(dev).clone()
        self.exp_avg_sq = None if state["exp_avg_sq"] is None else state["exp_avg_sq"].to(dev).clone()

    def zero_grad(self, set_to_none: bool = True) -> None:
        for p in self.model.parameters():
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()

    @torch.no_grad()
    def step(self) -> None:
        import ctypes as C

        from . import _lib
        from .model import _param_list
        lib = _lib.load()
        model = self.model
        flat_p = model.flatten_parameters()
        offsets, live, total = model.bucket_layout()
        if self.exp_avg is None or self.exp_avg.numel() != total or self.exp_avg.device != flat_p.device:
            self.exp_avg, self.exp_avg_sq = torch.zeros_like(flat_p), torch.zeros_like(flat_p)
        b1, b2 = self.betas
        self.lr = float(self.param_groups[0]["lr"])          # honour a scheduler writing the group's lr, like torch's optimisers
        stream = torch.cuda.current_stream(flat_p.device).cuda_stream

        def launch(p_ptr, g_ptr, off, n, step):
            bias_correction1 = 1 - b1 ** float(step)         # torch/optim/adam.py (_multi_tensor_adam), python doubles
            bias_correction2 = 1 - b2 ** float(step)
            neg_step_size = (self.lr / bias_correction1) * -1
            scal = (1 - b1, b2, 1 - b2, bias_correction2 ** 0.5, self.eps, neg_step_size)
            _lib.check(lib.drin_adam_step(p_ptr, g_ptr, self.exp_avg.data_ptr() + 4 * off, self.exp_avg_sq.data_ptr() + 4 * off,
                                          n, *(C.c_float(x) for x in scal), stream))

        params = _param_list(model)
        if len(self.steps) != len(params):                   # a state from before the per-parameter counts: everyone at t
            self.steps = [self.t if (i < len(offsets) and offsets[i] < live) else 0 for i in range(len(params))]
        # the kernel writes the parameters behind PyTorch's back: bump their version counters (no launch), so that everything
        # keyed on them - the folded weights of the fused inference path, the per-entity cache, autograd's saved-tensor
        # checks - sees the update exactly as it sees torch.optim.Adam's
        torch.autograd.graph.increment_version(params)
        with_grad = [i for i, p in enumerate(params) if p.grad is not None]
        for i in with_grad:
            self.steps[i] += 1
        self.t = max(self.steps) if self.steps else 0
        gflat = model.grad_bucket()
        live_set = [i for i, o in enumerate(offsets) if o < live]
        if gflat is not None and with_grad == live_set and len({self.steps[i] for i in live_set}) == 1:
            launch(flat_p.data_ptr(), gflat.data_ptr(), 0, live, self.steps[live_set[0]])   # the whole live prefix, slot padding included (zeros)
            self.one_launch_steps += 1
            return
        for i in with_grad:                                               # irregular step: one launch per tensor, its own step count
            p, g = params[i], params[i].grad
            if g.dtype != torch.float32 or not g.is_contiguous():
                g = g.to(torch.float32).contiguous()
            launch(p.data_ptr(), g.data_ptr(), offsets[i], p.numel(), self.steps[i])


def make_adam(model, lr: float, library: Optional[bool] = None, capturable: bool = False, fused: bool = False):
    """The optimiser of `train.py:55-56`.  `library` (default: when `model` is a `drin_amd.model.Model` on a GPU): the
    one-launch `LibraryAdam`; else `torch.optim.Adam` with its defaults (`fused=True` opts into torch's single-kernel
    variant, whose rounding differs)."""
    from .model import Model
    p0 = next(model.parameters())
    can = isinstance(model, Model) and p0.is_cuda and not capturable and not fused
    if library is None:
        library = can
    if library and not can:
        raise ValueError("LibraryAdam steps a drin_amd.model.Model on a GPU (no capturable / fused variants)")
    if library:
        return LibraryAdam(model, lr)
    return torch.optim.Adam(model.parameters(), lr=lr, capturable=capturable, **({"fused": True} if fused else {}))


class _GatherScores(torch.autograd.Function):
    """all_gather of the [B_local, N] scores with the matching reduce-scatter-free backward (each rank
    keeps the slice of the gradient that belongs to its own mentions; the loss is replicated)."""

    @staticmethod
    def forward(ctx, x):
        world = _world()                                     # (a process group of one rank gathers its own slice)
        parts = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(parts, x.contiguous())
        ctx.rows = x.shape[0]
        return torch.cat(parts, 0)

    @staticmethod
    def backward(ctx, g):
        r = _rank()
        return g[r * ctx.rows: (r + 1) * ctx.rows]


@dataclass
class StepLog:
    loss: float
    topk: List[float]


@dataclass
class History:
    train: List[StepLog] = field(default_factory=list)
    valid: List[StepLog] = field(default_factory=list)
    test: List[StepLog] = field(default_factory=list)
    seconds: float = 0.0


class StepProfiler:
    """The `profiling` switch of `train.py:64-70,84-85,101-109` (`args.py:131`): the reference wraps the fit in
    `torch.profiler.profile(schedule(wait=1, warmup=1, active=3, repeat=2), tensorboard_trace_handler("log/profiler"))`,
    started at fit start, stepped after every training batch, stopped at the end of the training epoch.  Same schedule
    here, over the library's own profiler (`drin_profile_begin/end`: HIP events around every launch, on the launch stream):
    each of the `repeat` cycles skips `wait` steps, runs `warmup` more, then times `active` steps and writes
    `<out_dir>/drin_profile_<cycle>.json` - kernel class -> ms per step and launches per step - and logs the line.
    Launch names for rocprofv3 come from `DRIN_ROCTX=1` (roctx ranges over the entry points)."""

    def __init__(self, wait: int = 1, warmup: int = 1, active: int = 3, repeat: int = 2, out_dir: str = "log/profiler",
                 log: Optional[Callable[[str], None]] = None):
        self.wait, self.warmup, self.active, self.repeat = wait, warmup, active, repeat
        self.out_dir, self.log = out_dir, log
        self.step_idx, self.cycle, self.open = 0, 0, False
        self.reports: List[dict] = []

    def start(self) -> None:
        self.step_idx, self.cycle = 0, 0
        self._maybe_open()

    def _maybe_open(self) -> None:
        from . import _lib
        if self.cycle < self.repeat and self.step_idx == self.wait + self.warmup and not self.open:
            _lib.profile_begin(1 << 16)
            self.open = True

    def step(self) -> None:
        """after every training batch (`on_train_batch_end`)"""
        self.step_idx += 1
        if self.open and self.step_idx == self.wait + self.warmup + self.active:
            self._close()
            self.step_idx = 0
            self.cycle += 1
        self._maybe_open()

    def _close(self) -> None:
        import json
        import os

        from . import _lib
        prof = _lib.profile_end()
        self.open = False
        steps = max(self.step_idx - self.wait - self.warmup, 1)
        report = {"cycle": self.cycle, "steps": steps,
                  "kernel_ms_per_step": {k: v[0] / steps for k, v in prof.items() if v[1]},
                  "launches_per_step": {k: v[1] / steps for k, v in prof.items() if v[1]}}
        self.reports.append(report)
        if _rank() == 0:
            os.makedirs(self.out_dir, exist_ok=True)
            with open(os.path.join(self.out_dir, f"drin_profile_{self.cycle}.json"), "w") as f:
                json.dump(report, f, indent=1)
            if self.log:
                self.log("profile cycle %d: %s" % (self.cycle, {k: round(v, 4) for k, v in report["kernel_ms_per_step"].items()}))

    def stop(self) -> None:
        """end of the training epoch (`on_train_epoch_end`): an unfinished cycle reports what it has"""
        if self.open:
            self._close()
        self.cycle = self.repeat


class MELRunner:
    """`MELModel` of `train.py:20-56` without Lightning."""

    def __init__(self, cfg: DrinConfig, model: nn.Module, device, global_batch_loss: bool = False,
                 log: Optional[Callable[[str], None]] = None, entity_table=None, device_loss: Optional[bool] = None,
                 fused_adam: bool = False, library_adam: Optional[bool] = None, output_test_result: Optional[str] = None,
                 profiling: bool = False, profile_dir: str = "log/profiler", force_collectives: bool = False,
                 overlap_allreduce: Optional[bool] = None):
        """`entity_table`: a device-resident `drin_amd.model.EntityTable`; the loaders then yield the 11-item
        table-form batches of `drin_amd.data.IndexedMELData` (candidate indices instead of gathered features).
        `device_loss`: loss + top-k counters through the library's `drin_triplet_topk` (default on a GPU; the
        gathered global-batch loss keeps the torch classes).
        `library_adam`: the one-launch `LibraryAdam` (default for a `drin_amd.model.Model` on a GPU), else `torch.optim.Adam`.
        `output_test_result`: a path - the per-sample dump of `train.py:16-17,40-43` (`args.output_test_result`): every
        test-split sample's score row and answer row, `"{index}:\t{scores}\n{answer}\n"` (rank r > 0 of a data-parallel
        run appends `.rank{r}` to the name and numbers its own shard's samples).
        `profiling`: the switch of `train.py:64-70` (`args.profiling`), see `StepProfiler` (GPU only).
        `force_collectives`: run the step's collectives (gradient all-reduce, score gather, metric sync) in a process
        group of ONE rank too - what a one-GPU box can exercise of the N-GPU path.
        `overlap_allreduce`: how the gradient all-reduce leaves the critical path when collectives run with the HIP `Model`
        on a GPU - "forward" (default then): `OverlappedStep`, collective + Adam on a side stream under the next step's
        parameter-free head; "backward": `GradBucket(overlap=True)`, the GCN layers' piece started inside the staged
        backward; "both"; False / "none": serial."""
        self.cfg, self.model, self.device = cfg, model, torch.device(device)
        self.entity_table = entity_table
        self.fused_adam, self.library_adam = fused_adam, library_adam
        self.result_file = None
        self._result_index = 0
        if output_test_result:
            self._result_path = output_test_result + (f".rank{_rank()}" if _rank() > 0 else "")
            self.result_file = open(self._result_path, "w")
        self.loss = TripletLoss(cfg.triplet_margin)
        self.metrics = [TopkAccuracy(k, self.device) for k in cfg.metrics_topk]
        if device_loss and global_batch_loss:
            raise ValueError("the gathered global-batch loss runs through the torch TripletLoss; pass device_loss=False")
        if device_loss is None:
            device_loss = self.device.type == "cuda" and not global_batch_loss
        self.device_loss = DeviceLossMetric(cfg.triplet_margin, cfg.metrics_topk, self.device) if device_loss else None
        self.force_collectives = force_collectives
        can = (self.device.type == "cuda" and hasattr(model, "_layers_ready_hook") and getattr(model, "grad_bucket_enabled", False))
        if overlap_allreduce is None:
            overlap_allreduce = "forward" if (can and _collectives(force_collectives)) else "none"
        mode = {True: "backward", False: "none"}.get(overlap_allreduce, overlap_allreduce)
        if mode not in ("none", "forward", "backward", "both"):
            raise ValueError(f"overlap_allreduce={overlap_allreduce!r}")
        if mode != "none" and not can:
            raise ValueError("overlap_allreduce needs the HIP Model (grad_bucket=True) on a GPU")
        self.overlap_mode = mode
        staged_bwd = mode in ("backward", "both")
        self.bucket = GradBucket(list(model.parameters()), force=force_collectives, overlap=staged_bwd,
                                 model=model if staged_bwd else None)
        self._pipe = None
        self.global_batch_loss = global_batch_loss
        self.log = log
        if profiling and self.device.type != "cuda":
            raise ValueError("profiling times the library's launches: it needs the model on a GPU")
        self.profiler = StepProfiler(out_dir=profile_dir, log=log) if profiling else None
        self._epoch = None                                             # (current, last) for the epoch banners; set by fit()

    def _to_device(self, batch):
        return [t.to(self.device, non_blocking=True) for t in batch]

    def forward_step(self, batch, split: int, batch_idx: int = 0):
        """`_forward_step` (`train.py:30-44`)."""
        batch = self._to_device(batch)
        y = batch[-1]
        if self.entity_table is not None:
            from .model import IndexedBatch
            y_hat = self.model(IndexedBatch(batch[:7], self.entity_table, batch[7], batch[8], batch[9]))
        else:
            y_hat = self.model(batch[:-1])
        if self.result_file is not None and split == 2:                # train.py:40-43
            # (the reference numbers a sample `i + batch_idx * batch_size` with the LOADER's batch size: a running count
            #  is that number whatever batch size the loaders were built with)
            for i, sample in enumerate(y_hat.detach().to("cpu").tolist()):
                self.result_file.write(f"{self._result_index + i}:\t{sample}\n{y[i]}\n")
            self._result_index += y_hat.shape[0]
            self.result_file.flush()
        # (evaluation shards may differ in length by one mention across ranks - no padding, so that the metrics count every
        #  mention exactly once - hence no per-step collective there: the gathered loss is a training-step construct)
        if self.global_batch_loss and _collectives(self.force_collectives) and self.model.training:
            world = _world()
            ys = [torch.empty_like(y) for _ in range(world)]
            dist.all_gather(ys, y.contiguous())
            # the gathered loss is the GLOBAL mean; scale so that averaging grads over ranks reproduces it
            loss = self.loss(torch.cat(ys, 0), _GatherScores.apply(y_hat)) * world
        elif self.device_loss is not None:
            return self.device_loss(y, y_hat)                          # loss and all top-k counters in one call
        else:
            loss = self.loss(y, y_hat)
        with torch.no_grad():
            for m in self.metrics:
                m.update(y_hat.detach(), y)
        return loss

    def _topk(self, split: int) -> List[float]:
        if self.device_loss is not None:
            return [a / (1 - self.cfg.acc_correction[split]) for a in self.device_loss.accuracies()]
        return [float(m.compute()) / (1 - self.cfg.acc_correction[split]) for m in self.metrics]

    def run_epoch(self, loader, split: int, optimizer: Optional[torch.optim.Optimizer]) -> StepLog:
        meters = [self.device_loss] if self.device_loss is not None else self.metrics
        for m in meters:                                               # EpochLogger.epoch_start (train.py:72-77)
            m.reset()
        if self.log and _rank() == 0 and self._epoch is not None:
            from datetime import datetime
            self.log(f"\n***** Epoch {self._epoch[0]}/{self._epoch[1]} - {('training', 'validating', 'testing')[split]} - {datetime.now()}")
        if split == 2 and self.result_file is not None:               # on_test_epoch_start (train.py:92-95)
            if self.result_file.closed:                                # a test pass after fit() closed the dump: continue it
                self.result_file = open(self._result_path, "a")
            self.result_file.write("==========  Test ==========\n")
            self._result_index = 0
        training = optimizer is not None
        self.model.train(training)
        total, steps = torch.zeros((), dtype=torch.float64, device=self.device), 0   # summed on the device: no per-step read-back
        for batch_idx, batch in enumerate(loader):
            if training:
                optimizer.zero_grad(set_to_none=True)
                loss = self.forward_step(batch, split, batch_idx)
                loss.backward()
                if self.overlap_mode in ("forward", "both"):
                    if self._pipe is None or self._pipe.optimizer is not optimizer:
                        self._pipe = OverlappedStep(self.model, self.bucket, optimizer)
                    self._pipe.run()                                   # collective + Adam on the side stream; the next forward waits
                else:
                    self.bucket.allreduce_mean()
                    optimizer.step()
                if self.profiler is not None:
                    self.profiler.step()                               # on_train_batch_end (train.py:105-109)
            else:
                with torch.no_grad():
                    loss = self.forward_step(batch, split, batch_idx)
            total += loss.detach()
            steps += 1
        if self._pipe is not None:
            self._pipe.finish()                                        # the last update lands before anyone else reads the weights
        if training and self.profiler is not None:
            self.profiler.stop()                                       # on_train_epoch_end (train.py:84-85)
        for m in meters:
            m.sync(force=self.force_collectives)
        mean_loss = float(total) / max(steps, 1)
        if hasattr(self.model, "check_indices"):
            self.model.check_indices()            # the loop has just synchronised: a candidate row outside the tables raises here at the latest
        if _collectives(self.force_collectives):
            t = torch.tensor([mean_loss], device=self.device)
            dist.all_reduce(t)
            mean_loss = float(t) / _world()
        return StepLog(mean_loss, self._topk(split))

    def close(self) -> None:
        """End of `main()` (`train.py:145-146`): the test-result dump is closed."""
        if self.result_file is not None and not self.result_file.closed:
            self.result_file.close()
        self.bucket.close()

    def test(self, loader) -> StepLog:
        """`trainer.test(model, datasets[2])` (`train.py:137-140,144`): one pass over the test split."""
        self._epoch = self._epoch or (1, self.cfg.num_epoch)
        return self.run_epoch(loader, 2, None)

    def fit(self, loaders, num_epoch: Optional[int] = None, test_epoch_interval: Optional[int] = None) -> History:
        """`main()` of `train.py:141-144`."""
        cfg = self.cfg
        num_epoch = cfg.num_epoch if num_epoch is None else num_epoch
        interval = cfg.test_epoch_interval if test_epoch_interval is None else test_epoch_interval
        hist = History()
        t0 = time.perf_counter()
        epoch = 0
        for _round in range(num_epoch // interval):
            # configure_optimizers (train.py:55-56), per Trainer.  `fused_adam` (opt-in): torch's single-kernel implementation
            # of the same update - 0.27 ms less per step, but its rounding differs from the default's, and Adam's division
            # by sqrt(v) amplifies that: after two epochs the loss is 1.4e-3 off the reference loop instead of 1e-4
            # LibraryAdam (default for the HIP Model on a GPU): the same update as torch's default, bit for bit, in one launch
            optimizer = make_adam(self.model, cfg.learning_rate, library=False if self.fused_adam else self.library_adam,
                                  fused=self.fused_adam)
            if self.profiler is not None:
                self.profiler.start()                                  # on_fit_start (train.py:101-103), per Trainer
            for _ in range(interval):
                sampler = getattr(loaders[0], "sampler", None)
                if hasattr(sampler, "set_epoch"):
                    sampler.set_epoch(epoch)
                self._epoch = (epoch + 1, num_epoch)
                tr = self.run_epoch(loaders[0], 0, optimizer)
                va = self.run_epoch(loaders[1], 1, None)
                hist.train.append(tr)
                hist.valid.append(va)
                epoch += 1
                if self.log and _rank() == 0:
                    self.log(f"epoch {epoch}/{num_epoch} train loss {tr.loss:.5f} top-k {tr.topk} | valid loss {va.loss:.5f} top-k {va.topk}")
            te = self.run_epoch(loaders[2], 2, None)
            hist.test.append(te)
            if self.log and _rank() == 0:
                self.log(f"test after epoch {epoch}: loss {te.loss:.5f} top-k {te.topk}")
        hist.seconds = time.perf_counter() - t0
        if self.result_file is not None:                               # train.py:145-146
            self.result_file.close()
        return hist


def seed_everything(seed: int) -> None:
    """`pl.seed_everything` (`train.py:134`): python, numpy and torch generators."""
    import random

    import numpy as np

    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def main(argv: Optional[Sequence[str]] = None) -> None:
    """python -m drin_amd.train --data DIR [--dataset wikidiverse|wikimel] [--epochs E] [--interval I] [--on-device]
    [--precision bf16x3|f32] [--torch-adam] [--output-test-result PATH] [--profiling] [--test-only] [--load-state PATH]
    [--force-collectives]; N GPUs: python -m torch.distributed.run --nproc-per-node N -m drin_amd.train ..."""
    import argparse
    import os

    from .config import wikimel_config
    from .data import create_datasets
    from .model import Model

    ap = argparse.ArgumentParser()
    ap.add_argument("--data", required=True)
    ap.add_argument("--dataset", default="wikidiverse")
    ap.add_argument("--epochs", type=int, default=None)
    ap.add_argument("--interval", type=int, default=None)
    ap.add_argument("--batch-size", type=int, default=None)
    ap.add_argument("--workers", type=int, default=0)
    ap.add_argument("--global-batch-loss", action="store_true")
    ap.add_argument("--precision", default="bf16x3", choices=["bf16x3", "f32"], help="contraction arithmetic of the HIP Model")
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam instead of the one-launch LibraryAdam (same arithmetic)")
    ap.add_argument("--output-test-result", default=None, metavar="PATH",
                    help="per-sample dump of the test split (args.output_test_result, train.py:16-17,40-43)")
    ap.add_argument("--profiling", action="store_true",
                    help="args.profiling (train.py:64-70): wait 1 / warmup 1 / active 3 steps x 2 cycles per fit through the library's "
                         "profiler, kernel class -> ms per step written to log/profiler/drin_profile_<cycle>.json")
    ap.add_argument("--test-only", action="store_true",
                    help="args.test_only (train.py:137-140, args.py:112): no fit, one pass over the test split with the freshly "
                         "initialised (or --load-state'd) model")
    ap.add_argument("--load-state", default=None, metavar="PATH", help="a torch-saved state_dict (the reference's keys) loaded before fit / test")
    ap.add_argument("--force-collectives", action="store_true",
                    help="initialise a process group even for ONE rank and run the step's collectives in it (RCCL exercised on a one-GPU box)")
    ap.add_argument("--on-device", action="store_true",
                    help="every split (and, wikimel, the entity tables) resident on the GPU (create_device_splits, load_entity_table): "
                         "no host gather, no host-to-device copy in the step")
    a = ap.parse_args(argv)
    cfg = wikimel_config() if a.dataset == "wikimel" else DrinConfig()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1 or a.force_collectives:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:                        # a free port: two such runs on one host must not collide
                import socket
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        dist.init_process_group("nccl", device_id=dev, rank=int(os.environ.get("RANK", "0")), world_size=world)
    seed_everything(cfg.seed)
    table = None
    if a.on_device:
        from .data import create_device_splits, load_entity_table
        table = load_entity_table(cfg, a.data, dev, entity_mmap="r") if cfg.dataset_name == "wikimel" else None
        loaders = create_device_splits(cfg, a.data, dev, a.batch_size, _rank(), _world(), mention_mmap="r")
    else:
        loaders = create_datasets(cfg, a.data, a.batch_size, a.workers, _rank(), _world())
    if _rank() == 0:                                                   # train.py:126-133: every setting, strings quoted
        import dataclasses
        print("=============== parameters ===============")
        for f in sorted(dataclasses.fields(cfg), key=lambda f: f.name):
            v = getattr(cfg, f.name)
            print(f.name, "'" + v + "'" if isinstance(v, str) else v, sep=" = ")
    model = Model(cfg, precision=a.precision).to(dev)
    if a.load_state:
        model.load_state_dict(torch.load(a.load_state, map_location=dev))
    runner = MELRunner(cfg, model, dev, a.global_batch_loss, log=print, entity_table=table,
                       library_adam=False if a.torch_adam else None, output_test_result=a.output_test_result,
                       profiling=a.profiling, force_collectives=a.force_collectives)
    if a.test_only:                                                    # train.py:137-140
        te = runner.test(loaders[2])
        if _rank() == 0:
            print(f"test: loss {te.loss:.5f} top-k {te.topk}")
    else:
        runner.fit(loaders, a.epochs, a.interval)
        if _rank() == 0:
            print("Training completed")                                # train.py:147
    runner.close()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
