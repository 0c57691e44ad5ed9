"""Caller-side loss and metric of `train.py:30-44`.

`TripletLoss` / `TopkAccuracy`: plain torch on whatever device the scores live on (the reference's classes
without torchmetrics).  `DeviceLossMetric`: both at once through `drin_triplet_topk` of libdrin_hip.so
 - one call per step, loss and counters stay on the device, no host read-back."""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import torch


class TripletLoss:
    """`common/utils.py:26-43`.  y_true: one-hot [B, N-1] (uint8), y_pred: scores [B, N].

    Each mention's positive distance is compared with the WHOLE batch's [B, N-1] distance matrix
    (`utils.py:41-42` broadcasts `positive_val[i] - y_pred`), gold column included; the Python loop
    over mentions of the reference is one broadcast here.
    """

    def __init__(self, margin: float):
        self.margin = margin

    def __call__(self, y_true: torch.Tensor, y_pred: torch.Tensor) -> torch.Tensor:
        if y_pred.shape[1] != y_true.shape[1]:
            y_pred = y_pred[:, :-1]                                   # utils.py:36-37 drops the answer slot
        y_pred = -y_pred
        positive_val = torch.sum(y_pred * y_true, dim=-1)             # utils.py:39
        per_mention = torch.clamp(positive_val[:, None, None] - y_pred[None] + self.margin, min=0).mean(dim=(1, 2))
        return per_mention.sum() / y_true.shape[0]                    # utils.py:43


class TopkAccuracy:
    """`common/utils.py:46-73` without torchmetrics: `correct` / `total` are plain int64 tensors; with
    more than one rank `sync()` sums them (the reference's `dist_reduce_fx="sum"`, utils.py:57-58)."""

    def __init__(self, top_k: int, device="cpu"):
        self.top_k = top_k
        self.correct = torch.zeros((), dtype=torch.int64, device=device)
        self.total = torch.zeros((), dtype=torch.int64, device=device)

    def update(self, y_pred: torch.Tensor, y_true: torch.Tensor) -> None:
        if y_pred.shape[1] != y_true.shape[1]:
            y_pred = y_pred[:, :-1]
        lower = torch.topk(y_pred, self.top_k)[0][:, -1:]             # utils.py:63
        self.correct += torch.sum(y_true * (y_pred >= lower)).to(torch.int64)   # ties count (utils.py:64-65)
        self.total += y_true.shape[0]

    def compute(self) -> torch.Tensor:
        return self.correct / self.total

    def reset(self) -> None:
        self.correct.zero_()
        self.total.zero_()

    def sync(self, force: bool = False) -> None:
        """`force`: also in a process group of one rank (the collective code path exercised on a one-GPU box)."""
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force):
            packed = torch.stack([self.correct, self.total])
            dist.all_reduce(packed, op=dist.ReduceOp.SUM)
            self.correct, self.total = packed[0], packed[1]


def corrected_topk(metrics: Sequence[TopkAccuracy], acc_correction: float):
    """The value `train.py:38` prints: accuracy / (1 - first-stage miss rate)."""
    return [float(m.compute()) / (1 - acc_correction) for m in metrics]


class _TripletTopk(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y_pred: torch.Tensor, y_true: torch.Tensor, margin: float, topk, correct):
        from . import _lib
        if not y_pred.is_cuda:
            raise RuntimeError("drin_amd: the fused loss/metric runs on the AMD GPU only (no CPU fallback); "
                               "use TripletLoss / TopkAccuracy for host tensors")
        lib = _lib.load()
        scores = y_pred.detach().contiguous().float()
        answer = y_true.contiguous()
        if answer.dtype != torch.uint8:
            answer = answer.to(torch.uint8)
        B, N = scores.shape
        if answer.shape != (B, N - 1):
            raise ValueError(f"answer {tuple(answer.shape)} does not match scores {tuple(scores.shape)} (expected [B, N-1])")
        need_grad = y_pred.requires_grad
        loss = torch.empty(1, dtype=torch.float32, device=scores.device)
        dscores = torch.empty_like(scores) if need_grad else None
        ws = torch.empty(lib.drin_loss_workspace_bytes(B), dtype=torch.uint8, device=scores.device)
        ks = (C.c_int32 * len(topk))(*topk)
        _lib.check(lib.drin_triplet_topk(
            scores.data_ptr(), answer.data_ptr(), B, N, float(margin), ks, len(topk), loss.data_ptr(),
            dscores.data_ptr() if need_grad else None, correct.data_ptr() if len(topk) else None,
            ws.data_ptr(), ws.numel(), torch.cuda.current_stream(scores.device).cuda_stream))
        ctx.dscores = dscores
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        return (ctx.dscores * g if ctx.dscores is not None else None), None, None, None, None


class DeviceLossMetric:
    """`TripletLoss(margin)(y, y_hat)` and every `TopkAccuracy(k).update(y_hat, y)` of `train.py:34-37` as ONE
    library call.  Returns the loss (differentiable w.r.t. the scores); `correct[k]` / `total` accumulate on
    the device until `reset()`; `accuracies()` is the only host read-back (once per epoch)."""

    def __init__(self, margin: float, top_k: Sequence[int], device):
        if len(top_k) > 8:
            raise ValueError("at most 8 top-k values per call")
        self.margin, self.top_k = margin, [int(k) for k in top_k]
        self.correct = torch.zeros(len(self.top_k), dtype=torch.int64, device=device)
        self.total = 0

    def __call__(self, y_true: torch.Tensor, y_pred: torch.Tensor, count: bool = True) -> torch.Tensor:
        loss = _TripletTopk.apply(y_pred, y_true, self.margin, self.top_k if count else [], self.correct)
        if count:
            self.total += y_true.shape[0]
        return loss

    def reset(self) -> None:
        self.correct.zero_()
        self.total = 0

    def sync(self, force: bool = False) -> None:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force):
            packed = torch.cat([self.correct, torch.tensor([self.total], dtype=torch.int64, device=self.correct.device)])
            dist.all_reduce(packed, op=dist.ReduceOp.SUM)
            self.correct, self.total = packed[:-1].clone(), int(packed[-1])

    def accuracies(self) -> List[float]:
        return [c / max(self.total, 1) for c in self.correct.tolist()]
