"""Caller-side loss and metric of `train.py:30-44`, in plain torch on
whatever device the scores live on.  These are not kernel targets: [B, N] scalars per step."""
from __future__ import annotations

from typing import Sequence

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

    def sync(self) -> None:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            packed = torch.stack([self.correct, self.total])
            dist.all_reduce(packed, op=dist.ReduceOp.SUM)
            self.correct, self.total = packed[0], packed[1]


def corrected_topk(metrics: Sequence[TopkAccuracy], acc_correction: float):
    """The value `train.py:38` prints: accuracy / (1 - first-stage miss rate)."""
    return [float(m.compute()) / (1 - acc_correction) for m in metrics]
