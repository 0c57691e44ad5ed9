"""world_size-2 data-parallel path on CPU (gloo): gradient bucket all-reduce, metric counter sync and
the optional global-batch TripletLoss, against single-process results on the same mentions."""
import os
import socket
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from drin_amd import synth
from drin_amd.config import DrinConfig
from drin_amd.metrics import TopkAccuracy
from drin_amd.train import GradBucket, MELRunner
from oracle import drin_oracle as O
from oracle.cases import TINY
from tests.helpers import OracleModel

CFG = DrinConfig(batch_size=4, **TINY)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, global_loss):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    try:
        model = OracleModel(CFG)
        model.load_state_dict(synth.make_state_dict(CFG, 8))
        batch = synth.make_batch(CFG, 8, 77)
        shard = [t[rank * 4:(rank + 1) * 4] for t in batch]          # contiguous mention ranges per rank
        runner = MELRunner(CFG, model, "cpu", global_batch_loss=global_loss)
        loss = runner.forward_step(shard, 0)
        loss.backward()
        runner.bucket.allreduce_mean()
        for m in runner.metrics:
            m.sync()
        torch.save({"grads": {k: (p.grad.clone() if p.grad is not None else None) for k, p in model.named_parameters()},
                    "loss": loss.item(), "counts": [(int(m.correct), int(m.total)) for m in runner.metrics]},
                   os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def _single_process(global_loss):
    model = OracleModel(CFG)
    model.load_state_dict(synth.make_state_dict(CFG, 8))
    batch = synth.make_batch(CFG, 8, 77)
    if global_loss:
        loss = O.triplet_loss(batch[-1], model(batch[:-1]), CFG.triplet_margin)
        loss.backward()
        grads = {k: p.grad.clone() if p.grad is not None else None for k, p in model.named_parameters()}
    else:
        acc = None
        for r in range(2):
            model.zero_grad()
            shard = [t[r * 4:(r + 1) * 4] for t in batch]
            O.triplet_loss(shard[-1], model(shard[:-1]), CFG.triplet_margin).backward()
            g = {k: p.grad.clone() if p.grad is not None else None for k, p in model.named_parameters()}
            acc = g if acc is None else {k: (acc[k] + g[k]) if g[k] is not None else None for k in g}
        grads = {k: v / 2 if v is not None else None for k, v in acc.items()}
    scores = model(batch[:-1]).detach()
    counts = [O.topk_counts(scores, batch[-1], k) for k in CFG.metrics_topk]
    return grads, counts


@pytest.mark.parametrize("global_loss", [False, True], ids=["per_rank_batch", "global_batch_loss"])
def test_two_rank_gradients_match_single_process(global_loss):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, _free_port(), d, global_loss), nprocs=2, join=True)
        r0, r1 = (torch.load(os.path.join(d, f"rank{r}.pt")) for r in range(2))
    ref_grads, ref_counts = _single_process(global_loss)
    for k, ref in ref_grads.items():
        a, b = r0["grads"][k], r1["grads"][k]
        assert (a is None) == (ref is None), k
        if ref is None:
            continue
        assert torch.equal(a, b), f"ranks disagree after all-reduce: {k}"
        assert torch.allclose(a, ref, atol=1e-6, rtol=1e-4), k
    assert r0["counts"] == r1["counts"] == ref_counts   # dist_reduce_fx="sum" (common/utils.py:57-58)


def test_grad_bucket_is_identity_without_process_group():
    m = OracleModel(CFG)
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    GradBucket(list(m.parameters())).allreduce_mean()
    assert all(torch.equal(p.grad, torch.ones_like(p)) for p in m.parameters())
