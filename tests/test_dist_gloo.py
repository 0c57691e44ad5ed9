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


# ---- the EPOCH loop under data parallelism with n % (world * batch) != 0 (ranks must never disagree on the
# number of steps or on the size of the last batch, or the per-step collectives pair up wrongly / hang) -------------------
def _epoch_worker(rank, world, port, root, out_dir, global_loss):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    try:
        from drin_amd.data import create_datasets
        model = OracleModel(CFG)
        model.load_state_dict(synth.make_state_dict(CFG, 8))
        loaders = create_datasets(CFG, root, rank=rank, world_size=world)
        runner = MELRunner(CFG, model, "cpu", global_batch_loss=global_loss)
        opt = torch.optim.Adam(model.parameters(), lr=CFG.learning_rate)
        steps_tr, steps_va = len(loaders[0]), len(loaders[1])
        tr = runner.run_epoch(loaders[0], 0, opt)
        va = runner.run_epoch(loaders[1], 1, None)
        torch.save({"w": {k: v.detach().clone() for k, v in model.state_dict().items()}, "train": (tr.loss, tr.topk),
                    "valid": (va.loss, va.topk), "steps": (steps_tr, steps_va),
                    "valid_total": int(runner.metrics[0].total)}, os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("global_loss", [False, True], ids=["per_rank_batch", "global_batch_loss"])
def test_epoch_with_ragged_shards_runs_in_lockstep(global_loss, tmp_path):
    from drin_amd.data import write_synthetic_dataset
    root = str(tmp_path / "data")
    # 13 train mentions on 2 ranks x batch 4: shards of 7 (one wrapped-around mention) -> 2 steps of sizes (4, 3) on BOTH
    # ranks; unpadded they would be 7 / 6 mentions = last batches of 3 / 2.  5 valid mentions: 3 + 2, never padded
    write_synthetic_dataset(CFG.with_(shuffle_train_data=True), root, sizes=(13, 5, 3), seed=11)
    out = str(tmp_path / "out")
    os.makedirs(out)
    mp.spawn(_epoch_worker, args=(2, _free_port(), root, out, global_loss), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(out, f"rank{r}.pt")) for r in range(2))
    assert r0["steps"][0] == r1["steps"][0] == 2                       # same number of training steps
    for k in r0["w"]:
        assert torch.equal(r0["w"][k], r1["w"][k]), f"replicas diverged: {k}"   # every step's all-reduce paired up
    assert r0["train"] == r1["train"] and r0["valid"][1] == r1["valid"][1]
    assert r0["valid_total"] == r1["valid_total"] == 5                 # evaluation counts every mention exactly once


def test_grad_bucket_reduces_a_flat_bucket_in_place_and_packs_anything_else():
    """Gradients that are views of one flat buffer (what drin_amd.Model's backward hands to autograd) are recognised and
    all-reduced in place; scattered gradients take the staging copy.  (world = 1 here: only the recognition is checked;
    the collective itself runs in the two-rank tests above and in bench.py --stub.)"""
    params = [torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7)), torch.nn.Parameter(torch.zeros(2))]
    flat = torch.arange(64 + 64 + 7, dtype=torch.float32)
    params[0].grad, params[1].grad = flat[0:15].view(5, 3), flat[64:71]      # 256-byte slots, params[2] has no gradient
    b = GradBucket(params)
    live = [p for p in b.params if p.grad is not None]
    view = b._aliased_bucket(live)
    assert view is not None and view.data_ptr() == flat.data_ptr() and view.numel() == 71
    view.mul_(2.0)
    assert params[1].grad[0].item() == 128.0                                 # the same memory
    params[1].grad = torch.ones(7)
    assert b._aliased_bucket([p for p in b.params if p.grad is not None]) is None
    assert b.nbytes() == 4 * (15 + 7)


# ---- the gradient all-reduce in two pieces, the first one started inside backward --------------------
def _flat_bucket_step(model, bucket, shard, overlap, reduce=True):
    """One step's gradients laid out as the HIP backward leaves them - every .grad a view of the model's flat bucket - with
    the staged hook called where `_DrinScore.backward` calls it (the layers' gradients final, the vertex encoders' not yet
    all-reduced), then the step's `allreduce_mean()`."""
    from drin_amd.metrics import TripletLoss
    from drin_amd.model import _param_list
    model.zero_grad(set_to_none=True)
    TripletLoss(CFG.triplet_margin)(shard[-1], model(shard[:-1])).backward()
    params = _param_list(model)
    offsets, live, total = model.bucket_layout()
    flat = torch.zeros(total)
    for o, p in zip(offsets, params):
        if p.grad is not None:
            view = flat[o:o + p.numel()].view(p.shape)
            view.copy_(p.grad)
            p.grad = view
    model._grad_flat = flat
    if overlap:
        model._layers_ready_hook(flat[:live], offsets[8], None)
    if reduce:
        bucket.allreduce_mean()
    return flat[:live]


def _overlap_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    try:
        from drin_amd.train import OverlappedStep
        out = {}
        for overlap in (False, True, "pipe"):
            model = OracleModel(CFG)
            model.load_state_dict(synth.make_state_dict(CFG, 8))
            bucket = GradBucket(list(model.parameters()), overlap=overlap is True, model=model)
            opt = torch.optim.Adam(model.parameters(), lr=1e-2)
            pipe = OverlappedStep(model, bucket, opt) if overlap == "pipe" else None   # host tensors: runs inline
            for step in range(3):
                batch = synth.make_batch(CFG, 8, 90 + step)
                shard = [t[rank * 4:(rank + 1) * 4] for t in batch]
                _flat_bucket_step(model, bucket, shard, overlap is True, reduce=pipe is None)
                if pipe is not None:
                    pipe.run()
                else:
                    opt.step()
                assert bucket.in_place
            assert bucket.overlapped == (3 if overlap is True else 0) and bucket.collectives == (6 if overlap is True else 3)
            bucket.close()
            assert model._layers_ready_hook is None
            out[overlap] = {k: v.detach().clone() for k, v in model.state_dict().items()}
        torch.save(out, os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_piece_overlapped_allreduce_is_bit_identical_to_the_one_piece_path():
    """Three optimiser steps on two ranks with the bucket all-reduced in one piece after backward, and with the GCN layers'
    piece started from the staged-backward hook + the vertex encoders' piece after it: identical parameters, bit for bit,
    on both ranks."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_overlap_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        r0, r1 = (torch.load(os.path.join(d, f"rank{r}.pt")) for r in range(2))
    for k in r0[False]:
        assert torch.equal(r0[False][k], r0[True][k]), f"two-piece all-reduce changed {k}"
        assert torch.equal(r0[False][k], r0["pipe"][k]), f"OverlappedStep changed {k}"
        assert torch.equal(r0[True][k], r1[True][k]), f"replicas diverged: {k}"


def test_overlapped_bucket_refuses_gradients_that_left_the_bucket():
    """A piece in flight on memory that is no longer the gradient (accumulation replaced the views) must raise, not
    silently reduce the wrong bytes.  World of one rank, forced collectives."""
    port = _free_port()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        model = OracleModel(CFG)
        model.load_state_dict(synth.make_state_dict(CFG, 8))
        bucket = GradBucket(list(model.parameters()), force=True, overlap=True, model=model)
        batch = synth.make_batch(CFG, 4, 3)
        before = _flat_bucket_step(model, bucket, batch, True).clone()
        assert bucket.overlapped == 1 and bucket.collectives == 2
        assert torch.equal(before, model._grad_flat[:before.numel()])        # a world of one: the mean is the value itself
        # now the hook fires but the gradients are replaced before the step's all-reduce
        from drin_amd.model import _param_list
        offsets, live, _ = model.bucket_layout()
        model._layers_ready_hook(model._grad_flat[:live], offsets[8], None)
        for p in _param_list(model):
            if p.grad is not None:
                p.grad = p.grad.clone()
        with pytest.raises(RuntimeError, match="overlap=False"):
            bucket.allreduce_mean()
        plain = GradBucket(list(model.parameters()))                         # no force: a world of one does nothing
        plain.allreduce_mean()
        assert plain.collectives == 0
    finally:
        dist.destroy_process_group()
