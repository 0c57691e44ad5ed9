"""-m gpu, round 3: a backward pass without atomics (bit-reproducible gradients, shared weight-gradient destinations), the
staged backward (`drin_backward_staged`) and the two-piece overlapped gradient all-reduce - through RCCL itself in a
process group of one rank, and through gloo with two ranks sharing device 0 - two scoring calls in one graph, LibraryAdam's
per-parameter step counts."""
import ctypes as C
import os
import socket
import subprocess
import sys
import tempfile

import pytest
import torch

from drin_amd import _lib, synth
from drin_amd.config import DrinConfig, wikimel_config
from drin_amd.metrics import TripletLoss
from drin_amd.model import Model, _param_list
from drin_amd.train import GradBucket, LibraryAdam, make_adam
from oracle import drin_oracle as O
from oracle.cases import TINY

pytestmark = pytest.mark.gpu
DEV = "cuda"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(cfg, B, seed=31, precision="bf16x3", wseed=8, **kw):
    sd = synth.make_state_dict(cfg, wseed)
    model = Model(cfg, precision=precision, **kw).to(DEV)
    model.load_state_dict(sd)
    batch = [t.to(DEV) for t in synth.make_batch(cfg, B, seed)]
    return model, batch, sd


def _loss(model, batch, cfg):
    return TripletLoss(cfg.triplet_margin)(batch[14], model(batch[:14]))


def _grads(model, batch, cfg):
    model.zero_grad(set_to_none=True)
    _loss(model, batch, cfg).backward()
    return {k: (None if p.grad is None else p.grad.clone()) for k, p in model.named_parameters()}


# ---- no atomics anywhere in the backward pass: the same bits every run (train.py:134 seed_everything) --------
CASES = {
    "wikidiverse_b64": (lambda: DrinConfig(), 64, "bf16x3"),
    "wikimel_b8": (lambda: wikimel_config(max_entity_attr_token_len=16), 8, "bf16x3"),
    "wikidiverse_b64_f32": (lambda: DrinConfig(), 64, "f32"),
    "wikimel_b24_f32": (lambda: wikimel_config(max_entity_attr_token_len=8), 24, "f32"),       # 2 424 pair rows: the sliced exact-fp32 dW
    "vector_tiny": (lambda: DrinConfig(gcn_edge_feature="vector", **TINY), 40, "bf16x3"),
    "wikimel_b64": (lambda: wikimel_config(max_entity_attr_token_len=8), 64, "bf16x3"),        # the grouped split-bf16 dW launch
}


@pytest.mark.parametrize("case", list(CASES))
def test_backward_is_bit_reproducible(case):
    maker, B, prec = CASES[case]
    cfg = maker()
    model, batch, _ = _setup(cfg, B, precision=prec)
    first = _grads(model, batch, cfg)
    for _ in range(2):
        again = _grads(model, batch, cfg)
        for k, g in first.items():
            assert (g is None) == (again[k] is None), k
            if g is not None:
                assert torch.equal(g, again[k]), f"{k}: {int((g != again[k]).sum())} of {g.numel()} gradient entries differ between two backward passes"
    # and through the per-tensor gradient path (another memory layout, the same arithmetic)
    plain, _, _ = _setup(cfg, B, precision=prec, grad_bucket=False)
    other = _grads(plain, batch, cfg)
    for k, g in first.items():
        if g is not None:
            assert torch.equal(g, other[k]), k


@pytest.mark.parametrize("maker,B,prec", [(lambda: DrinConfig(), 64, "bf16x3"), (lambda: wikimel_config(max_entity_attr_token_len=8), 8, "bf16x3"),
                                          (lambda: DrinConfig(**TINY), 16, "f32")], ids=["wikidiverse_b64", "wikimel_b8", "tiny_f32"])
def test_two_training_loops_end_bit_identical(maker, B, prec):
    """Five steps of forward / TripletLoss / backward / LibraryAdam, run twice from the same state: bit-identical parameters
    (Adam's division by sqrt(v) turns a last-bit gradient difference into a +-lr step - there is none to amplify)."""
    cfg = maker()
    runs = []
    for _ in range(2):
        model, batch, _ = _setup(cfg, B, precision=prec)
        opt = make_adam(model, 1e-3)
        assert isinstance(opt, LibraryAdam)
        for _step in range(5):
            opt.zero_grad(set_to_none=True)
            _loss(model, batch, cfg).backward()
            opt.step()
        assert opt.one_launch_steps == 5
        runs.append({k: v.clone() for k, v in model.state_dict().items()})
    for k in runs[0]:
        assert torch.equal(runs[0][k], runs[1][k]), k


def test_weight_gradient_products_that_share_a_destination(monkeypatch):
    """dW_h takes the mention rows AND the entity rows.  With one candidate per mention and 1 024 mentions
    both products are pair-sized (2 048 / 4 096 rows), ride in the same grouped split-bf16 launch and reduce onto the same
    matrix: they must be summed by ONE slice-sum entry (no lost update), reproducibly, and match the oracle's autograd."""
    cfg = DrinConfig(num_candidates_data=1)
    B = 1024
    model, batch, sd = _setup(cfg, B, seed=5)
    got = _grads(model, batch, cfg)
    for _ in range(3):
        again = _grads(model, batch, cfg)
        for k in ("gcn_layers.0.w_h.weight", "gcn_layers.1.w_h.weight", "gcn_layers.0.w_v.weight"):
            assert torch.equal(got[k], again[k]), k
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    host = [t.cpu() for t in batch]
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    O.triplet_loss(host[14], O.forward(p, host[:14]), cfg.triplet_margin).backward()
    for k, v in p.items():
        if v.grad is None:
            assert got[k] is None, k
            continue
        rel = (got[k].cpu() - v.grad).norm().item() / (v.grad.norm().item() + 1e-12)
        assert rel <= 2e-4, (k, rel)


def test_linear_bwd_without_scratch_is_deterministic_and_right():
    """drin_linear_bwd with no scratch: one workgroup per output tile / per 256 columns walks the whole reduction - slower,
    still no atomics; with scratch the sliced path: both match torch and repeat bit for bit."""
    lib = _lib.load()
    g = torch.Generator(device=DEV).manual_seed(2)
    rows, n_out, k = 5000, 192, 320
    x, dy = torch.randn(rows, k, device=DEV, generator=g), torch.randn(rows, n_out, device=DEV, generator=g)
    want_dw, want_db = dy.double().t() @ x.double(), dy.double().sum(0)
    st = torch.cuda.current_stream().cuda_stream
    for scratch_floats in (0, 1 << 22):
        scratch = torch.empty(max(scratch_floats, 1), device=DEV)
        outs = []
        for _ in range(2):
            dw, db = torch.zeros(n_out, k, device=DEV), torch.zeros(n_out, device=DEV)
            _lib.check(lib.drin_linear_bwd(x.data_ptr(), None, dy.data_ptr(), None, dw.data_ptr(), db.data_ptr(), rows, n_out, k,
                                           _lib.PREC_F32, scratch.data_ptr() if scratch_floats else None, scratch_floats, st))
            outs.append((dw, db))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        assert (outs[0][0].double() - want_dw).abs().max().item() <= 2e-3 and (outs[0][1].double() - want_db).abs().max().item() <= 2e-3
        # accumulation semantics (+=) survive both paths
        dw, db = outs[0]
        before = dw.clone()
        _lib.check(lib.drin_linear_bwd(x.data_ptr(), None, dy.data_ptr(), None, dw.data_ptr(), None, rows, n_out, k, _lib.PREC_F32,
                                       scratch.data_ptr() if scratch_floats else None, scratch_floats, st))
        assert (dw - 2 * before).abs().max().item() <= 1e-3


# ---- two scoring calls in one graph must not share the flat gradient bucket ------------------------------
def test_two_forwards_one_backward_match_the_per_tensor_path():
    cfg = DrinConfig(**TINY)
    model, b1, _ = _setup(cfg, 6, seed=3)
    plain, _, _ = _setup(cfg, 6, seed=3, grad_bucket=False)
    b2 = [t.to(DEV) for t in synth.make_batch(cfg, 5, 4)]
    for m in (model, plain):
        (_loss(m, b1, cfg) + 2.0 * _loss(m, b2, cfg)).backward()
    for (k, p), (_, q) in zip(model.named_parameters(), plain.named_parameters()):
        assert (p.grad is None) == (q.grad is None), k
        if p.grad is not None:
            assert torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-7), k
    # the next ordinary step uses the bucket again
    model.zero_grad(set_to_none=True)
    _loss(model, b1, cfg).backward()
    assert model.grad_bucket() is not None and not model._bucket_in_flight


def test_training_batch_above_the_call_limit_accumulates_correctly(monkeypatch):
    cfg = DrinConfig(**TINY)
    model, batch, _ = _setup(cfg, 10, seed=6)
    plain, _, _ = _setup(cfg, 10, seed=6, grad_bucket=False)
    monkeypatch.setattr(Model, "MAX_CALL_MENTIONS", 4)            # 10 mentions -> three scoring calls in one graph
    for m in (model, plain):
        _loss(m, batch, cfg).backward()
    whole, _, _ = _setup(cfg, 10, seed=6, grad_bucket=False)
    monkeypatch.setattr(Model, "MAX_CALL_MENTIONS", 32768)
    _loss(whole, batch, cfg).backward()
    for (k, p), (_, q), (_, w) in zip(model.named_parameters(), plain.named_parameters(), whole.named_parameters()):
        if q.grad is not None:
            assert torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-7), k
            assert torch.allclose(p.grad, w.grad, rtol=2e-4, atol=1e-6), k


# ---- LibraryAdam counts its steps per parameter, like torch ------------------------------------------------------
def test_library_adam_parameter_unfrozen_later_follows_torch_bitwise():
    cfg = DrinConfig(**TINY)
    a, batch, _ = _setup(cfg, 8, precision="f32")
    b, _, _ = _setup(cfg, 8, precision="f32", grad_bucket=False)
    late = "vertex_encoder.mention_image_linear.weight"
    oa, ob = make_adam(a, 1e-2), torch.optim.Adam(b.parameters(), lr=1e-2)
    for step in range(5):
        frozen = step < 2
        dict(a.named_parameters())[late].requires_grad_(not frozen)
        oa.zero_grad(set_to_none=True)
        ob.zero_grad(set_to_none=True)
        _loss(a, batch, cfg).backward()
        for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):   # the SAME gradients into both optimisers
            q.grad = None if p.grad is None else p.grad.clone()
        oa.step()
        ob.step()
        for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
            assert torch.equal(p, q), f"step {step}: {k}"
    idx = [id(p) for p in _param_list(a)].index(id(dict(a.named_parameters())[late]))
    assert oa.t == 5 and oa.steps[idx] == 3 and oa.one_launch_steps == 0


# ---- drin_backward_staged: the event in the right place, the same gradients up to the summation split -----------------------
@pytest.mark.parametrize("maker,B,prec", [(lambda: wikimel_config(max_entity_attr_token_len=8), 64, "bf16x3"), (lambda: DrinConfig(**TINY), 12, "f32"),
                                          (lambda: DrinConfig(), 1200, "bf16x3")], ids=["wikimel_b64", "tiny_f32", "wikidiverse_b1200"])
def test_staged_backward_matches_plain_backward(maker, B, prec):
    cfg = maker()
    model, batch, _ = _setup(cfg, B, precision=prec)
    plain = _grads(model, batch, cfg)
    seen = []

    def hook(live_flat, split, ready):
        # called right after drin_backward_staged was enqueued: wait for the event on a side stream and snapshot the GCN
        # layers' piece there - it must already hold its final values, while the step's stream is still busy with the encoders
        side = torch.cuda.Stream()
        side.wait_event(ready)
        with torch.cuda.stream(side):
            seen.append((split, live_flat[split:].clone(), side))

    model._layers_ready_hook = hook
    staged = _grads(model, batch, cfg)
    again = _grads(model, batch, cfg)
    model._layers_ready_hook = None
    assert len(seen) == 2
    for k, g in plain.items():
        assert (g is None) == (staged[k] is None), k
        if g is not None:
            assert torch.equal(staged[k], again[k]), f"{k}: the staged pass does not repeat its own bits"
            # the two parts deal the chip's workgroups separately: another split of the same sums for the pair-sized dW
            assert (g - staged[k]).abs().max().item() <= 2e-6 * g.abs().max().item() + 1e-12, k
    split, early, side = seen[1]
    side.synchronize()
    offsets, live, _ = model.bucket_layout()
    assert split == offsets[8] and torch.equal(early, model._grad_flat[split:live]), "the layers' gradients changed after the event"


def test_staged_forward_waits_for_the_update_on_the_side_stream():
    """`drin_forward_staged`: the parameters are rewritten on a side stream by a kernel queue the step's stream knows nothing
    about except through the event; the scores must be those of the NEW parameters (and equal drin_forward's bit for bit)."""
    cfg = wikimel_config(max_entity_attr_token_len=8)
    model, batch, sd = _setup(cfg, 16)
    model.train()
    want_old = model(batch[:14]).detach().clone()
    new_sd = synth.make_state_dict(cfg, 9)
    other, _, _ = _setup(cfg, 16, wseed=9)
    other.train()
    want_new = other(batch[:14]).detach().clone()
    assert not torch.equal(want_old, want_new)
    side = torch.cuda.Stream()
    spin = torch.empty(64 << 20, device=DEV)
    with torch.cuda.stream(side):
        for _ in range(20):                                   # ~ms of work in front of the parameter write
            spin.add_(1.0)
        with torch.no_grad():
            for k, p in model.named_parameters():
                p.copy_(new_sd[k].to(DEV))
        ready = torch.cuda.Event()
        ready.record(side)
    model._params_ready = ready
    got = model(batch[:14]).detach()
    assert model._params_ready is None                           # consumed by the staged forward
    torch.cuda.synchronize()
    assert torch.equal(got, want_new)


# ---- RCCL itself, in a process group of ONE rank -------------------------------------------------------
_RCCL_CHILD = r'''
import json, os, sys
sys.path.insert(0, {repo!r})
import torch, torch.distributed as dist
from drin_amd import synth
from drin_amd.config import wikimel_config
from drin_amd.model import Model
from drin_amd.train import MELRunner, GradBucket, _GatherScores, make_adam
from drin_amd.metrics import TripletLoss, TopkAccuracy
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev, world_size=1, rank=0)
cfg = wikimel_config(max_entity_attr_token_len=8, batch_size=16)
sd = synth.make_state_dict(cfg, 8)
out = {{"backend": dist.get_backend()}}
finals = {{}}
from drin_amd.train import OverlappedStep
for mode in ("off", "none", "forward", "backward", "both"):
    model = Model(cfg).to(dev)
    model.load_state_dict(sd)
    runner = MELRunner(cfg, model, dev, force_collectives=mode != "off", overlap_allreduce="none" if mode == "off" else mode)
    opt = make_adam(model, 1e-3)
    pipe = OverlappedStep(model, runner.bucket, opt) if mode in ("forward", "both") else None
    for step in range(3):
        batch = [t.to(dev) for t in synth.make_batch(cfg, 16, 40 + step)]
        opt.zero_grad(set_to_none=True)
        loss = runner.forward_step(batch, 0)
        loss.backward()
        if pipe is not None:
            pipe.run()
        else:
            before = model.grad_bucket().clone()
            runner.bucket.allreduce_mean()
            assert torch.equal(before, model.grad_bucket()), "ReduceOp.AVG over one rank changed the gradients"
            opt.step()
    if pipe is not None:
        assert model._params_ready is not None
        pipe.finish()
    torch.cuda.synchronize()
    out[mode] = {{"collectives": runner.bucket.collectives, "early": runner.bucket.overlapped, "in_place": runner.bucket.in_place}}
    runner.device_loss.sync(force=True)
    out[mode]["total"] = runner.device_loss.total
    finals[mode] = {{k: v.clone() for k, v in model.state_dict().items()}}
    runner.close()
for k in finals["off"]:
    assert torch.equal(finals["off"][k], finals["none"][k]) and torch.equal(finals["off"][k], finals["forward"][k]), k
    assert torch.equal(finals["backward"][k], finals["both"][k]), k
    assert (finals["off"][k] - finals["both"][k]).abs().max().item() <= 5e-3, k   # another split of the dW sums, through 3 Adam steps of lr 1e-3
# a whole epoch through the runner's own loop in its default mode for collectives ("forward")
model = Model(cfg).to(dev)
model.load_state_dict(sd)
runner = MELRunner(cfg, model, dev, force_collectives=True)
assert runner.overlap_mode == "forward"
batches = [synth.make_batch(cfg, 16, 40 + i) for i in range(3)]
log = runner.run_epoch(batches, 0, make_adam(model, 1e-3))
assert model._params_ready is None and runner._pipe.steps == 3
for k, v in model.state_dict().items():
    assert torch.equal(v, finals["off"][k]), k
out["epoch_loss"] = log.loss
runner.close()
# the score gather and the torch metric's sync through RCCL
x = torch.randn(5, 7, device=dev, requires_grad=True)
g = _GatherScores.apply(x)
g.sum().backward()
assert torch.equal(g, x) and torch.equal(x.grad, torch.ones_like(x))
m = TopkAccuracy(1, dev)
m.update(torch.eye(4, 5, device=dev), torch.eye(4, 4, device=dev).to(torch.uint8))
m.sync(force=True)
out["topk"] = [int(m.correct), int(m.total)]
# the gathered global-batch loss through the runner (all_gather of answers and scores)
model = Model(cfg).to(dev)
model.load_state_dict(sd)
runner = MELRunner(cfg, model, dev, global_batch_loss=True, force_collectives=True, overlap_allreduce=False)
batch = [t.to(dev) for t in synth.make_batch(cfg, 16, 40)]
model.train()
l_g = float(runner.forward_step(batch, 0))
l_p = float(TripletLoss(cfg.triplet_margin)(batch[14], model(batch[:14])))
out["global_loss"] = [l_g, l_p]
dist.barrier()
dist.destroy_process_group()
print("RCCL_CHILD " + json.dumps(out), flush=True)
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _child_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


def test_rccl_process_group_of_one_rank_drives_the_real_collectives():
    """`dist.init_process_group("nccl", world_size=1)` in a fresh process: the flat gradient bucket of the HIP Model is
    all-reduced IN PLACE with ReduceOp.AVG through librccl - serially, on the side stream under the next forward's head
    (OverlappedStep), in two pieces with the first started inside the staged backward, and both - leaving the gradients
    bit-unchanged and the parameters after three LibraryAdam steps identical to a run without collectives (the staged
    backward: to its own summation split); the score gather, both metric syncs and the gathered global-batch loss run
    through RCCL too."""
    import json
    r = subprocess.run([sys.executable, "-c", _RCCL_CHILD.format(repo=REPO)], capture_output=True, text=True, env=_child_env(), timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RCCL_CHILD ")][-1]
    out = json.loads(line[len("RCCL_CHILD "):])
    assert out["backend"] == "nccl"
    assert out["off"] == {"collectives": 0, "early": 0, "in_place": False, "total": 48}
    assert out["none"] == {"collectives": 3, "early": 0, "in_place": True, "total": 48}
    assert out["forward"] == {"collectives": 3, "early": 0, "in_place": True, "total": 48}
    assert out["backward"] == {"collectives": 6, "early": 3, "in_place": True, "total": 48}
    assert out["both"] == {"collectives": 6, "early": 3, "in_place": True, "total": 48}
    assert 0 < out["epoch_loss"] < 1
    assert out["topk"] == [4, 4]
    assert abs(out["global_loss"][0] - out["global_loss"][1]) <= 1e-6 * max(1.0, abs(out["global_loss"][1]))


def test_bench_force_collective_reports_a_nonzero_rccl_allreduce():
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    full_path = os.path.join(tempfile.mkdtemp(prefix="drin_bench_"), "full.json")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--mode", "train", "--gpus", "1", "--force-collective", "--batch", "64",
                        "--steps", "5", "--warmup", "5", "--legs-file", full_path], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    head = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert len([ln for ln in r.stdout.splitlines() if ln.strip()]) == 1, r.stdout[:500]   # RCCL's banner went to stderr
    assert head["collective"]["backend"] == "nccl" and head["collective"]["world"] == 1 and head["allreduce_ms"] > 0
    line = json.load(open(full_path))                                                       # the full record: in the legs file
    c = line["collective"]
    assert c["backend"] == "nccl" and c["world"] == 1 and c["forced_world_of_one"] and c["overlap"] == "forward" and c["in_place"]
    assert c["steps_overlapped_under_next_forward"] >= 10 and line["allreduce_ms"] > 0 and line["allreduce_exposed_ms"] >= 0
    assert c["serial_ms_per_step"] > 0 and c["no_collective_ms_per_step"] > 0
    assert line["allreduce_bytes"] == 26775552 and line["step_floor_ms"] > 0 and line["step_floor_ms"] < line["ms_per_step"]


# ---- two ranks sharing device 0 over gloo: the overlapped all-reduce through the HIP Model -----------------------------------
def _overlap_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    from drin_amd.train import OverlappedStep
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        cfg = wikimel_config(max_entity_attr_token_len=8, batch_size=16)
        sd = synth.make_state_dict(cfg, 8)
        out = {}
        # serial: collective after backward, Adam behind it.  forward: both on a side stream under the next forward's head.
        # staged: the staged backward with the bucket still reduced in ONE piece.  backward / both: the layers' piece early.
        for mode in ("serial", "forward", "staged", "backward", "both"):
            model = Model(cfg).to(dev)
            model.load_state_dict(sd)
            early = mode in ("backward", "both")
            bucket = GradBucket(list(model.parameters()), overlap=early, model=model if early else None)
            if mode == "staged":
                model._layers_ready_hook = lambda flat, split, ev: None
            opt = make_adam(model, 1e-2)
            pipe = OverlappedStep(model, bucket, opt) if mode in ("forward", "both") else None
            for step in range(3):
                full = synth.make_batch(cfg, 32, 60 + step)
                shard = [t[rank * 16:(rank + 1) * 16].to(dev) for t in full]
                opt.zero_grad(set_to_none=True)
                TripletLoss(cfg.triplet_margin)(shard[14], model(shard[:14])).backward()
                if pipe is not None:
                    pipe.run()
                else:
                    bucket.allreduce_mean()
                    opt.step()
            if pipe is not None:
                pipe.finish()
            torch.cuda.synchronize()
            assert bucket.in_place and bucket.overlapped == (3 if early else 0) and bucket.collectives == (6 if early else 3)
            bucket.close()
            out[mode] = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        torch.save(out, os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_overlapped_allreduce_through_the_hip_model_is_bit_identical_on_two_ranks():
    """two ranks on device 0 over gloo (RCCL refuses two ranks on one device), three optimiser steps.  The
    collective + Adam on a side stream under the next forward's parameter-free head gives the serial loop's parameters bit
    for bit; so does the two-piece all-reduce started inside the staged backward against the one-piece all-reduce behind the
    same staged backward; replicas stay identical in every mode."""
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_overlap_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        r0, r1 = (torch.load(os.path.join(d, f"rank{r}.pt")) for r in range(2))
    for k in r0["serial"]:
        assert torch.equal(r0["serial"][k], r0["forward"][k]), f"the side-stream update changed {k}"
        assert torch.equal(r0["staged"][k], r0["backward"][k]), f"the two-piece all-reduce changed {k}"
        assert torch.equal(r0["staged"][k], r0["both"][k]), f"both overlaps together changed {k}"
        for mode in r0:
            assert torch.equal(r0[mode][k], r1[mode][k]), f"replicas diverged ({mode}): {k}"
    w0 = synth.make_state_dict(wikimel_config(max_entity_attr_token_len=8), 8)["gcn_layers.0.w_h.weight"]
    assert not torch.equal(r0["forward"]["gcn_layers.0.w_h.weight"], w0)


def test_overlapped_step_with_a_split_training_batch(monkeypatch):
    """`OverlappedStep` + a training batch above `MAX_CALL_MENTIONS` (several scoring calls per forward): the first call takes
    the pending update's event, the others find none; two steps end with the serial loop's parameters bit for bit."""
    from drin_amd.train import OverlappedStep
    cfg = DrinConfig(**TINY)
    monkeypatch.setattr(Model, "MAX_CALL_MENTIONS", 5)
    finals = []
    for overlapped in (False, True):
        model, batch, _ = _setup(cfg, 12, seed=11)
        opt = make_adam(model, 1e-2)
        bucket = GradBucket(list(model.parameters()))
        pipe = OverlappedStep(model, bucket, opt) if overlapped else None
        for _step in range(3):
            opt.zero_grad(set_to_none=True)
            _loss(model, batch, cfg).backward()
            if pipe is not None:
                pipe.run()
                assert model._params_ready is not None
            else:
                opt.step()
        if pipe is not None:
            pipe.finish()
            assert model._params_ready is None
        torch.cuda.synchronize()
        finals.append({k: v.clone() for k, v in model.state_dict().items()})
    for k in finals[0]:
        assert torch.equal(finals[0][k], finals[1][k]), k


def test_train_cli_test_only(tmp_path, monkeypatch):
    """`args.test_only` (train.py:137-140): `python -m drin_amd.train --test-only --load-state PATH` makes one pass over the test
    split with the loaded weights and writes the per-sample dump; no training epoch runs."""
    from drin_amd import train as T
    from drin_amd.data import write_synthetic_dataset
    cfg = DrinConfig(batch_size=4, num_epoch=2, test_epoch_interval=1, **TINY)
    write_synthetic_dataset(cfg, str(tmp_path), sizes=(8, 4, 6), seed=3)
    state = str(tmp_path / "weights.pt")
    sd = synth.make_state_dict(cfg, 8)
    torch.save(sd, state)
    dump = str(tmp_path / "test-result.txt")
    monkeypatch.setattr(T, "DrinConfig", lambda **kw: cfg)
    lines = []
    monkeypatch.setattr("builtins.print", lambda *a, **k: lines.append(" ".join(map(str, a))))
    T.main(["--data", str(tmp_path), "--test-only", "--load-state", state, "--output-test-result", dump])
    assert sum(ln.startswith("test: loss") for ln in lines) == 1 and not any(ln.startswith("epoch ") for ln in lines)
    assert not any("Training completed" in ln for ln in lines)
    rows = open(dump).read().splitlines()
    assert rows[0] == "==========  Test ==========" and sum(r.split(":\t")[0].isdigit() for r in rows[1:]) == 6
    # the scores in the dump are those of the loaded weights
    model = Model(cfg).to(DEV).eval()
    model.load_state_dict(sd)
    from drin_amd.data import create_datasets
    batch = next(iter(create_datasets(cfg, str(tmp_path))[2]))
    with torch.no_grad():
        want = model([t.to(DEV) for t in batch[:-1]])[0].tolist()
    got = eval(rows[1].split(":\t", 1)[1])
    assert max(abs(a - b) for a, b in zip(got, want)) <= 1e-6


def test_four_phase_gemm_with_a_single_k_block():
    """D = R = 32: every pair-sized contraction of the fused path is ONE 32-wide K-block (the four-phase kernels' prologue,
    one block that re-fetches itself, epilogue) with 32 output columns of a 256-column tile; 80 000 pair rows = 313 tiles,
    ragged last row tile.  Slices against the oracle, mention independence."""
    cfg = DrinConfig(num_candidates_data=3, bert_embed_dim=32, gcn_embed_dim=32, resnet_embed_dim=32,
                     max_mention_sentence_len=12, resnet_num_region=5)
    sd = synth.make_state_dict(cfg, 8)
    model = Model(cfg).to(DEV).eval()
    model.load_state_dict(sd)
    B = 20_011
    batch = synth.make_device_batch(cfg, B, 9, DEV)[:14]
    with torch.no_grad():
        out = model(batch)
        assert out.shape == (B, 4) and torch.isfinite(out).all()
        for rows in (slice(0, 6), slice(B - 5, B), slice(10_000, 10_006)):
            ref = O.forward(sd, [t[rows].cpu() for t in batch])
            assert (out[rows].cpu() - ref).abs().max().item() <= 1e-5
            assert (out[rows] - model([t[rows] for t in batch])).abs().max().item() <= 5e-6


@pytest.mark.parametrize("B", [1100, 2100])
def test_large_call_candidate_chunking(B):
    """From 1 024 mentions up a workgroup of the stream / pair kernels takes 48 candidates of its mention, from 2 048 up to 128
    (here: all 37, ten per wave, ragged) instead of 16: slices against the oracle and against the same mentions scored in a
    small call (16-candidate workgroups; equal up to fp32 re-association of the per-mention sums)."""
    cfg = DrinConfig(dataset_name="wikimel", num_candidates_data=36, max_entity_attr_token_len=9, **TINY)
    sd = synth.make_state_dict(cfg, 8)
    model = Model(cfg).to(DEV).eval()
    model.load_state_dict(sd)
    batch = synth.make_device_batch(cfg, B, 13, DEV)[:14]
    with torch.no_grad():
        out = model(batch)
        assert out.shape == (B, 37) and torch.isfinite(out).all()
        for rows in (slice(0, 5), slice(B - 4, B), slice(B // 2, B // 2 + 5)):
            ref = O.forward(sd, [t[rows].cpu() for t in batch])
            assert (out[rows].cpu() - ref).abs().max().item() <= 1e-5
            assert (out[rows] - model([t[rows] for t in batch])).abs().max().item() <= 5e-6
        assert torch.equal(out, model(batch))
