"""-m gpu, round 2: flat gradient / parameter buckets and the one-launch Adam, data-parallel semantics through the HIP
`Model`, call splitting above 32 768 mentions, BASELINE config 5 at FULL width (N = 1001, D = 768, R = 2048, a >= 1 M-row
entity table, fused path and per-entity cache) against the oracle, cache / fold invalidation."""
import ctypes as C

import pytest
import torch

from drin_amd import _lib, synth
from drin_amd.config import DrinConfig, wikimel_config
from drin_amd.model import EntityTable, IndexedBatch, Model, _param_list
from drin_amd.train import GradBucket, LibraryAdam, make_adam
from oracle import drin_oracle as O
from oracle.cases import TINY

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _to_dev(batch):
    return [t.to(DEV) for t in batch]


# ---- one-launch Adam (train.py:55-56) ---------------------------------------------------------------------------------
def _adam_call(p, g, m, v, t, lr):
    lib = _lib.load()
    b1, b2, eps = 0.9, 0.999, 1e-8
    bc1, bc2 = 1 - b1 ** float(t), 1 - b2 ** float(t)
    scal = (1 - b1, b2, 1 - b2, bc2 ** 0.5, eps, (lr / bc1) * -1)
    _lib.check(lib.drin_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), *(C.c_float(x) for x in scal),
                                  torch.cuda.current_stream().cuda_stream))


def test_library_adam_matches_torch_adam_bitwise():
    """drin_adam_step == torch.optim.Adam (default multi-tensor implementation) BIT FOR BIT over several steps, on values
    spanning many magnitudes: the training loop's trajectory is then the reference loop's (Adam amplifies rounding)."""
    g = torch.Generator(device=DEV).manual_seed(5)
    n = 3 * 768 * 768 + 13                                    # ragged tail included
    p0 = torch.randn(n, device=DEV, generator=g) * torch.logspace(-6, 2, n, device=DEV)
    grads = [torch.randn(n, device=DEV, generator=g) * torch.logspace(-8, 1, n, device=DEV).flip(0) for _ in range(6)]
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-3)
    p, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for t, gr in enumerate(grads, 1):
        ref.grad = gr.clone()
        opt.step()
        _adam_call(p, gr, m, v, t, 1e-3)
        st = opt.state[ref]
        for name, mine, theirs in (("exp_avg", m, st["exp_avg"]), ("exp_avg_sq", v, st["exp_avg_sq"]), ("param", p, ref.data)):
            assert torch.equal(mine, theirs), f"step {t}: {name} differs in {int((mine != theirs).sum())} of {n} elements"
    assert _lib.load().drin_adam_step(None, None, None, None, 4, 0.1, 0.999, 0.001, 1.0, 1e-8, -1e-3, None) == _lib.E_NULL


def _train_setup(cfg, B, seed=31, precision="bf16x3", **kw):
    sd = synth.make_state_dict(cfg, 8)
    model = Model(cfg, precision=precision, **kw).to(DEV)
    model.load_state_dict(sd)
    batch = _to_dev(synth.make_batch(cfg, B, seed))
    return model, batch, sd


def _step_loss(model, batch, cfg):
    from drin_amd.metrics import TripletLoss
    return TripletLoss(cfg.triplet_margin)(batch[14], model(batch[:14]))


@pytest.mark.parametrize("maker", [lambda: DrinConfig(), lambda: DrinConfig(gcn_edge_type="static", **TINY),
                                   lambda: DrinConfig(gcn_edge_feature="vector", **TINY),
                                   lambda: wikimel_config(max_entity_attr_token_len=8)],
                         ids=["wikidiverse", "static", "vector", "wikimel"])
def test_gradients_are_views_of_one_flat_bucket(maker):
    """backward writes into ONE flat bucket: every .grad is a view of it at its slot, dead parameters have none, the values
    equal those of the per-tensor path, and a second backward before zero_grad accumulates like torch (fresh bucket)."""
    cfg = maker()
    model, batch, sd = _train_setup(cfg, 6)
    plain, _, _ = _train_setup(cfg, 6, grad_bucket=False)
    _step_loss(model, batch, cfg).backward()
    _step_loss(plain, batch, cfg).backward()
    flat = model.grad_bucket()
    assert flat is not None and flat.data_ptr() == model._grad_flat.data_ptr()
    offsets, live, total = model.bucket_layout()
    n_live = 0
    for o, p, q in zip(offsets, _param_list(model), _param_list(plain)):
        assert (p.grad is None) == (q.grad is None)
        if p.grad is None:
            assert o >= live                                          # dead parameters sit behind the all-reduced prefix
            continue
        assert p.grad.data_ptr() == flat.data_ptr() + 4 * o and o + p.numel() <= live
        # same kernels, same order - except that the weight-gradient atomics of small products may differ in the last bits
        assert (p.grad - q.grad).abs().max().item() <= 2e-5 * q.grad.abs().max().item() + 1e-12
        n_live += p.numel()
    assert n_live <= live < n_live + 64 * len(offsets)
    bucket = GradBucket(list(model.parameters()))
    assert bucket._aliased_bucket([p for p in bucket.params if p.grad is not None]) is not None
    # accumulate a second backward: the bucket the .grads alias must not be zeroed under them
    first = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    _step_loss(model, batch, cfg).backward()
    for k, p in model.named_parameters():
        if p.grad is not None:
            assert (p.grad - 2 * first[k]).abs().max().item() <= 1e-4 * first[k].abs().max().item() + 1e-12, k


def test_data_parallel_step_through_the_hip_model_equals_the_full_batch():
    """SURVEY.md 8e with the HIP Model and its flat bucket (world = 1, the collective injected as "sum of shards / 2"):
    two half-batches, each with its own per-rank loss, bucket-averaged == the mean of the two half-batch gradients computed
    tensor by tensor, and the one-launch Adam on the averaged bucket == torch.optim.Adam on those means."""
    cfg = wikimel_config(max_entity_attr_token_len=8, batch_size=4)
    model, batch, sd = _train_setup(cfg, 8, precision="f32")
    halves = [[t[:4] for t in batch], [t[4:] for t in batch]]
    buckets = []
    for h in halves:                                                    # what rank r computes
        model.zero_grad(set_to_none=True)
        _step_loss(model, h, cfg).backward()
        buckets.append(model.grad_bucket().clone())
    ref, _, _ = _train_setup(cfg, 8, precision="f32", grad_bucket=False)
    mean = {}
    for h in halves:
        ref.zero_grad(set_to_none=True)
        _step_loss(ref, h, cfg).backward()
        for k, p in ref.named_parameters():
            if p.grad is not None:
                mean[k] = mean.get(k, 0) + p.grad / 2
    # the all-reduce (mean) acts on the bucket in place: emulate it on rank 1's live bucket
    live = model.grad_bucket()
    live.copy_((buckets[0] + buckets[1]) / 2)
    for k, p in model.named_parameters():
        assert (p.grad is None) == (k not in mean), k
        if p.grad is not None:
            assert (p.grad - mean[k]).abs().max().item() <= 2e-5 * mean[k].abs().max().item() + 1e-12, k
    assert {k for k, p in model.named_parameters() if p.grad is None} == {f"gcn_layers.1.w_{x}.{y}" for x in "uv" for y in ("weight", "bias")}
    opt = make_adam(model, 1e-3)
    assert isinstance(opt, LibraryAdam)
    # torch.optim.Adam on the SAME averaged gradients (a near-zero gradient entry moves by +-lr on the sign of its last bit,
    # so the two models' own gradients - equal to 2e-5 - would not do): bit-identical parameters
    topt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    for (k, p), (_, q) in zip(ref.named_parameters(), model.named_parameters()):
        p.grad = None if q.grad is None else q.grad.clone()
    topt.step()
    version = model.gcn_layers[0].w_h.weight._version
    opt.step()
    assert opt.one_launch_steps == 1 and model.gcn_layers[0].w_h.weight._version > version
    for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        assert torch.equal(p, q), k
    # the parameters now live in one flat bucket too; state_dict / load_state_dict are unaffected
    flat = model.flatten_parameters()
    assert all(p.data_ptr() >= flat.data_ptr() and p.data_ptr() < flat.data_ptr() + 4 * flat.numel() for p in model.parameters())
    model.load_state_dict(sd)
    assert torch.equal(model.gcn_layers[0].w_h.weight, sd["gcn_layers.0.w_h.weight"].to(DEV))
    with torch.no_grad():
        s1 = model.eval()(batch[:14])
    fresh = Model(cfg, precision="f32").to(DEV).eval()
    fresh.load_state_dict(sd)
    with torch.no_grad():
        assert torch.equal(s1, fresh(batch[:14]))


def test_library_adam_loop_tracks_torch_adam_loop():
    """Five optimisation steps of the same model / data with LibraryAdam and with torch.optim.Adam: identical weights
    (bit for bit when the default arithmetic matches torch's - asserted separately - and the gradients are deterministic)."""
    cfg = DrinConfig(**TINY)
    a, batch, _ = _train_setup(cfg, 16, precision="f32")
    b, _, _ = _train_setup(cfg, 16, precision="f32")
    oa, ob = make_adam(a, 1e-3), make_adam(b, 1e-3, library=False)
    assert isinstance(ob, torch.optim.Adam)
    for _ in range(5):
        for m, o in ((a, oa), (b, ob)):
            o.zero_grad(set_to_none=True)
            _step_loss(m, batch, cfg).backward()
            o.step()
    assert oa.one_launch_steps == 5
    # the optimiser state travels: a fresh LibraryAdam loaded with it continues the same trajectory
    c, _, _ = _train_setup(cfg, 16, precision="f32")
    c.load_state_dict(a.state_dict())
    oc = make_adam(c, 1e-3)
    oc.load_state_dict(oa.state_dict())
    for m, o in ((a, oa), (c, oc)):
        o.zero_grad(set_to_none=True)
        _step_loss(m, batch, cfg).backward()
        o.step()
    assert oc.t == 6 and all(torch.equal(p, q) for p, q in zip(a.parameters(), c.parameters()))
    ob.zero_grad(set_to_none=True)
    _step_loss(b, batch, cfg).backward()
    ob.step()
    for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        # identical optimiser arithmetic on identical gradients (no atomics in the backward pass since round 3): identical bits
        assert torch.equal(p, q), k


def test_frozen_parameters_take_the_per_tensor_adam_path():
    cfg = DrinConfig(**TINY)
    model, batch, _ = _train_setup(cfg, 8)
    model.vertex_encoder.mention_image_linear.weight.requires_grad_(False)
    before = model.vertex_encoder.mention_image_linear.weight.clone()
    w0 = model.gcn_layers[0].w_h.weight.clone()
    opt = make_adam(model, 1e-2)
    opt.zero_grad()
    _step_loss(model, batch, cfg).backward()
    opt.step()
    assert opt.one_launch_steps == 0
    assert torch.equal(model.vertex_encoder.mention_image_linear.weight, before)
    assert not torch.equal(model.gcn_layers[0].w_h.weight, w0)


# ---- calls above 32 768 mentions are split (launch_entity_stream refused B > 65 535) ------------------------
def test_large_batches_are_split_into_calls():
    cfg = DrinConfig(num_candidates_data=3, **TINY)
    sd = synth.make_state_dict(cfg, 8)
    model = Model(cfg).to(DEV).eval()
    model.load_state_dict(sd)
    B = 70_000                                                       # > 65 535: the grids of the row kernels stop there
    batch = synth.make_device_batch(cfg, B, 9, DEV)[:14]
    with torch.no_grad():
        out = model(batch)
        assert out.shape == (B, 4) and torch.isfinite(out).all()
        for rows in (slice(0, 5), slice(32766, 32771), slice(B - 3, B)):    # mention independence: any slice alone
            # (a handful of mentions takes other tile shapes / K-splits than 32 768: equal to fp32 re-association)
            assert (out[rows] - model([t[rows] for t in batch])).abs().max().item() <= 5e-6
        ref = O.forward(sd, [t[32760:32776].cpu() for t in batch])
        assert (out[32760:32776].cpu() - ref).abs().max().item() <= 1e-5
        # table form splits too
        tab = synth.make_device_batch(cfg.with_(num_candidates_data=499), 1, 4, DEV)
        table = EntityTable(tab[7][0], None, tab[9][0], tab[10][0], tab[11][0])
        cand = torch.randint(0, 500, (B, 4), device=DEV)
        ib = IndexedBatch(batch[:7], table, cand, batch[12], batch[13])
        big = model(ib)
        sub = IndexedBatch([t[40000:40004] for t in batch[:7]], table, cand[40000:40004], batch[12][40000:40004], batch[13][40000:40004])
        assert (big[40000:40004] - model(sub)).abs().max().item() <= 5e-6
        ref = O.forward(sd, [t.cpu() for t in sub.gathered()])
        assert (big[40000:40004].cpu() - ref).abs().max().item() <= 1e-5


# ---- BASELINE config 5 at full width (only ever tested at D = 64) -------------------------------------------
@pytest.fixture(scope="module")
def million_entity_table():
    cfg = DrinConfig(num_candidates_data=1000)
    E, D, R = 1_000_003, cfg.bert_embed_dim, cfg.resnet_embed_dim
    g = torch.Generator(device=DEV).manual_seed(7)
    table = EntityTable(torch.randn(E, D, device=DEV, generator=g), None, torch.randn(E, R, device=DEV, generator=g),
                        torch.randn(E, 1, R, device=DEV, generator=g), torch.rand(E, 1, device=DEV, generator=g))
    yield cfg, table
    table.invalidate()
    del table
    torch.cuda.empty_cache()


@pytest.mark.parametrize("path", ["fused", "cache", "exact_f32"])
def test_config5_full_width_against_the_oracle(million_entity_table, path):
    """N = 1001 candidates per mention gathered from a 1 000 003-row table at D = 768 / R = 2048 (workgroups of 16
    candidates in k_entity_stream at this call size, entity_index up to 2^20, k_cached_pairs over 23.5 KB cache rows): three mentions'
    slices against the CPU oracle on the gathered rows; the remaining mentions against each other across paths."""
    cfg, table = million_entity_table
    sd = synth.make_state_dict(cfg, 7)
    B, N, E = 37, cfg.num_candidates_model, table.num_entities
    g = torch.Generator(device=DEV).manual_seed(11)
    men = synth.make_device_batch(cfg.with_(num_candidates_data=0), B, 12, DEV)
    cand = torch.randint(0, E, (B, N), device=DEV, generator=g)
    cand[0, :3] = torch.tensor([0, E - 1, E - 2], device=DEV)          # first / last rows of the table
    sims = 20.0 + 5.0 * torch.randn(2, B, N, device=DEV, generator=g)
    ib = IndexedBatch(men[:7], table, cand, sims[0], sims[1])
    model = Model(cfg, precision="f32" if path == "exact_f32" else "bf16x3").to(DEV).eval()
    model.load_state_dict(sd)
    table.enable_cache(path == "cache")
    with torch.no_grad():
        got = model(ib)
        again = model(ib)
    table.enable_cache(False)
    assert got.shape == (B, N) and torch.isfinite(got).all() and torch.equal(got, again)
    rows = [0, 17, B - 1]
    sub = IndexedBatch([t[rows] for t in men[:7]], table, cand[rows], sims[0][rows], sims[1][rows])
    ref = O.forward(sd, [t.cpu() for t in sub.gathered()])
    err = (got[rows].cpu() - ref).abs().max().item()
    top1 = (got[rows, :-1].argmax(1).cpu() == ref[:, :-1].argmax(1)).float().mean().item()
    print(f"config 5 full width, {path}: max |score - oracle| = {err:.2e}, top-1 agreement {top1}")
    assert err <= 2e-5 and top1 == 1.0                                 # bar of the path: 1e-4
    if path == "cache":
        plain = Model(cfg).to(DEV).eval()
        plain.load_state_dict(sd)
        with torch.no_grad():
            assert (plain(ib) - got).abs().max().item() <= 1e-5         # every mention: cached == un-cached


def test_config5_timed_instantiation_128_candidate_workgroups(million_entity_table):
    """`bench.py`'s config-5 leg scores chunks of 4 096 mentions, where a workgroup of `k_cached_pairs` /
    `k_pair_final` walks 128 candidates (`csrc/entity_cache.hip` cached_chunk_default: B >= 2 048) - every other full-width test
    of the cached path stayed at B = 37 (64-candidate workgroups).  B = 2 048 x N = 1 001 on the 1 000 003-row table through the
    per-entity cache: mentions (first, middle, last) against the CPU oracle on the gathered rows, <= 1e-5, and against the
    same mentions scored in a small call (64-candidate workgroups), <= 5e-6."""
    import os
    cfg, table = million_entity_table
    lib = _lib.load()
    sd = synth.make_state_dict(cfg, 7)
    B, N, E = 2048, cfg.num_candidates_model, table.num_entities

    def cfg_c(batch):
        c = _lib.DrinConfigC()
        _lib.check(lib.drin_default_config(C.byref(c)))
        c.batch, c.num_candidates, c.precision, c.num_entities = batch, N, _lib.PREC_BF16X3, E
        return c

    assert lib.drin_workgroups_per_mention(C.byref(cfg_c(B)), 1) == 8            # ceil(1001 / 128)
    assert lib.drin_workgroups_per_mention(C.byref(cfg_c(37)), 1) == 16          # ceil(1001 / 64)
    g = torch.Generator(device=DEV).manual_seed(17)
    men = synth.make_device_batch(cfg.with_(num_candidates_data=0), B, 18, DEV)
    cand = torch.randint(0, E, (B, N), device=DEV, generator=g)
    sims = 20.0 + 5.0 * torch.randn(2, B, N, device=DEV, generator=g)
    ib = IndexedBatch(men[:7], table, cand, sims[0], sims[1])
    model = Model(cfg).to(DEV).eval()
    model.load_state_dict(sd)
    table.enable_cache(True)
    try:
        with torch.no_grad():
            model(IndexedBatch([t[:2] for t in men[:7]], table, cand[:2], sims[0][:2], sims[1][:2]))   # folds the weights, builds the cache
            _lib.profile_begin()
            got = model(ib)
            prof = _lib.profile_end()
            assert prof["stream"][1] == 1 and prof["gemm_planes"][1] >= 1       # k_cached_pairs once, et' W_h2^T on planes
            assert got.shape == (B, N) and torch.isfinite(got).all() and torch.equal(got, model(ib))
            worst = small = 0.0
            torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
            for rows in ([0, 1], [B // 2, B // 2 + 1], [B - 2, B - 1]):
                sub = IndexedBatch([t[rows] for t in men[:7]], table, cand[rows], sims[0][rows], sims[1][rows])
                ref = O.forward(sd, [t.cpu() for t in sub.gathered()])
                worst = max(worst, (got[rows].cpu() - ref).abs().max().item())
                assert (got[rows, :-1].argmax(1).cpu() == ref[:, :-1].argmax(1)).all()
                small = max(small, (got[rows] - model(sub)).abs().max().item())
    finally:
        table.enable_cache(False)
    print(f"config 5, B = 2048 through the cache: max |score - oracle| {worst:.2e}, against the small call {small:.2e}")
    assert worst <= 1e-5 and small <= 5e-6


def test_table_edits_and_data_writes_are_seen_or_can_be_declared():
    """the cache keyed on the text pointer alone missed edits of the image / object tables; writes through
    `.data` bypass the version counter and need `invalidate()`."""
    cfg = DrinConfig(num_candidates_data=20, **TINY)
    sd = synth.make_state_dict(cfg, 8)
    E, B, N = 83, 5, cfg.num_candidates_model
    tab = synth.make_batch(cfg.with_(num_candidates_data=E - 1), 1, 71)
    table = EntityTable(tab[7][0], None, tab[9][0], tab[10][0], tab[11][0]).to(DEV).enable_cache()
    men = _to_dev(synth.make_batch(cfg, B, 72))
    cand = torch.randint(0, E, (B, N), generator=torch.Generator().manual_seed(3)).to(DEV)
    ib = IndexedBatch(men[:7], table, cand, men[12], men[13])
    model = Model(cfg).to(DEV).eval()
    model.load_state_dict(sd)

    def oracle():
        return O.forward({k: v.detach().cpu() for k, v in model.state_dict().items()}, [t.cpu() for t in ib.gathered()])

    with torch.no_grad():
        assert (model(ib).cpu() - oracle()).abs().max().item() <= 1e-5
        table.image.mul_(-0.5)                                        # in-place torch op on the IMAGE table: seen
        assert (model(ib).cpu() - oracle()).abs().max().item() <= 1e-5
        table.object.data.neg_()                                      # a .data write: invisible ...
        stale = (model(ib).cpu() - oracle()).abs().max().item()
        table.invalidate()                                            # ... until declared
        assert (model(ib).cpu() - oracle()).abs().max().item() <= 1e-5 < stale
        model.gcn_layers[0].w_h.weight.data.mul_(1.25)                # the same for the weights
        stale = (model(ib).cpu() - oracle()).abs().max().item()
        model.invalidate()
        assert (model(ib).cpu() - oracle()).abs().max().item() <= 1e-5 < stale
        model.load_state_dict(sd)                                     # load_state_dict invalidates by itself
        assert (model(ib).cpu() - oracle()).abs().max().item() <= 1e-5


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two visible GPUs")
def test_second_device_opts_into_large_lds():
    """the dynamic-LDS opt-in is per device; a process that ran on cuda:0 first must still launch on cuda:1."""
    cfg = wikimel_config(max_entity_attr_token_len=8)
    sd = synth.make_state_dict(cfg, 7)
    outs = []
    for d in ("cuda:0", "cuda:1"):
        m = Model(cfg).to(d).eval()
        m.load_state_dict(sd)
        with torch.no_grad():
            outs.append(m([t.to(d) for t in synth.make_batch(cfg, 300, 5)[:14]]).cpu())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("layers,mentions", [(3, 600), (5, 600), (5, 20)])
def test_deep_backward_at_512_mentions_keeps_its_operands(layers, mentions):
    """Regression (round 2): the split-bf16 dX = dY W products run against weights transposed ONCE per backward; a product that
    is transposed on the fly (W_u at >= 512 mentions: 2 B >= 1024 rows) must not overwrite one of them.  And the pair-sized
    weight gradients run as ONE group at the end of the pass (TnGroup: up to eight products; five layers have eleven, so the
    group goes in instalments) from per-level gradient buffers nothing may have overwritten.  Dynamic layers, 600 mentions,
    TINY widths raised to the split-bf16 kernel's range; every gradient against autograd through the oracle.  (20 mentions:
    every weight gradient is mention-sized - 22 exact-fp32 products through the F32GemmGroup of eight, in instalments too.)"""
    from drin_amd.metrics import TripletLoss
    cfg = DrinConfig(num_gcn_layers=layers, num_candidates_data=2, bert_embed_dim=128, gcn_embed_dim=128, resnet_embed_dim=128,
                     max_mention_sentence_len=8, resnet_num_region=2)
    sd = synth.make_state_dict(cfg, 9)
    batch = synth.make_batch(cfg, mentions, 45)
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref_loss = O.triplet_loss(batch[-1], O.forward(p, batch, **O.config_kwargs(cfg)), cfg.triplet_margin)
    ref = torch.autograd.grad(ref_loss, list(p.values()), allow_unused=True)
    model = Model(cfg, precision="bf16x3").to(DEV)
    model.load_state_dict(sd)
    dbatch = _to_dev(batch)
    loss = TripletLoss(cfg.triplet_margin)(dbatch[-1], model(dbatch[:-1]))
    loss.backward()
    assert abs(loss.item() - ref_loss.item()) <= 1e-5
    for (k, q), r in zip(model.named_parameters(), ref):
        assert (q.grad is None) == (r is None), k
        if r is not None:
            rel = (q.grad.cpu() - r).norm().item() / (r.norm().item() + 1e-12)
            assert rel <= 2e-4, (k, rel)


def test_bench_two_ranks_on_one_gpu_over_gloo():
    """The N > 1 code path of bench.py with the HIP Model on the GPU: two ranks launched by bench.py itself, both on device 0,
    talking over gloo (RCCL refuses two ranks on one device; everything else - launcher, barrier / max-over-ranks bracket on
    device tensors, flat gradient bucket all-reduced in place, one-launch Adam - is what an N-GPU run executes)."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DRIN_BENCH_SHARE_GPU="1", DRIN_BENCH_BACKEND="gloo")
    import tempfile
    full_path = os.path.join(tempfile.mkdtemp(prefix="drin_bench_"), "full.json")
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--mode", "train", "--batch", "16", "--steps", "3",
                        "--warmup", "2", "--legs-file", full_path], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096                  # ONE headline line the driver can parse (round 4's was 32 KB)
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and len(line["rank_ms_per_step"]) == 2 and line["config"]["global_batch"] == 32
    assert line["allreduce_ms"] > 0 and line["allreduce_bytes"] == 26775552 and 0 < line["final_loss"] < 1
    assert line["collective"]["world"] == 2 and len(line["rank_roofline_avg_launch_ms"]) == 2 and "expected_weak_scaling_efficiency" in line["scaling_model"]
    assert "library Adam" in json.load(open(full_path))["optimizer"]  # the full record: in the legs file
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--batch", "64", "--steps", "2", "--warmup", "1",
                        "--legs", "none", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["pairs_per_step"] == 2 * 64 * 101 and line["value"] > 0


# ---- size-independent properties at BASELINE's full sizes (WikiMEL-shaped, 512 mentions x 101 candidates) --------------
@pytest.mark.parametrize("fused", [True, False], ids=["fused", "layerwise"])
def test_full_size_disabled_edges_cut_their_inputs_off(fused):
    """gcn_edge_enabled = (1, 0, 1, 0) (model.py:122) multiplies the ti and ii edges by zero: the scores must not depend on
    the CLIP text-image similarities nor on any object feature / score - replacing them with other finite values leaves
    every score bit-identical on either path; the enabled edges' inputs do matter."""
    cfg = wikimel_config(gcn_edge_enabled=(1, 0, 1, 0), max_entity_attr_token_len=16)
    sd = synth.make_state_dict(cfg, 7)
    model = Model(cfg, fused=fused).to(DEV).eval()
    model.load_state_dict(sd)
    B = 512
    batch = synth.make_device_batch(cfg, B, 21, DEV)[:14]
    g = torch.Generator(device=DEV).manual_seed(3)
    other = list(batch)
    other[13] = 20.0 + 5.0 * torch.randn(batch[13].shape, device=DEV, generator=g)            # mtei similarity -> ti edge
    for i in (5, 10):                                                                          # mention / entity object features
        other[i] = torch.randn(batch[i].shape, device=DEV, generator=g)
    for i in (6, 11):                                                                          # their scores
        other[i] = torch.rand(batch[i].shape, device=DEV, generator=g)
    with torch.no_grad():
        a, b = model(batch), model(other)
        assert torch.equal(a, b)
        moved = list(batch)
        moved[12] = batch[12] + 1.0                                                            # miet similarity -> the ENABLED it edge
        assert (model(moved) - a).abs().max().item() > 1e-5
        ref = O.forward(sd, [t[:2].cpu() for t in batch], edge_enabled=cfg.gcn_edge_enabled)
        assert (a[:2].cpu() - ref).abs().max().item() <= 1e-5


def test_full_size_cosine_edges_ignore_positive_rescaling():
    """The two cosine edges (model.py:71-92) are invariant to a positive rescaling of the rows they compare: scaling every
    entity OBJECT row by a power of two (exact in fp32) leaves the ii edge, hence every score, bit-identical."""
    cfg = wikimel_config(max_entity_attr_token_len=16)
    sd = synth.make_state_dict(cfg, 7)
    model = Model(cfg).to(DEV).eval()
    model.load_state_dict(sd)
    batch = synth.make_device_batch(cfg, 512, 22, DEV)[:14]
    scaled = list(batch)
    scaled[10] = batch[10] * 4.0
    scaled[5] = batch[5] * 0.25
    with torch.no_grad():
        assert torch.equal(model(batch), model(scaled))


@pytest.mark.parametrize("dataset", ["wikidiverse", "wikimel"])
def test_train_cli_runs_end_to_end(tmp_path, dataset, monkeypatch):
    """`python -m drin_amd.train` (the Lightning-free `train.py`): synthetic preprocessed directory -> loaders or device-resident
    splits -> HIP forward / backward -> library loss + Adam -> per-sample test dump (args.output_test_result)."""
    from drin_amd import train as T
    from drin_amd.data import write_synthetic_dataset
    cfg = wikimel_config(batch_size=8, num_epoch=2, test_epoch_interval=1) if dataset == "wikimel" else (
        DrinConfig(batch_size=8, num_epoch=2, test_epoch_interval=1))
    write_synthetic_dataset(cfg, str(tmp_path), sizes=(24, 8, 8), seed=3, num_entities=300, lean=True)
    dump = str(tmp_path / "test-result.txt")
    monkeypatch.setattr(T, "wikimel_config", lambda **kw: cfg, raising=False)
    monkeypatch.setattr("drin_amd.config.wikimel_config", lambda **kw: cfg)
    monkeypatch.setattr(T, "DrinConfig", lambda **kw: cfg)
    lines = []
    monkeypatch.setattr("builtins.print", lambda *a, **k: lines.append(" ".join(map(str, a))))
    monkeypatch.chdir(tmp_path)
    T.main(["--data", str(tmp_path), "--dataset", dataset, "--on-device", "--output-test-result", dump]
           + (["--profiling"] if dataset == "wikimel" else []))
    assert sum("test after epoch" in ln for ln in lines) == 2 and sum(ln.startswith("epoch ") for ln in lines) == 2
    # train.py:126-133 (every setting, strings quoted) and the epoch banners of train.py:72-77
    assert lines[0] == "=============== parameters ===============" and f"dataset_name '{dataset}'" in lines  # (the patched print joins with a blank)
    assert sum("***** Epoch" in ln and " - training - " in ln for ln in lines) == 2 and sum(" - testing - " in ln for ln in lines) == 2
    if dataset == "wikimel":   # args.profiling (train.py:64-70): wait 1, warmup 1, then the third (last) step of each fit is timed
        import json
        rep = json.load(open(tmp_path / "log" / "profiler" / "drin_profile_0.json"))
        assert rep["steps"] == 1 and rep["kernel_ms_per_step"]["optim"] > 0 and rep["launches_per_step"]["optim"] == 1
        assert sum(rep["launches_per_step"].values()) >= 30 and sum(ln.startswith("profile cycle 0") for ln in lines) == 2
    rows = open(dump).read().splitlines()
    assert rows.count("==========  Test ==========") == 2          # on_test_epoch_start (train.py:92-95), once per test pass
    rows = [r for r in rows if r != "==========  Test =========="]
    import re
    assert sum(bool(re.match(r"^\d+:\t\[", r)) for r in rows) == 2 * 8   # two test passes x 8 mentions (the answer rows' repr may wrap)
    assert rows[0].startswith("0:\t[") and len(eval(rows[0].split(":\t", 1)[1])) == cfg.num_candidates_model


def test_config5_mixed_f16_cache_rows_every_score_of_a_chunk(million_entity_table):
    """`EntityTable.enable_cache(format="mixed_f16")` at config 5's own shape (2 048 mentions x 1 001 candidates over the
    1 000 003-row table, 128-candidate workgroups of `k_cached_pairs<3, 8, exact, MIXED>`): ALL 2 050 048 scores against the
    fp32 cache rows - the storage format's own effect, emulated at 2e-7 for 101 candidates - top-1 unchanged on every mention; three slices against the CPU oracle."""
    import os
    cfg, table = million_entity_table
    sd = synth.make_state_dict(cfg, 7)
    B, N, E = 2048, cfg.num_candidates_model, table.num_entities
    g = torch.Generator(device=DEV).manual_seed(23)
    men = synth.make_device_batch(cfg.with_(num_candidates_data=0), B, 24, DEV)
    cand = torch.randint(0, E, (B, N), device=DEV, generator=g)
    sims = 20.0 + 5.0 * torch.randn(2, B, N, device=DEV, generator=g)
    ib = IndexedBatch(men[:7], table, cand, sims[0], sims[1])
    model = Model(cfg).to(DEV).eval()
    model.load_state_dict(sd)
    try:
        with torch.no_grad():
            table.enable_cache(True)
            full = model(ib)
            table.enable_cache(True, format="mixed_f16")
            mixed = model(ib)
            assert table._cache.numel() == E * 16400 and torch.equal(mixed, model(ib))
            err = (mixed - full).abs().max().item()
            flips = int((mixed[:, :-1].argmax(1) != full[:, :-1].argmax(1)).sum())
            torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
            worst = 0.0
            for rows in ([0, 1], [B // 2], [B - 1]):
                sub = IndexedBatch([t[rows] for t in men[:7]], table, cand[rows], sims[0][rows], sims[1][rows])
                ref = O.forward(sd, [t.cpu() for t in sub.gathered()])
                worst = max(worst, (mixed[rows].cpu() - ref).abs().max().item())
    finally:
        table.enable_cache(False)
    print(f"config 5, mixed-f16 cache rows: max |score - fp32 rows| over {B * N} scores {err:.2e}, top-1 flips {flips}, vs oracle {worst:.2e}")
    assert err <= 2e-6 and flips == 0 and worst <= 1e-5
