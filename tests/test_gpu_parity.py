"""-m gpu: the HIP path (through the C ABI of libdrin_hip.so) against the CPU oracle and the golden
vectors of the reference.  Tolerance of the path per BASELINE.json north_star: 1e-4 fp32 on scores."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from drin_amd import _lib, synth
from drin_amd.config import DrinConfig, wikimel_config
from drin_amd.model import Model
from oracle import drin_oracle as O
from oracle.cases import CASES, TINY, build_case

pytestmark = pytest.mark.gpu
SCORE_TOL = 1e-4   # north_star: "within 1e-4 fp32"
DEV = "cuda"


def _golden(golden_dir, name):
    return np.load(os.path.join(golden_dir, f"{name}.npz"))


def _model(cfg, sd):
    m = Model(cfg, precision="f32").to(DEV)
    m.load_state_dict(sd)
    return m.eval()


def _to_dev(batch):
    return [t.to(DEV) for t in batch]


# ---- building blocks through the C ABI ---------------------------------------------------------------
@pytest.mark.parametrize("rows,n_out,k", [(1, 64, 64), (5, 64, 128), (44, 768, 768), (129, 768, 2048), (404, 768, 768),
                                          (1000, 200, 96), (128, 128, 32), (257, 130, 40)])
def test_linear_fwd(rows, n_out, k):
    lib = _lib.load()
    g = torch.Generator().manual_seed(rows * 7 + k)
    x = torch.randn(rows, k, generator=g)
    w = torch.randn(n_out, k, generator=g) / k ** 0.5
    b = torch.randn(n_out, generator=g)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    y = torch.full((rows, n_out), float("nan"), device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.drin_linear_fwd(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), rows, n_out, k, 0, st))
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    err = (y.cpu().double() - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err
    # exact-fp32 claim: a permuted fmaf chain is as accurate as torch's own fp32 matmul
    err32 = (torch.nn.functional.linear(x, w, b).double() - ref).abs().max().item()
    assert err <= 4 * err32 + 1e-6


@pytest.mark.parametrize("rows,n_out,k", [(1, 64, 64), (300, 768, 768), (1024, 768, 2048), (2500, 200, 96), (513, 258, 40)])
def test_linear_fwd_bf16x3(rows, n_out, k):
    """split-bf16 contraction: error ~1e-5 of sum |x||w| per output (3 of the 4 partial products kept)."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(rows + k)
    x = torch.randn(rows, k, generator=g)
    w = torch.randn(n_out, k, generator=g) / k ** 0.5
    b = torch.randn(n_out, generator=g)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    y = torch.full((rows, n_out), float("nan"), device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.drin_linear_fwd(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), rows, n_out, k,
                                   _lib.PREC_BF16X3_ALL, st))
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    scale = torch.nn.functional.linear(x.abs().double(), w.abs().double())
    rel = ((y.cpu().double() - ref).abs() / scale).max().item()
    print(f"bf16x3 {rows}x{n_out}x{k}: max err / sum|x||w| = {rel:.2e}")
    assert rel <= 2e-5


@pytest.mark.parametrize("rows,n_out,k", [(7, 64, 64), (300, 768, 768), (1000, 768, 2048), (513, 258, 96)])
def test_linear_planes_fwd(rows, n_out, k):
    """drin_split_planes + drin_linear_planes_fwd (LDS-DMA split-bf16 GEMM) against fp64."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(rows + 3 * k)
    x = torch.randn(rows, k, generator=g)
    w = torch.randn(n_out, k, generator=g) / k ** 0.5
    b = torch.randn(n_out, generator=g)
    st = torch.cuda.current_stream().cuda_stream
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    planes = [torch.empty(t.shape, dtype=torch.bfloat16, device=DEV) for t in (x, x, w, w)]
    _lib.check(lib.drin_split_planes(xd.data_ptr(), planes[0].data_ptr(), planes[1].data_ptr(), xd.numel(), st))
    _lib.check(lib.drin_split_planes(wd.data_ptr(), planes[2].data_ptr(), planes[3].data_ptr(), wd.numel(), st))
    # hi + lo reproduces the fp32 value to ~2^-17 relative
    rec = planes[0].float() + planes[1].float()
    assert ((rec - xd).abs() <= xd.abs() * 2.0 ** -16 + 1e-30).all()
    y = torch.full((rows, n_out), float("nan"), device=DEV)
    _lib.check(lib.drin_linear_planes_fwd(*(p.data_ptr() for p in planes), bd.data_ptr(), y.data_ptr(), rows, n_out, k, st))
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    scale = torch.nn.functional.linear(x.abs().double(), w.abs().double())
    rel = ((y.cpu().double() - ref).abs() / scale).max().item()
    print(f"planes {rows}x{n_out}x{k}: max err / sum|x||w| = {rel:.2e}")
    assert rel <= 2e-5


@pytest.mark.parametrize("rows,n_out,k", [(70, 64, 64), (1500, 768, 768), (4133, 768, 2048), (2048, 384, 768), (3000, 200, 96),
                                          (1024, 128, 128), (9000, 768, 768),
                                          # 303 / 354 tiles of 256 x 256 on 256 CUs: the last round splits K in 4 / 2 (two-stage only:
                                          # the split needs scratch); 2048-wide reduction of dx: 64 K-blocks
                                          (25856, 768, 768), (30000, 768, 768), (25856, 2048, 768)])
@pytest.mark.parametrize("precision", ["f32", "bf16x3", "bf16x3_two_stage"])
def test_linear_bwd(rows, n_out, k, precision):
    """dx = dy W, dW += dy^T x, db += colsum(dy) - exact fp32 MFMA and split-bf16 (transposing-stager kernel of
    gemm_tn_bf16x3.hip; NT kernel against W^T) - against float64 on the host; dW / db accumulate."""
    if rows > 20000 and precision == "f32":
        pytest.skip("tail-split cases: split-bf16 only")
    g = torch.Generator().manual_seed(rows + n_out + k)
    x, w, dy = torch.randn(rows, k, generator=g), torch.randn(n_out, k, generator=g) / k ** 0.5, torch.randn(rows, n_out, generator=g)
    dw0, db0 = torch.randn(n_out, k, generator=g), torch.randn(n_out, generator=g)
    xd, wd, dyd = x.to(DEV), w.to(DEV), dy.to(DEV)
    dx, dw, db = torch.empty(rows, k, device=DEV), dw0.to(DEV), db0.to(DEV)
    two_stage = precision == "bf16x3_two_stage"   # slices stored to scratch and added in order instead of atomics
    scratch = torch.empty(n_out * k * (28 if two_stage else 1), device=DEV)
    prec = {"f32": _lib.PREC_F32, "bf16x3": _lib.PREC_BF16X3, "bf16x3_two_stage": _lib.PREC_BF16X3}[precision]
    lib = _lib.load()

    def run():
        dw.copy_(dw0)
        db.copy_(db0)
        _lib.check(lib.drin_linear_bwd(xd.data_ptr(), wd.data_ptr(), dyd.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                       rows, n_out, k, prec, scratch.data_ptr(), scratch.numel(), torch.cuda.current_stream().cuda_stream))

    _lib.profile_begin()
    run()
    prof = _lib.profile_end()
    big = precision != "f32" and rows >= 1024 and n_out >= 128 and k >= 128
    if two_stage and big:
        first = dw.clone()
        run()
        assert torch.equal(first, dw), "the two-stage reduction is bit-reproducible"
    assert (prof["gemm_x3"][1] > 0) == big, prof
    ref_dx = (dy.double() @ w.double())
    ref_dw = dw0.double() + dy.double().t() @ x.double()
    ref_db = db0.double() + dy.double().sum(0)
    tol = 2e-5 if precision != "f32" else 5e-6
    for got, ref, scale in ((dx, ref_dx, (dy.abs().double() @ w.abs().double())), (dw, ref_dw, dy.abs().double().t() @ x.abs().double()),
                            (db, ref_db, dy.abs().double().sum(0))):
        err = ((got.cpu().double() - ref).abs() / (scale + 1e-30)).max().item()
        assert err <= tol, (precision, err)


def test_linear_fwd_errors():
    lib = _lib.load()
    x = torch.zeros(4, 6, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.drin_linear_fwd(x.data_ptr(), x.data_ptr(), None, x.data_ptr(), 4, 4, 6, 0, st) == _lib.E_SHAPE   # K % 4
    assert b"multiple of 4" in lib.drin_last_error()
    assert lib.drin_linear_fwd(None, x.data_ptr(), None, x.data_ptr(), 4, 4, 8, 0, st) == _lib.E_NULL


@pytest.mark.parametrize("name", ["tiny_wd", "tiny_wm", "wm_b2", "wd_b4"])
def test_pool_and_edges(name):
    """drin_pool_fwd / drin_edges_fwd == ghmfc.py:54-60,245-249 and model.py:60-94,201-204."""
    lib = _lib.load()
    cfg, sd, batch = build_case(name)
    from drin_amd.model import _Call
    call = _Call(cfg, _to_dev(batch), 0)
    B, N, D, R = call.B, call.N, call.D, cfg.resnet_embed_dim
    st = torch.cuda.current_stream().cuda_stream
    edges = torch.full((4, B, N), float("nan"), device=DEV)
    span = torch.full((B, D), float("nan"), device=DEV)
    _lib.check(lib.drin_edges_fwd(C.byref(call.cfg), C.byref(call.batch), edges.data_ptr(), span.data_ptr(), st))
    mtet, miei = O.edge_encoder(batch)
    np.testing.assert_allclose(span.cpu().numpy(), O.span_mean(batch[0], batch[2], batch[3]).numpy(), atol=1e-6)
    np.testing.assert_allclose(edges[0].cpu().numpy(), mtet.numpy(), atol=2e-6)
    np.testing.assert_allclose(edges[1].cpu().numpy(), (batch[13] / 100).numpy(), atol=0)
    np.testing.assert_allclose(edges[2].cpu().numpy(), (batch[12] / 100).numpy(), atol=0)
    np.testing.assert_allclose(edges[3].cpu().numpy(), miei.numpy(), atol=2e-6)
    mimg = torch.full((B, R), float("nan"), device=DEV)
    eimg = torch.full((B, N, R), float("nan"), device=DEV)
    xet = torch.full((B, N, D), float("nan"), device=DEV) if cfg.token_level_entities else None
    _lib.check(lib.drin_pool_fwd(C.byref(call.cfg), C.byref(call.batch), xet.data_ptr() if xet is not None else None,
                                 mimg.data_ptr(), eimg.data_ptr(), st))
    np.testing.assert_allclose(mimg.cpu().numpy(), batch[4].mean(-2).numpy(), atol=1e-6)
    e = batch[9].mean(-2) if batch[9].dim() == 4 else batch[9]
    np.testing.assert_allclose(eimg.cpu().numpy(), e.numpy(), atol=1e-6)
    if xet is not None:
        np.testing.assert_allclose(xet.cpu().numpy(), O.entity_token_mean(batch[7], batch[8]).numpy(), atol=1e-6)


def test_entity_token_mean_slice_corner_cases():
    """ntok in {0, 1, 2, 3, T}: python slice semantics of ghmfc.py:249, NaN for empty slices."""
    lib = _lib.load()
    cfg = DrinConfig(dataset_name="wikimel", num_candidates_data=4, max_entity_attr_token_len=6, **TINY)
    batch = synth.make_batch(cfg, 2, 31)
    ntoks = [[0, 1, 2, 3, 6], [6, 5, 4, 3, 2]]
    for b in range(2):
        for n in range(5):
            batch[8][b, n] = 0
            batch[8][b, n, : ntoks[b][n]] = 1
    from drin_amd.model import _Call
    call = _Call(cfg, _to_dev(batch), 0)
    xet = torch.zeros(2, 5, 64, device=DEV)
    _lib.check(lib.drin_pool_fwd(C.byref(call.cfg), C.byref(call.batch), xet.data_ptr(), None, None,
                                 torch.cuda.current_stream().cuda_stream))
    ref = O.entity_token_mean(batch[7], batch[8])
    got = xet.cpu()
    assert torch.equal(torch.isnan(got), torch.isnan(ref))
    np.testing.assert_allclose(torch.nan_to_num(got).numpy(), torch.nan_to_num(ref).numpy(), atol=1e-6)


# ---- the whole path against the reference's golden vectors -----------------------------------------------
@pytest.mark.parametrize("name", list(CASES))
def test_forward_matches_reference_golden(golden_dir, name):
    cfg, sd, batch = build_case(name)
    g = _golden(golden_dir, name)
    model = _model(cfg, sd)
    out = {k: v.cpu() for k, v in model.forward_traced(_to_dev(batch)).items()}
    err = np.abs(out["scores"].numpy() - g["scores"]).max()
    print(f"{name}: max |score - reference| = {err:.3e}")
    assert err <= SCORE_TOL
    assert err <= 2e-5, "fp32 MFMA path should sit at fp32 rounding level, far inside the 1e-4 budget"
    full = CASES[name][4]
    nl = cfg.num_gcn_layers
    e0 = out["edges0"]
    if cfg.gcn_edge_feature == "vector":      # model.py:182-183 repeats the scalar edge along the embedding
        assert e0.shape == (4, e0.shape[1], cfg.num_candidates_model, cfg.gcn_embed_dim)
        assert torch.equal(e0, e0[..., :1].expand_as(e0))
        e0 = e0[..., 0]
    np.testing.assert_allclose(e0[0].numpy(), g["mtet"], atol=2e-6)
    np.testing.assert_allclose(e0[3].numpy(), g["miei"], atol=2e-6)
    for l in range(nl + 1):
        np.testing.assert_allclose(out[f"mt{l}"].numpy(), g[f"mt{l}"], atol=2e-5, rtol=1e-4)
        np.testing.assert_allclose(out[f"mi{l}"].numpy(), g[f"mi{l}"], atol=2e-5, rtol=1e-4)
        for nm in ("et", "ei"):
            t = out[f"{nm}{l}"]
            np.testing.assert_allclose((t if full else t[0]).numpy(), g[f"{nm}{l}"], atol=2e-5, rtol=1e-4)
            assert abs(t.double().norm().item() - float(g[f"{nm}{l}_l2"])) <= 1e-5 * float(g[f"{nm}{l}_l2"])
        if l > 0:
            np.testing.assert_allclose(out[f"edges{l}"].numpy(), g[f"edges{l}"], atol=5e-6)
    # the Module call on the generic path (dead work of the last layer skipped) gives the same scores
    unfused = Model(cfg, precision="f32", fused=False).to(DEV).eval()
    unfused.load_state_dict(sd)
    with torch.no_grad():
        s2 = unfused(_to_dev(batch)).cpu()
        s3 = model(_to_dev(batch)).cpu()       # fused two-layer path where the geometry allows it
    # (equal to fp32 re-association: with the last layer's image vertices skipped a product of one or two rows takes the
    #  row-vector kernel where the traced call's four rows take the MFMA tile kernel)
    assert (s2 - out["scores"]).abs().max().item() <= 2e-6
    err3 = np.abs(s3.numpy() - g["scores"]).max()
    print(f"{name}: fused-path max |score - reference| = {err3:.3e}")
    assert err3 <= 2e-5


@pytest.mark.parametrize("name", ["wd_b4", "wm_b2", "tiny_wd", "tiny_wm_n37"])
def test_forward_bf16x3_matches_reference_golden(golden_dir, name):
    """The split-bf16 contraction path against the reference's fp32 forward: same 1e-4 bar, and a 1e-5
    guard (measured ~1e-6)."""
    cfg, sd, batch = build_case(name)
    g = _golden(golden_dir, name)
    model = Model(cfg, precision="bf16x3_all").to(DEV).eval()
    model.load_state_dict(sd)
    with torch.no_grad():
        s = model(_to_dev(batch)).cpu().numpy()
    err = np.abs(s - g["scores"]).max()
    print(f"{name} bf16x3: max |score - reference| = {err:.3e}")
    assert err <= SCORE_TOL and err <= 1e-5


def test_bf16x3_reference_batch_vs_oracle():
    cfg = wikimel_config()
    sd = synth.make_state_dict(cfg, 7)
    batch = synth.make_batch(cfg, 16, 45)
    ref = O.forward(sd, batch)
    model = Model(cfg, precision="bf16x3").to(DEV).eval()
    model.load_state_dict(sd)
    with torch.no_grad():
        got = model(_to_dev(batch)).cpu()
    err = (got - ref).abs().max().item()
    print(f"wikimel B=16 bf16x3: max err {err:.3e}")
    assert err <= 1e-5
    assert torch.equal(got[:, :-1].argmax(1), ref[:, :-1].argmax(1))


def test_empty_span_nan_row(golden_dir):
    cfg = DrinConfig(**TINY)
    model = _model(cfg, synth.make_state_dict(cfg, 8))
    batch = synth.make_batch(cfg, 3, 11)
    batch[3][1] = batch[2][1]
    with torch.no_grad():
        s = model(_to_dev(batch)).cpu().numpy()
    g = _golden(golden_dir, "tiny_wd_nan")["scores"]
    assert np.isnan(s[1]).all()
    np.testing.assert_allclose(s[[0, 2]], g[[0, 2]], atol=SCORE_TOL)


def test_same_seed_init_matches_reference(golden_dir):
    """train.py:134-136: seed -> Model() gives the reference's initial weights."""
    g = _golden(golden_dir, "init_order")
    torch.manual_seed(0)
    m = Model(DrinConfig())
    sd = m.state_dict()
    assert [k for k in sd] == [k[len("l2/"):] for k in g.files if k.startswith("l2/")]
    for k, v in sd.items():
        np.testing.assert_array_equal(v.flatten()[:8].numpy(), g[f"head/{k}"])
        assert abs(v.double().norm().item() - float(g[f"l2/{k}"])) < 1e-9


@pytest.mark.parametrize("maker,B", [(DrinConfig, 64), (wikimel_config, 8)])
def test_reference_batch_sizes_vs_oracle(maker, B):
    """Reference-sized batches (args.py:118,126) against the oracle on the same seeded inputs."""
    cfg = maker()
    sd = synth.make_state_dict(cfg, 7)
    batch = synth.make_batch(cfg, B, 41)
    ref = O.forward(sd, batch)
    with torch.no_grad():
        got = _model(cfg, sd)(_to_dev(batch)).cpu()
    err = (got - ref).abs().max().item()
    print(f"{cfg.dataset_name} B={B}: max err {err:.3e}")
    assert err <= SCORE_TOL
    # top-1 agreement (the candidate columns, answer slot dropped as common/utils.py:61-62)
    assert torch.equal(got[:, :-1].argmax(1), ref[:, :-1].argmax(1))


def test_full_size_properties():
    """BASELINE-sized WikiMEL batch drawn on the device: mention independence (a sub-batch scores
    bit-identically), candidate-permutation equivariance, and slices against the oracle."""
    cfg = wikimel_config()
    sd = synth.make_state_dict(cfg, 7)
    model = _model(cfg, sd)
    B = 96
    batch = synth.make_device_batch(cfg, B, 5, DEV)
    with torch.no_grad():
        full = model(batch)
        sub = model([t[32:40] for t in batch])
    assert torch.isfinite(full).all() and full.abs().max() <= 1 + 1e-5
    assert torch.equal(full[32:40], sub), "mentions must not interact"
    # permuting the candidates of every mention permutes its scores (the mean over n is symmetric)
    perm = torch.randperm(cfg.num_candidates_model, device=DEV)
    pb = list(batch)
    for i in (7, 8, 9, 10, 11, 12, 13):
        pb[i] = batch[i][:, perm].contiguous()
    with torch.no_grad():
        permuted = model(pb)
    assert (permuted - full[:, perm]).abs().max().item() <= 2e-6
    # three mentions against the oracle on the host
    idx = [0, 47, 95]
    host = [t[idx].cpu() for t in batch]
    ref = O.forward(sd, host)
    assert (full[idx].cpu() - ref).abs().max().item() <= SCORE_TOL


@pytest.mark.parametrize("maker,B", [(DrinConfig, 64), (wikimel_config, 8), (wikimel_config, 3)])
@pytest.mark.parametrize("precision", ["f32", "bf16x3_all"])
def test_fused_path_vs_oracle(maker, B, precision):
    """The folded inference path (csrc/fused_forward.hip) on reference-sized batches, both precisions."""
    cfg = maker()
    sd = synth.make_state_dict(cfg, 7)
    batch = synth.make_batch(cfg, B, 51)
    ref = O.forward(sd, batch)
    model = Model(cfg, precision=precision).to(DEV).eval()
    model.load_state_dict(sd)
    lib = _lib.load()
    from drin_amd.model import _Call
    assert lib.drin_fused_supported(C.byref(_Call(cfg, _to_dev(batch), 0).cfg)) == _lib.OK
    _lib.profile_begin(256)
    with torch.no_grad():
        got = model(_to_dev(batch)).cpu()
    prof = _lib.profile_end()
    assert prof["edge"][1] == 0 and prof["stream"][1] == 1, f"expected the fused launch sequence, got {prof}"
    err = (got - ref).abs().max().item()
    print(f"fused {cfg.dataset_name} B={B} {precision}: max err {err:.3e}")
    assert err <= (2e-6 if precision == "f32" else 1e-5)
    assert torch.equal(got[:, :-1].argmax(1), ref[:, :-1].argmax(1))


def test_fused_path_follows_weight_updates():
    """The folded weights are cached per weight version: an optimizer step must invalidate them."""
    cfg, sd, batch = build_case("tiny_wd")
    model = Model(cfg, precision="f32").to(DEV)
    model.load_state_dict(sd)
    dbatch = _to_dev(batch)
    with torch.no_grad():
        a = model(dbatch).clone()
    opt = torch.optim.SGD(model.parameters(), lr=0.5)
    model(dbatch).sum().backward()
    opt.step()
    with torch.no_grad():
        b = model(dbatch)
    ref = O.forward({k: v.detach().cpu() for k, v in model.state_dict().items()}, batch)
    assert (a - b).abs().max().item() > 1e-4
    assert (b.cpu() - ref).abs().max().item() <= 2e-5


@pytest.mark.parametrize("precision", ["bf16x3", "f32"])
def test_scoring_call_is_graph_capturable(precision):
    """The library allocates nothing and never synchronises: after one eager call (weights folded, kernel attributes
    set) a scoring call captures into a hipGraph and the replay reproduces the eager scores bit for bit."""
    cfg = wikimel_config(max_entity_attr_token_len=8, max_mention_sentence_len=16, resnet_num_region=4)
    model = Model(cfg, precision=precision).to(DEV).eval()
    model.load_state_dict(synth.make_state_dict(cfg, 7))
    batch = _to_dev(synth.make_batch(cfg, 6, 13)[:14])
    with torch.no_grad():
        eager = model(batch).clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            model(batch)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            captured = model(batch)
        for t in batch:                      # new inputs through the same static tensors
            if t.is_floating_point():
                t.mul_(1.0)
        graph.replay()
        torch.cuda.synchronize()
    assert torch.equal(captured, eager)


@pytest.mark.parametrize("name", ["tiny_wd", "tiny_wm"])
def test_c_abi_caller_without_torch(tmp_path, name):
    """examples/score_c_abi.cpp - hipMalloc'ed buffers and the C ABI only, no torch anywhere - scores the same case
    bit-identically to the Module on both forward paths: the boundary really is plain pointers and sizes."""
    import shutil
    import struct
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not on this box")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "score_c_abi")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", f"-I{repo}/include", f"{repo}/examples/score_c_abi.cpp",
                    f"-L{repo}/drin_amd", "-ldrin_hip", f"-Wl,-rpath,{repo}/drin_amd", "-o", exe], check=True, capture_output=True)
    cfg, sd, batch = build_case(name)
    B, N, D, R = batch[0].shape[0], cfg.num_candidates_model, cfg.bert_embed_dim, cfg.resnet_embed_dim
    T = cfg.max_entity_attr_token_len if cfg.token_level_entities else 0
    blob = struct.pack("12i", B, N, D, R, cfg.max_mention_sentence_len, cfg.resnet_num_region, cfg.object_topk_mention,
                       cfg.object_topk_entity, T, int(cfg.gcn_edge_type == "dynamic"), cfg.num_gcn_layers, 1)
    order = [0, 2, 3, 4, 5, 6, 7] + ([8] if T else []) + [9, 10, 11, 12, 13]      # drin_batch order (mask only when token-level)
    for i in order:
        blob += batch[i].contiguous().numpy().tobytes()
    for k in synth.STATE_DICT_SHAPES(D, R, cfg.num_gcn_layers):
        blob += sd[k[0]].contiguous().numpy().tobytes()
    answer = batch[14].to(torch.uint8).contiguous()
    blob += answer.numpy().tobytes()
    case, out = str(tmp_path / "case.bin"), str(tmp_path / "scores.bin")
    open(case, "wb").write(blob)
    dbatch = _to_dev(batch[:14])
    from drin_amd.metrics import DeviceLossMetric
    for prec_arg, prec in (("f32", "f32"), ("bf16x3", "bf16x3_all")):
        r = subprocess.run([exe, case, out, prec_arg], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        flat = torch.from_numpy(np.fromfile(out, dtype=np.float32))
        got = flat[:2 * B * N].reshape(2, B, N)
        with torch.no_grad():
            layerwise = Model(cfg, precision=prec, fused=False).to(DEV).eval()
            layerwise.load_state_dict(sd)
            folded = Model(cfg, precision=prec).to(DEV).eval()
            folded.load_state_dict(sd)
            assert torch.equal(got[0], layerwise(dbatch).cpu()), prec
            assert torch.equal(got[1], folded(dbatch).cpu()), prec
        # the training step's device work through the same three entry points the Module's autograd edge uses
        layerwise.train()
        metric = DeviceLossMetric(cfg.triplet_margin, (1,), DEV)
        loss = metric(answer.to(DEV), layerwise(dbatch))
        loss.backward()
        rest = flat[2 * B * N:]
        assert rest[0].item() == loss.item(), prec
        assert int(rest[1].item()) == int(metric.correct[0].item())
        at = 2
        for name, shape in synth.STATE_DICT_SHAPES(D, R, cfg.num_gcn_layers):
            n = int(np.prod(shape))
            g_c = rest[at:at + n].reshape(shape)
            at += n
            g_t = dict(layerwise.named_parameters())[name].grad
            if g_t is None:
                assert not g_c.any(), name        # dead parameter: the C caller passed no buffer, its arena stayed zero
            else:
                assert torch.allclose(g_c, g_t.cpu(), rtol=1e-4, atol=1e-7), (prec, name)
        # (4) the optimiser step through drin_adam_step: the C caller's updated parameters == torch.optim.Adam's first step on
        #     the C caller's own gradients, bit for bit; parameters without a gradient untouched
        grads_c, at_g = {}, 2
        for name, shape in synth.STATE_DICT_SHAPES(D, R, cfg.num_gcn_layers):
            n = int(np.prod(shape))
            grads_c[name] = rest[at_g:at_g + n].reshape(shape)
            at_g += n
        twin = {k: torch.nn.Parameter(v.clone().to(DEV)) for k, v in sd.items()}
        for k, q in twin.items():
            q.grad = None if dict(layerwise.named_parameters())[k].grad is None else grads_c[k].to(DEV)
        torch.optim.Adam(list(twin.values()), lr=1e-3).step()
        for name, shape in synth.STATE_DICT_SHAPES(D, R, cfg.num_gcn_layers):
            n = int(np.prod(shape))
            p_c = rest[at:at + n].reshape(shape)
            at += n
            assert torch.equal(p_c, twin[name].detach().cpu()), (prec, name)
        assert at == rest.numel()


def test_refuses_cpu_tensors():
    cfg = DrinConfig(**TINY)
    m = Model(cfg)
    with pytest.raises(RuntimeError, match="AMD GPU only"):
        m(synth.make_batch(cfg, 2, 1))


def test_determinism():
    cfg, sd, batch = build_case("wd_b4")
    model = _model(cfg, sd)
    with torch.no_grad():
        a = model(_to_dev(batch))
        b = model(_to_dev(batch))
    assert torch.equal(a, b)


# ---- backward (loss.backward() of train.py:33-34) ---------------------------------------------------------
def _grads_of(model, loss):
    model.zero_grad()
    loss.backward()
    return {k: (p.grad.detach().cpu() if p.grad is not None else None) for k, p in model.named_parameters()}


@pytest.mark.parametrize("name", [n for n in CASES if CASES[n][5]])
def test_backward_matches_reference_golden(golden_dir, name):
    """Gradient of a fixed linear functional of the scores, and of the triplet loss, against the
    reference's autograd (golden) - including which parameters get no gradient at all."""
    from drin_amd.metrics import TripletLoss
    cfg, sd, batch = build_case(name)
    g = _golden(golden_dir, name)
    model = Model(cfg, precision="f32").to(DEV)
    model.load_state_dict(sd)
    dbatch = _to_dev(batch)
    scores = model(dbatch[:-1])
    rng = np.random.Generator(np.random.Philox(key=[int(g["functional_weights_seed"]), 99]))
    w = torch.from_numpy(rng.standard_normal(size=tuple(scores.shape), dtype=np.float32)).to(DEV)
    grads = _grads_of(model, (scores * w).sum())
    assert sorted(k for k, v in grads.items() if v is None) == sorted(g["grad_none"].tolist())
    full = CASES[name][4]
    worst = 0.0
    for k, gr in grads.items():
        if gr is None:
            continue
        l2 = float(g[f"lin_grad_l2/{k}"])
        rel = abs(gr.double().norm().item() - l2) / (l2 + 1e-12)
        worst = max(worst, rel)
        assert rel <= 2e-4, (k, rel)
        ref = g[f"lin_grad/{k}"]
        got = gr.numpy() if full else gr.flatten()[:16].numpy()
        np.testing.assert_allclose(got, ref, atol=3e-4 * l2 / np.sqrt(gr.numel()) + 1e-7, rtol=2e-3, err_msg=k)
    print(f"{name}: worst relative grad-norm error {worst:.2e}")
    loss = TripletLoss(cfg.triplet_margin)(dbatch[-1], model(dbatch[:-1]))
    assert abs(loss.item() - float(g["triplet_loss"])) <= 1e-5
    grads = _grads_of(model, loss)
    for k, gr in grads.items():
        if gr is not None:
            l2 = float(g[f"loss_grad_l2/{k}"])
            assert abs(gr.double().norm().item() - l2) <= 1e-3 * l2 + 1e-9, k
    # one Adam step (train.py:55-56) lands on the reference's weights
    opt = torch.optim.Adam(model.parameters(), lr=cfg.learning_rate)
    opt.step()
    for k, p in model.named_parameters():
        np.testing.assert_allclose(p.detach().flatten()[:8].cpu().numpy(), g[f"adam_head/{k}"], atol=2e-5, rtol=1e-4, err_msg=k)


def test_backward_vs_oracle_autograd_reference_batch():
    """WikiDiverse reference batch (B=64): every parameter gradient against autograd through the oracle."""
    from drin_amd.metrics import TripletLoss
    cfg = DrinConfig()
    sd = synth.make_state_dict(cfg, 7)
    batch = synth.make_batch(cfg, 64, 43)
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref_loss = O.triplet_loss(batch[-1], O.forward(p, batch), cfg.triplet_margin)
    ref = torch.autograd.grad(ref_loss, list(p.values()), allow_unused=True)
    model = Model(cfg, precision="f32").to(DEV)
    model.load_state_dict(sd)
    dbatch = _to_dev(batch)
    loss = TripletLoss(cfg.triplet_margin)(dbatch[-1], model(dbatch[:-1]))
    assert abs(loss.item() - ref_loss.item()) <= 1e-5
    grads = _grads_of(model, loss)
    for (k, got), r in zip(grads.items(), ref):
        assert (got is None) == (r is None), k
        if r is None:
            continue
        denom = r.norm().item() + 1e-12
        rel = (got - r).norm().item() / denom
        assert rel <= 2e-4, (k, rel)


@pytest.mark.parametrize("maker,B,edge_type", [(DrinConfig, 16, "dynamic"), (wikimel_config, 3, "dynamic"), (DrinConfig, 5, "static")])
def test_vector_edges_vs_oracle(maker, B, edge_type):
    """gcn_edge_feature="vector" (model.py:112-116,136-149) at the reference's widths: scores and every
    parameter gradient against autograd through the oracle; the Module never takes the folded path."""
    from drin_amd.metrics import TripletLoss
    cfg = maker(gcn_edge_feature="vector", gcn_edge_type=edge_type)
    sd = synth.make_state_dict(cfg, 17)
    batch = synth.make_batch(cfg, B, 47)
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref_scores = O.forward(p, batch, dynamic=edge_type == "dynamic", vector=True)
    ref_loss = O.triplet_loss(batch[-1], ref_scores, cfg.triplet_margin)
    ref = torch.autograd.grad(ref_loss, list(p.values()), allow_unused=True)
    model = Model(cfg, precision="f32").to(DEV)
    model.load_state_dict(sd)
    dbatch = _to_dev(batch)
    with torch.no_grad():
        _lib.profile_begin()
        s = model(dbatch[:-1])
        prof = _lib.profile_end()
    assert prof["stream"][1] == 0, "vector edges are not folded: the layer-by-layer path must run"
    err = (s.cpu() - ref_scores.detach()).abs().max().item()
    print(f"vector {cfg.dataset_name} {edge_type}: max |score - oracle| = {err:.3e}")
    assert err <= 2e-5
    loss = TripletLoss(cfg.triplet_margin)(dbatch[-1], model(dbatch[:-1]))
    assert abs(loss.item() - ref_loss.item()) <= 1e-5
    grads = _grads_of(model, loss)
    for (k, got), r in zip(grads.items(), ref):
        assert (got is None) == (r is None), k
        if r is not None:
            rel = (got - r).norm().item() / (r.norm().item() + 1e-12)
            assert rel <= 2e-4, (k, rel)


def test_backward_bf16x3_vs_oracle_autograd():
    """Training in split-bf16 precision (forward, dX and dW contractions on the bf16 matrix cores): every
    parameter gradient of a WikiMEL-shaped batch against autograd through the fp32 oracle."""
    from drin_amd.metrics import TripletLoss
    cfg = wikimel_config(max_entity_attr_token_len=8, max_mention_sentence_len=16, resnet_num_region=4)
    sd = synth.make_state_dict(cfg, 7)
    batch = synth.make_batch(cfg, 16, 44)
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref_loss = O.triplet_loss(batch[-1], O.forward(p, batch), cfg.triplet_margin)
    ref = torch.autograd.grad(ref_loss, list(p.values()), allow_unused=True)
    model = Model(cfg, precision="bf16x3").to(DEV)
    model.load_state_dict(sd)
    dbatch = _to_dev(batch)
    _lib.profile_begin()
    loss = TripletLoss(cfg.triplet_margin)(dbatch[-1], model(dbatch[:-1]))
    grads = _grads_of(model, loss)
    prof = _lib.profile_end()
    assert prof["gemm_x3"][1] >= 9, "forward and backward pair-sized contractions should run split-bf16 (the weight gradients as one group)"
    assert abs(loss.item() - ref_loss.item()) <= 1e-5
    worst = 0.0
    for (k, got), r in zip(grads.items(), ref):
        assert (got is None) == (r is None), k
        if r is not None:
            rel = (got - r).norm().item() / (r.norm().item() + 1e-12)
            worst = max(worst, rel)
            assert rel <= 2e-4, (k, rel)
    print(f"bf16x3 training: worst relative gradient error {worst:.2e}")


def test_training_loop_on_device_tracks_cpu_oracle_loop(tmp_path):
    """config 1/3 plumbing on the GPU: .npy -> loader -> HIP forward/backward -> Adam, against the same
    loop driven by the CPU oracle (same seed, same data): per-epoch losses and top-k agree."""
    from drin_amd.data import create_datasets, write_synthetic_dataset
    from drin_amd.train import MELRunner, seed_everything
    from tests.helpers import OracleModel
    cfg = DrinConfig(batch_size=8, num_epoch=2, test_epoch_interval=1, shuffle_train_data=False, **TINY)
    write_synthetic_dataset(cfg, str(tmp_path), sizes=(32, 8, 8), seed=2)
    hist = {}
    for kind in ("hip", "oracle"):
        seed_everything(cfg.seed)
        loaders = create_datasets(cfg, str(tmp_path))
        model = Model(cfg, precision="f32").to(DEV) if kind == "hip" else OracleModel(cfg)
        hist[kind] = MELRunner(cfg, model, DEV if kind == "hip" else "cpu").fit(loaders)
    for a, b in zip(hist["hip"].train + hist["hip"].valid + hist["hip"].test,
                    hist["oracle"].train + hist["oracle"].valid + hist["oracle"].test):
        # Adam divides by sqrt(v): last-bit gradient differences on near-zero entries are amplified step
        # after step, so the two trajectories agree to ~1e-4 after 2 epochs, not to fp32 rounding
        assert abs(a.loss - b.loss) <= 5e-4, (a, b)
        assert a.topk == pytest.approx(b.topk, abs=1e-9)
    assert hist["hip"].train[-1].loss < hist["hip"].train[0].loss


# ---- table form: on-device entity gather (drin/data.py:87-93) -----------------------------------
@pytest.mark.parametrize("token_level", [True, False], ids=["wikimel_tokens", "pooled_text"])
def test_indexed_batch_matches_gathered_batch(token_level):
    """Candidate indices into device-resident entity tables score exactly like the gathered 14-sequence: the
    fused bf16x3 path gathers inside the stream kernel (bit-identical), every other mode gathers with torch."""
    from drin_amd.model import EntityTable, IndexedBatch
    cfg = DrinConfig(dataset_name="wikimel" if token_level else "wikidiverse", num_candidates_data=20,
                     max_entity_attr_token_len=10, **TINY)
    sd = synth.make_state_dict(cfg, 8)
    E, B, N = 57, 6, cfg.num_candidates_model
    tab = synth.make_batch(cfg.with_(num_candidates_data=E - 1), 1, 71)
    table = EntityTable(tab[7][0], tab[8][0] if token_level else None, tab[9][0], tab[10][0], tab[11][0]).to(DEV)
    men = _to_dev(synth.make_batch(cfg, B, 72))
    g = torch.Generator().manual_seed(3)
    cand = torch.randint(0, E, (B, N), generator=g).to(DEV)
    ib = IndexedBatch(men[:7], table, cand, men[12], men[13])
    gathered = ib.gathered()
    ref = O.forward(sd, [t.cpu() for t in gathered])
    for precision in ("bf16x3_all", "f32"):
        model = Model(cfg, precision=precision).to(DEV).eval()
        model.load_state_dict(sd)
        with torch.no_grad():
            a = model(ib)
            b = model(gathered)
        assert torch.equal(a, b), precision
        assert (a.cpu() - ref).abs().max().item() <= 1e-5
    # training through the table form: same gradients as through the gathered tensors
    model = Model(cfg, precision="f32").to(DEV)
    model.load_state_dict(sd)
    s1 = model(ib)
    s1.sum().backward()
    g1 = [p.grad.clone() if p.grad is not None else None for p in model.parameters()]
    model.zero_grad()
    s2 = model(gathered)
    s2.sum().backward()
    assert torch.equal(s1, s2)
    for x, p in zip(g1, model.parameters()):
        assert (x is None) == (p.grad is None)
        if x is not None:
            assert torch.allclose(x, p.grad, rtol=1e-4, atol=1e-6)
    if token_level:
        # the training step took the pooled-ahead form: every entity's token mean computed once, by the library
        pooled, cls = table.pooled_text(cfg)
        assert table._pooled is not None and pooled.shape == (E, cfg.bert_embed_dim)
        ref_pool = O.entity_token_mean(table.text.cpu().unsqueeze(0), table.mask.cpu().unsqueeze(0))[0]
        np.testing.assert_allclose(pooled.cpu().numpy(), ref_pool.numpy(), atol=1e-6)
        assert torch.equal(cls, table.text[:, 0, :])
        # and the C ABI refuses the combinations it does not implement
        from drin_amd.model import _Call
        seq, cls_rows = ib.gathered_pooled(cfg)
        call = _Call(cfg, seq, _lib.PREC_BF16X3, entity_text_cls=cls_rows)
        lib = _lib.load()
        st = torch.cuda.current_stream().cuda_stream
        edges = torch.empty(4, B, N, device=DEV)
        span = torch.empty(B, cfg.bert_embed_dim, device=DEV)
        _lib.check(lib.drin_edges_fwd(C.byref(call.cfg), C.byref(call.batch), edges.data_ptr(), span.data_ptr(), st))
        full = _Call(cfg, gathered, _lib.PREC_BF16X3)
        edges2 = torch.empty_like(edges)
        _lib.check(lib.drin_edges_fwd(C.byref(full.cfg), C.byref(full.batch), edges2.data_ptr(), span.data_ptr(), st))
        assert torch.equal(edges, edges2)
        full.batch.entity_text_cls = cls_rows.data_ptr()               # token block + cls rows: not a form of the ABI
        assert lib.drin_edges_fwd(C.byref(full.cfg), C.byref(full.batch), edges2.data_ptr(), span.data_ptr(), st) == _lib.E_UNSUPPORTED
        assert lib.drin_fused_supported(C.byref(call.cfg)) == _lib.OK
        ws = torch.empty(max(lib.drin_fused_workspace_bytes(C.byref(call.cfg)), 16), dtype=torch.uint8, device=DEV)
        sc = torch.empty(B, N, device=DEV)
        rc = lib.drin_forward_prepared(C.byref(call.cfg), C.byref(call.batch), None, None, ws.data_ptr(), ws.numel(), sc.data_ptr(), st)
        assert rc in (_lib.E_UNSUPPORTED, _lib.E_NULL)


def test_indexed_loader_and_runner(tmp_path):
    """.npy tables -> device EntityTable + index batches -> MELRunner: same losses as the gathered loader."""
    from drin_amd.data import create_datasets, create_indexed_datasets, load_entity_table, write_synthetic_dataset
    from drin_amd.train import MELRunner, seed_everything
    cfg = DrinConfig(dataset_name="wikimel", num_candidates_data=6, max_entity_attr_token_len=6, batch_size=4, num_epoch=1,
                     test_epoch_interval=1, shuffle_train_data=False, metrics_topk=(1, 3), acc_correction=(0.0, 0.0, 0.0), **TINY)
    write_synthetic_dataset(cfg, str(tmp_path), sizes=(12, 4, 4), seed=4, num_entities=30)
    from drin_amd.data import create_device_splits
    hist = {}
    for kind in ("gathered", "indexed", "device"):
        seed_everything(cfg.seed)
        model = Model(cfg, precision="f32").to(DEV)
        if kind == "gathered":
            hist[kind] = MELRunner(cfg, model, DEV).fit(create_datasets(cfg, str(tmp_path)))
        else:
            table = load_entity_table(cfg, str(tmp_path), DEV)
            # "device": the whole split resident on the GPU, batches are slices - no host work in the step
            loaders = create_indexed_datasets(cfg, str(tmp_path)) if kind == "indexed" else create_device_splits(cfg, str(tmp_path), DEV)
            hist[kind] = MELRunner(cfg, model, DEV, entity_table=table).fit(loaders)
    for a, b in zip(hist["gathered"].train + hist["gathered"].test, hist["indexed"].train + hist["indexed"].test):
        assert abs(a.loss - b.loss) <= 1e-4 and a.topk == pytest.approx(b.topk, abs=1e-9)
    for a, b in zip(hist["indexed"].train + hist["indexed"].test, hist["device"].train + hist["device"].test):
        # the same batches through the same kernels (weight-gradient atomics of the small products may move last bits)
        assert abs(a.loss - b.loss) <= 1e-6 and a.topk == pytest.approx(b.topk, abs=1e-9)


# ---- ragged / unusual geometries against the oracle ---------------------------------------------------------
@pytest.mark.parametrize("kw,B", [
    (dict(object_topk_mention=2, object_topk_entity=2), 3),            # Km, Ke other than (3, 1): i-major / j-minor sum
    (dict(num_candidates_data=0), 4),                                   # N = 1: only the answer slot
    (dict(num_candidates_data=1000), 2),                                # BASELINE config 5 candidate count
    (dict(dataset_name="wikimel", num_candidates_data=5, max_entity_attr_token_len=1), 3),   # T = 1: every slice empty -> NaN
    (dict(dataset_name="wikimel", num_candidates_data=17, max_entity_attr_token_len=70), 2),  # T > 64: mask count over 2 wave passes
    (dict(max_mention_sentence_len=4, resnet_num_region=1), 5),
])
@pytest.mark.parametrize("fused", [True, False], ids=["fused", "layerwise"])
def test_unusual_geometries_vs_oracle(kw, B, fused):
    base = dict(TINY)
    base.update(kw)
    cfg = DrinConfig(**base)
    sd = synth.make_state_dict(cfg, 8)
    batch = synth.make_batch(cfg, B, 61, min_tokens=1)
    ref = O.forward(sd, batch)
    model = Model(cfg, precision="f32", fused=fused).to(DEV).eval()
    model.load_state_dict(sd)
    with torch.no_grad():
        got = model(_to_dev(batch)).cpu()
    assert got.shape == ref.shape
    assert torch.equal(torch.isnan(got), torch.isnan(ref))
    assert (torch.nan_to_num(got) - torch.nan_to_num(ref)).abs().max().item() <= 2e-5


def _random_geometry(seed):
    g = np.random.Generator(np.random.Philox(key=[seed, 4242]))
    pick = lambda xs: xs[int(g.integers(0, len(xs)))]  # noqa: E731
    wm = bool(g.integers(0, 2))
    D = pick([32, 64, 96, 128, 192])
    kw = dict(dataset_name="wikimel" if wm else "wikidiverse", bert_embed_dim=D, gcn_embed_dim=D, resnet_embed_dim=pick([32, 64, 128, 160, 256]),
              num_candidates_data=int(g.integers(1, 41)), max_entity_attr_token_len=int(g.integers(3, 13)),
              max_mention_sentence_len=int(g.integers(6, 20)), resnet_num_region=int(g.integers(1, 8)),
              object_topk_mention=int(g.integers(1, 5)), object_topk_entity=int(g.integers(1, 3)),
              num_gcn_layers=pick([1, 2, 2, 2, 3]), gcn_edge_type=pick(["dynamic", "dynamic", "static"]),
              gcn_edge_feature=pick(["scaler", "scaler", "scaler", "vector"]),
              gcn_edge_enabled=tuple(float(x) for x in pick([(1, 1, 1, 1), (1, 1, 1, 1), (1, 0, 1, 1), (0, 1, 1, 0), (1, 1, 0, 1)])))
    B = int(g.integers(1, 10))
    # activations by name (args.py:35-36): drawn last, so that the geometries of the earlier seeds stay what they were
    kw["gcn_vertex_activation"] = pick(["gelu", "gelu", "gelu", "relu", "tanh", "silu", "sigmoid"])
    kw["gcn_edge_activation"] = pick(["sigmoid", "sigmoid", "sigmoid", "tanh", "relu"])
    if seed >= 24:   # round 3: the edge activations whose backward reads the kept pre-activation (drawn for the new seeds only)
        kw["gcn_edge_activation"] = pick(["gelu", "silu", "gelu", "silu", "sigmoid"])
    return DrinConfig(**kw), B


@pytest.mark.parametrize("seed", range(32))
def test_random_geometries_all_paths_vs_oracle(seed):
    """Seeded sweep over widths / counts / switches: the folded path (both precisions), the layer-by-layer path and the
    backward against the oracle - the corner cases no hand-written list anticipates (widths that are not multiples of
    32, one candidate, one region, masks, 1 and 3 layers fall back to the right kernels)."""
    from drin_amd.metrics import TripletLoss
    cfg, B = _random_geometry(seed)
    sd = synth.make_state_dict(cfg, 100 + seed)
    batch = synth.make_batch(cfg, B, 200 + seed, min_tokens=3)
    kw = O.config_kwargs(cfg)
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.forward(p, batch, **kw)
    ref_loss = O.triplet_loss(batch[-1], ref, cfg.triplet_margin)
    ref_grads = torch.autograd.grad(ref_loss, list(p.values()), allow_unused=True)
    ref = ref.detach()
    dbatch = _to_dev(batch)
    for precision, fused in (("f32", True), ("bf16x3_all", True), ("f32", False)):
        model = Model(cfg, precision=precision, fused=fused).to(DEV).eval()
        model.load_state_dict(sd)
        with torch.no_grad():
            got = model(dbatch[:-1]).cpu()
        assert torch.equal(torch.isnan(got), torch.isnan(ref)), (precision, fused)
        err = (torch.nan_to_num(got) - torch.nan_to_num(ref)).abs().max().item()
        assert err <= 2e-5, (seed, precision, fused, err, cfg)
    if not torch.isnan(ref).any():
        model = Model(cfg, precision="bf16x3").to(DEV)
        model.load_state_dict(sd)
        loss = TripletLoss(cfg.triplet_margin)(dbatch[-1], model(dbatch[:-1]))
        assert abs(loss.item() - ref_loss.item()) <= 2e-5
        grads = _grads_of(model, loss)
        for (k, got), r in zip(grads.items(), ref_grads):
            assert (got is None) == (r is None), (seed, k)
            if r is not None and r.norm().item() > 1e-8:
                rel = (got - r).norm().item() / r.norm().item()
                assert rel <= 2e-4, (seed, k, rel)


@pytest.mark.parametrize("seed", range(10))
def test_random_table_forms(seed):
    """Seeded sweep of the table form: in-kernel gather == gathered batch (bit-exact), the per-entity cache and the
    bf16-stored tables within fp32 re-association of the oracle on the gathered (widened) tensors."""
    from drin_amd.model import EntityTable, IndexedBatch
    g = np.random.Generator(np.random.Philox(key=[seed, 777]))
    pick = lambda xs: xs[int(g.integers(0, len(xs)))]  # noqa: E731
    wm = bool(g.integers(0, 2))
    D = pick([32, 64, 128])
    cfg = DrinConfig(dataset_name="wikimel" if wm else "wikidiverse", bert_embed_dim=D, gcn_embed_dim=D,
                     resnet_embed_dim=pick([32, 64, 128, 256]), num_candidates_data=int(g.integers(1, 45)),
                     max_entity_attr_token_len=int(g.integers(3, 11)), max_mention_sentence_len=12, resnet_num_region=3,
                     object_topk_mention=int(g.integers(1, 4)), object_topk_entity=int(g.integers(1, 3)),
                     gcn_edge_type=pick(["dynamic", "dynamic", "static"]))
    sd = synth.make_state_dict(cfg, 300 + seed)
    E, B, N = int(g.integers(5, 90)), int(g.integers(1, 8)), cfg.num_candidates_model
    tab = synth.make_batch(cfg.with_(num_candidates_data=E - 1), 1, 400 + seed)
    men = synth.make_batch(cfg, B, 500 + seed)
    cand = torch.from_numpy(g.integers(0, E, size=(B, N)))
    dyn = cfg.gcn_edge_type == "dynamic"
    for stored in (torch.float32, torch.bfloat16):
        feat = lambda t: t.to(stored)  # noqa: E731
        table = EntityTable(feat(tab[7][0]), tab[8][0] if wm else None, feat(tab[9][0]), feat(tab[10][0]), tab[11][0]).to(DEV)
        mention = [feat(t) if i in (0, 4, 5) else t for i, t in enumerate(men[:7])]
        ib = IndexedBatch(_to_dev(mention), table, cand.to(DEV), men[12].to(DEV), men[13].to(DEV))
        gathered = ib.gathered()
        ref = O.forward(sd, [t.float().cpu() if t.dtype == torch.bfloat16 else t.cpu() for t in gathered], dynamic=dyn)
        model = Model(cfg, precision="bf16x3_all").to(DEV).eval()
        model.load_state_dict(sd)
        with torch.no_grad():
            a, b = model(ib), model(gathered)
            assert torch.equal(a, b), (seed, stored)
            assert (a.cpu() - ref).abs().max().item() <= 2e-5
            if stored == torch.float32:
                table.enable_cache()
                c = model(ib)
                assert (c - a).abs().max().item() <= 6e-6 and (c.cpu() - ref).abs().max().item() <= 2e-5, (seed, cfg)
                table.enable_cache(False)


def test_empty_batch():
    cfg = DrinConfig(**TINY)
    model = Model(cfg, precision="f32").to(DEV).eval()
    batch = [t[:0] for t in _to_dev(synth.make_batch(cfg, 2, 1))]
    with torch.no_grad():
        out = model(batch)
    assert tuple(out.shape) == (0, cfg.num_candidates_model)


def test_inner_feature_dims_take_the_pooled_path():
    """mention objects [B, Km, 2, R] / entity image [B, N, 3, R] / entity objects [B, N, Ke, 2, R]: the means of
    model.py:43-44,78-83 are real reductions here (both datasets store a singleton there)."""
    cfg = DrinConfig(**TINY)
    sd = synth.make_state_dict(cfg, 8)
    batch = synth.make_batch(cfg, 3, 62)
    g = torch.Generator().manual_seed(5)
    batch[5] = torch.randn(3, 3, 2, cfg.resnet_embed_dim, generator=g)
    batch[9] = torch.randn(3, cfg.num_candidates_model, 3, cfg.resnet_embed_dim, generator=g)
    batch[10] = torch.randn(3, cfg.num_candidates_model, 1, 2, cfg.resnet_embed_dim, generator=g)
    ref = O.forward(sd, batch)
    model = Model(cfg, precision="f32").to(DEV).eval()
    model.load_state_dict(sd)
    with torch.no_grad():
        got = model(_to_dev(batch)).cpu()
    assert (got - ref).abs().max().item() <= 2e-5


# ---- modes outside the path's tolerance are gone (ABI 6) ------------------------------------------------------
def test_precisions_outside_the_bar_are_not_offered():
    """`precision="bf16"` (every pair-sized contraction in one bf16 pass: 5e-4 on the scores) and `"bf16x3_i1"` (the image
    contraction in one bf16 pass: 1.6e-4 on trained weights) were outside the path's 1e-4 bar and were removed in round 5: the
    module refuses their names, the library their enum values (2, 4) at every entry point."""
    cfg = DrinConfig()
    for name in ("bf16", "bf16x3_i1"):
        with pytest.raises(ValueError, match="precision"):
            Model(cfg, precision=name)
    lib = _lib.load()
    c = _lib.DrinConfigC()
    lib.drin_default_config(C.byref(c))
    c.batch = 4
    for value in (2, 4, 6, -1):
        c.precision = value
        assert lib.drin_fused_supported(C.byref(c)) == _lib.E_UNSUPPORTED and b"precision" in lib.drin_last_error()
        assert lib.drin_workspace_bytes(C.byref(c), 0) == 0
    # the one precision-by-contraction mode left is a mode of the fused inference entry point only
    c.precision = _lib.PREC_BF16X3_IF16
    assert lib.drin_fused_supported(C.byref(c)) == _lib.OK
    bt = _lib.DrinBatchC()
    assert lib.drin_forward(C.byref(c), C.byref(bt), None, None, 0, None, 0, None, None) in (_lib.E_UNSUPPORTED, _lib.E_NULL)


# ---- features stored as bf16 (BASELINE configs 2-3) -----------------------------------------------------------
FEATURE_SLOTS = (0, 4, 5, 7, 9, 10)   # mention text / image / object, entity text / image / object


def _bf16_features(batch):
    return [t.to(torch.bfloat16) if i in FEATURE_SLOTS else t for i, t in enumerate(batch)]


@pytest.mark.parametrize("kw,B", [
    (dict(dataset_name="wikimel", num_candidates_data=100, max_entity_attr_token_len=8, max_mention_sentence_len=16,
          resnet_num_region=4), 5),
    (dict(max_mention_sentence_len=16, resnet_num_region=4), 40),
    (dict(dataset_name="wikimel", num_candidates_data=20, max_entity_attr_token_len=10, **TINY), 6),
    (dict(num_candidates_data=7, gcn_edge_type="static", **TINY), 9),
    # 4 .. 27 tokens per entity: the flat two-rows-per-three-loads walk of the bf16 token block at D = 768 with four pairs in
    # flight, single pairs, the unpaired last row, and none at all (round 4)
    (dict(dataset_name="wikimel", num_candidates_data=100, max_entity_attr_token_len=27, max_mention_sentence_len=16,
          resnet_num_region=4), 3),
], ids=["wikimel_dims", "wikidiverse_dims", "tiny_tokens", "tiny_pooled_static", "wikimel_dims_long_token_lists"])
def test_bf16_stored_features(kw, B):
    """Features stored as bf16 and read in place by the fused path (half the bytes of the HBM-bound pass): the scores
    equal the reference forward on the SAME features widened to fp32 (oracle, 1e-5) and the library's own fp32-feature
    run on the widened tensors (fp32 re-association only); every other path widens them first, exactly."""
    cfg = DrinConfig(**kw)
    sd = synth.make_state_dict(cfg, 9)
    stored = _bf16_features(synth.make_batch(cfg, B, 31)[:14])
    widened = [t.float() if t.dtype == torch.bfloat16 else t for t in stored]
    ref = O.forward(sd, widened, dynamic=cfg.gcn_edge_type == "dynamic")
    model = Model(cfg, precision="bf16x3_all").to(DEV).eval()
    model.load_state_dict(sd)
    with torch.no_grad():
        _lib.profile_begin()
        s_bf16 = model(_to_dev(stored))
        prof = _lib.profile_end()
        s_f32 = model(_to_dev(widened))
    assert prof["stream"][1] == 1, "the fused path reads the bf16 features in place"
    err = (s_bf16.cpu() - ref).abs().max().item()
    print(f"bf16-stored features: max |score - oracle(widened)| = {err:.3e}, vs fp32-feature run {(s_bf16 - s_f32).abs().max().item():.3e}")
    assert err <= 1e-5 and (s_bf16 - s_f32).abs().max().item() <= 4e-6
    # exact-fp32 precision and training widen the features (exactly): same bits as the widened tensors
    exact = Model(cfg, precision="f32").to(DEV).eval()
    exact.load_state_dict(sd)
    with torch.no_grad():
        assert torch.equal(exact(_to_dev(stored)), exact(_to_dev(widened)))
    model.train()
    model(_to_dev(stored)).sum().backward()
    g1 = [p.grad.clone() for p in model.parameters() if p.grad is not None]
    model.zero_grad()
    model(_to_dev(widened)).sum().backward()
    g2 = [p.grad for p in model.parameters() if p.grad is not None]
    assert len(g1) == len(g2) and all((a - b).abs().max().item() <= 1e-6 * (b.abs().max().item() + 1e-12) + 1e-9 for a, b in zip(g1, g2))
    # all six or none
    mixed = list(stored)
    mixed[9] = mixed[9].float()
    with pytest.raises(ValueError, match="all six"):
        with torch.no_grad():
            model.eval()(_to_dev(mixed))


def test_bf16_stored_entity_table():
    """Table form with bf16 tables: gathered inside the stream kernel, equal to the gathered bf16 batch."""
    from drin_amd.model import EntityTable, IndexedBatch
    cfg = DrinConfig(dataset_name="wikimel", num_candidates_data=20, max_entity_attr_token_len=10, **TINY)
    sd = synth.make_state_dict(cfg, 8)
    E, B, N = 57, 6, cfg.num_candidates_model
    tab = synth.make_batch(cfg.with_(num_candidates_data=E - 1), 1, 71)
    bf = torch.bfloat16
    table = EntityTable(tab[7][0].to(bf), tab[8][0], tab[9][0].to(bf), tab[10][0].to(bf), tab[11][0]).to(DEV)
    men = _to_dev(_bf16_features(synth.make_batch(cfg, B, 72)))
    cand = torch.randint(0, E, (B, N), generator=torch.Generator().manual_seed(3)).to(DEV)
    ib = IndexedBatch(men[:7], table, cand, men[12], men[13])
    model = Model(cfg, precision="bf16x3_all").to(DEV).eval()
    model.load_state_dict(sd)
    with torch.no_grad():
        a, b = model(ib), model(ib.gathered())
    assert torch.equal(a, b)
    ref = O.forward(sd, [t.float().cpu() if t.dtype == bf else t.cpu() for t in ib.gathered()])
    assert (a.cpu() - ref).abs().max().item() <= 1e-5
    with pytest.raises(ValueError, match="fp32 tables"):
        table.enable_cache()
        with torch.no_grad():
            model(ib)
    table.enable_cache(False)
    # training from the bf16-stored table: tokens widened (exactly) and pooled once per entity - the same scores as a
    # step on the gathered, widened 14-sequence, bit for bit
    model.train()
    assert torch.equal(model(ib), model(ib.gathered()))
    assert table._pooled is not None and table._pooled[1].dtype == torch.float32


# ---- per-entity precompute cache ----------------------------------------------------------
@pytest.mark.parametrize("kw", [
    dict(dataset_name="wikimel", num_candidates_data=20, max_entity_attr_token_len=10, **TINY),
    dict(num_candidates_data=20, **TINY),
    dict(num_candidates_data=20, gcn_edge_type="static", **TINY),
    dict(dataset_name="wikimel", num_candidates_data=36, max_entity_attr_token_len=7, gcn_edge_enabled=(1, 0, 1, 1), **TINY),
    dict(dataset_name="wikimel", num_candidates_data=100, max_entity_attr_token_len=8, max_mention_sentence_len=16,
         resnet_num_region=4),
    dict(num_candidates_data=20, gcn_vertex_activation="silu", gcn_edge_activation="tanh", **TINY),
    dict(dataset_name="wikimel", num_candidates_data=100, max_entity_attr_token_len=8, max_mention_sentence_len=16,
         resnet_num_region=4, gcn_vertex_activation="relu", gcn_edge_activation="relu"),
], ids=["tiny_tokens", "tiny_pooled", "tiny_static", "tiny_mask", "wikimel_dims", "tiny_silu_tanh", "wikimel_dims_relu"])
def test_entity_cache_scores_match_oracle_and_uncached(kw):
    """Scoring from the per-entity cache == the oracle on the gathered 14-sequence (1e-5) == the un-cached
    table path (fp32 re-association only), for both precisions; a weight update rebuilds the cache."""
    from drin_amd.model import EntityTable, IndexedBatch
    cfg = DrinConfig(**kw)
    token_level = cfg.token_level_entities
    sd = synth.make_state_dict(cfg, 8)
    E, B, N = 83, 5, cfg.num_candidates_model
    tab = synth.make_batch(cfg.with_(num_candidates_data=E - 1), 1, 71)
    table = EntityTable(tab[7][0], tab[8][0] if token_level else None, tab[9][0], tab[10][0], tab[11][0]).to(DEV)
    men = _to_dev(synth.make_batch(cfg, B, 72))
    cand = torch.randint(0, E, (B, N), generator=torch.Generator().manual_seed(3)).to(DEV)
    ib = IndexedBatch(men[:7], table, cand, men[12], men[13])
    ref = O.forward(sd, [t.cpu() for t in ib.gathered()], dynamic=cfg.gcn_edge_type == "dynamic",
                    edge_enabled=cfg.gcn_edge_enabled, vertex_activation=cfg.gcn_vertex_activation,
                    edge_activation=cfg.gcn_edge_activation)
    for precision in ("bf16x3_all", "f32"):
        model = Model(cfg, precision=precision).to(DEV).eval()
        model.load_state_dict(sd)
        with torch.no_grad():
            table.enable_cache(False)
            plain = model(ib)
            table.enable_cache()
            _lib.profile_begin()
            cached = model(ib)
            prof = _lib.profile_end()
            again = model(ib)                                   # second call: cache reused, same bits
        err = (cached.cpu() - ref).abs().max().item()
        print(f"{precision}: cached max |score - oracle| = {err:.3e}, vs un-cached {(cached - plain).abs().max().item():.3e}")
        assert err <= 1e-5 and (cached - plain).abs().max().item() <= 4e-6
        assert torch.equal(cached, again)
        assert table._cache is not None and table._cache.numel() == E * (5 * cfg.gcn_embed_dim + cfg.resnet_embed_dim + 4) * 4
        # a weight update invalidates the cache (it is a function of the weights)
        key = table._cache_key
        with torch.no_grad():
            model.vertex_encoder.entity_image_linear.weight.mul_(1.5)
            moved = model(ib)
            table.enable_cache(False)
            moved_plain = model(ib)
            table.enable_cache()
        assert table._cache_key != key
        assert (moved - moved_plain).abs().max().item() <= 4e-6 and (moved - cached).abs().max().item() > 1e-4
    # training never takes the cache
    model = Model(cfg, precision="f32").to(DEV)
    model.load_state_dict(sd)
    model(ib).sum().backward()
    assert model.gcn_layers[0].w_h.weight.grad is not None
    table.enable_cache(False)


def test_entity_cache_out_of_range_index_and_nan_entity():
    """Out-of-range candidate indices clamp like the stream kernel's gather (and are reported: tests/test_gpu_round6.py); an entity whose token slice is
    empty (ntok <= 2, ghmfc.py:248) scores NaN for its pairs only."""
    from drin_amd.model import EntityTable, IndexedBatch
    cfg = DrinConfig(dataset_name="wikimel", num_candidates_data=12, max_entity_attr_token_len=6, **TINY)
    sd = synth.make_state_dict(cfg, 8)
    E, B, N = 30, 3, cfg.num_candidates_model
    tab = synth.make_batch(cfg.with_(num_candidates_data=E - 1), 1, 5)
    mask = tab[8][0].clone()
    mask[7] = 0
    mask[7, :2] = 1                                             # ntok = 2: mean of an empty slice
    table = EntityTable(tab[7][0], mask, tab[9][0], tab[10][0], tab[11][0]).to(DEV).enable_cache()
    men = _to_dev(synth.make_batch(cfg, B, 6))
    cand = torch.randint(0, E, (B, N), generator=torch.Generator().manual_seed(1))
    cand[cand == 7] = 8
    cand[0, 3] = 7
    cand[1, 0], cand[2, 5] = -4, E + 9
    model = Model(cfg, precision="bf16x3_all").to(DEV).eval()
    model.load_state_dict(sd)
    model.validate_indices = False                                  # (whatever DRIN_VALIDATE says: this test wants the clamped scores back)
    with torch.no_grad():
        s = model(IndexedBatch(men[:7], table, cand.to(DEV), men[12], men[13])).cpu()
        with pytest.raises(IndexError):                              # clamped for memory safety AND reported (round 6; data.py:87-93 raises)
            model.check_indices()
        clamped = cand.clamp(0, E - 1)
        table.enable_cache(False)
        s2 = model(IndexedBatch(men[:7], table, clamped.to(DEV), men[12], men[13])).cpu()
    assert torch.isnan(s[0]).all() and torch.isnan(s2[0]).all()   # the NaN vertex reaches the mention aggregates
    assert not torch.isnan(s[1:]).any()
    assert (s[1:] - s2[1:]).abs().max().item() <= 4e-6


# ---- caller-side loss + metric on the device ---------------------------------------------
def _loss_case(B, N, seed, ties=False):
    g = np.random.Generator(np.random.Philox(key=[seed, 21]))
    yhat = (g.random(size=(B, N), dtype=np.float32) * 2 - 1)
    if ties:
        yhat = np.round(yhat * 4) / 4                       # many equal scores: the `>=` tie rule matters
    onehot = np.concatenate([np.eye(N - 1, dtype=np.uint8), np.zeros((1, N - 1), dtype=np.uint8)], 0)
    y = onehot[g.integers(0, N, size=B)]
    return torch.from_numpy(yhat), torch.from_numpy(y)


def test_device_loss_matches_reference_triplet_golden(golden_dir):
    from drin_amd.metrics import DeviceLossMetric
    g = _golden(golden_dir, "triplet")
    for i in range(3):
        yhat, y = torch.from_numpy(g[f"yhat{i}"]).to(DEV), torch.from_numpy(g[f"y{i}"]).to(DEV)
        loss = DeviceLossMetric(0.25, (), DEV)(y, yhat)
        assert abs(loss.item() - float(g[f"loss{i}"])) <= 2e-7 * max(1.0, abs(float(g[f"loss{i}"]))), i


@pytest.mark.parametrize("B,N,ties", [(1, 2, False), (3, 4, True), (64, 11, False), (64, 101, False), (64, 101, True),
                                      (300, 37, True), (4500, 11, False)])
def test_device_loss_metric_vs_oracle(B, N, ties):
    """Loss, d loss / d scores and the top-k counters against the oracle's restatement of utils.py:26-73
    (pinned by the reference's own TripletLoss goldens and hand-computed TopkAccuracy cases)."""
    from drin_amd.metrics import DeviceLossMetric
    yhat, y = _loss_case(B, N, 100 + B + N, ties)
    ks = [k for k in (1, 3, 5, 10, 20, 50) if k <= N - 1]
    ref_in = yhat.clone().requires_grad_(True)
    ref = O.triplet_loss(y, ref_in, 0.25)
    ref.backward()
    m = DeviceLossMetric(0.25, ks, DEV)
    d_in = yhat.to(DEV).requires_grad_(True)
    loss = m(y.to(DEV), d_in)
    (loss * 3.0).backward()                                  # an upstream factor must scale the gradient
    assert abs(loss.item() - ref.item()) <= 3e-7 * max(1.0, abs(ref.item()))
    np.testing.assert_allclose(d_in.grad.cpu().numpy() / 3.0, ref_in.grad.numpy(), atol=2e-7 * ref_in.grad.abs().max().item() + 1e-12,
                               rtol=2e-6)
    assert torch.count_nonzero(d_in.grad[:, -1]).item() == 0
    m(y.to(DEV), yhat.to(DEV))                               # second batch: counters accumulate
    want = [2 * int(O.topk_counts(yhat, y, k)[0]) for k in ks]
    assert m.correct.tolist() == want and m.total == 2 * B
    m.reset()
    assert m.correct.tolist() == [0] * len(ks) and m.total == 0


def test_device_loss_nan_row_and_missing_gold():
    """A NaN score row (empty mention span, ghmfc.py:59) poisons the loss like torch.clamp does and is never
    counted; an all-zero answer row (data.py:159-161) contributes no hit."""
    from drin_amd.metrics import DeviceLossMetric, TopkAccuracy, TripletLoss
    yhat, y = _loss_case(6, 11, 5)
    y[2] = 0
    m = DeviceLossMetric(0.25, (1, 3), DEV)
    m(y.to(DEV), yhat.to(DEV))
    assert m.correct.tolist() == [int(O.topk_counts(yhat, y, k)[0]) for k in (1, 3)]
    yhat[4] = float("nan")
    m.reset()
    loss = m(y.to(DEV), yhat.to(DEV))
    assert torch.isnan(loss) and torch.isnan(TripletLoss(0.25)(y, yhat))
    for q, k in enumerate((1, 3)):
        t = TopkAccuracy(k)
        t.update(yhat, y)
        assert m.correct[q].item() == t.correct.item()


def test_end_to_end_training_learns_a_planted_signal(tmp_path):
    """Stand-in for "reproduces top-1 accuracy" (the datasets are not available offline): on a synthetic task whose gold
    candidate is planted in the text features, the full pipeline - loader, HIP forward / backward in split-bf16 precision,
    library loss + metric, Adam, the reference's round structure - raises held-out top-1 well above its initial value."""
    from drin_amd.data import create_datasets, write_synthetic_dataset
    from drin_amd.train import MELRunner, seed_everything
    cfg = DrinConfig(batch_size=32, metrics_topk=(1, 3), acc_correction=(0.0, 0.0, 0.0), shuffle_train_data=True,
                     learning_rate=1e-3, **TINY)
    write_synthetic_dataset(cfg, str(tmp_path), sizes=(512, 128, 128), seed=21, learnable=2.0)
    seed_everything(0)
    loaders = create_datasets(cfg, str(tmp_path), num_workers=0)
    model = Model(cfg, precision="bf16x3").to(DEV)
    runner = MELRunner(cfg, model, DEV)
    before = runner.run_epoch(loaders[2], 2, None)
    hist = runner.fit(loaders, num_epoch=8, test_epoch_interval=4)
    after = hist.test[-1]
    print(f"held-out top-1 {before.topk[0]:.3f} -> {after.topk[0]:.3f}, loss {before.loss:.4f} -> {after.loss:.4f}")
    # the randomly initialised model already sees the planted cosine through the text-text edge; training must sharpen it
    assert after.topk[0] >= before.topk[0] + 0.12 and after.topk[0] > 0.85 and after.loss < 0.5 * before.loss


def test_runner_with_device_loss_matches_torch_loss(tmp_path):
    """MELRunner with the library's loss/metric call reports the same history as with the torch classes."""
    from drin_amd.data import create_datasets, write_synthetic_dataset
    from drin_amd.train import MELRunner, seed_everything
    cfg = DrinConfig(batch_size=8, metrics_topk=(1, 3, 5), shuffle_train_data=False, **TINY)
    write_synthetic_dataset(cfg, str(tmp_path), sizes=(32, 16, 16), seed=11)
    hist = {}
    for fused in (True, False):
        seed_everything(0)
        loaders = create_datasets(cfg, str(tmp_path), num_workers=0)
        model = Model(cfg, precision="f32").to(DEV)
        runner = MELRunner(cfg, model, DEV, device_loss=fused)
        assert (runner.device_loss is not None) == fused
        hist[fused] = runner.fit(loaders, num_epoch=2, test_epoch_interval=2)
    for a, b in zip(hist[True].train + hist[True].valid + hist[True].test, hist[False].train + hist[False].valid + hist[False].test):
        assert abs(a.loss - b.loss) <= 2e-5 and a.topk == pytest.approx(b.topk, abs=1e-9)


def test_table_rows_beyond_2_31_elements():
    """A token-level entity table of 60 000 rows (11.8 GB: element offsets pass 2^31 from row 43 691 on) gathered inside
    the stream kernel, and pooled once per entity for training, scores exactly like the gathered 14-sequence."""
    from drin_amd.model import EntityTable, IndexedBatch
    cfg = DrinConfig(dataset_name="wikimel", num_candidates_data=100)
    E, B, N = 60_000, 3, cfg.num_candidates_model
    T, D, R = cfg.max_entity_attr_token_len, cfg.bert_embed_dim, cfg.resnet_embed_dim
    g = torch.Generator(device=DEV)
    g.manual_seed(5)
    text = torch.randn(E, T, D, device=DEV, generator=g)
    ntok = torch.randint(4, T + 1, (E,), device=DEV, generator=g)
    mask = (torch.arange(T, device=DEV)[None] < ntok[:, None]).to(torch.int64)
    table = EntityTable(text, mask, torch.randn(E, 1, R, device=DEV, generator=g), torch.randn(E, 1, 1, R, device=DEV, generator=g),
                        torch.rand(E, 1, device=DEV, generator=g))
    men = synth.make_device_batch(cfg, B, 9, DEV)
    cand = torch.randint(43_691, E, (B, N), device=DEV, generator=g)     # every candidate beyond the 32-bit offset range
    cand[0, :4] = torch.tensor([0, E - 1, 43_690, 43_691], device=DEV)
    ib = IndexedBatch(men[:7], table, cand, men[12], men[13])
    sd = synth.make_state_dict(cfg, 3)
    model = Model(cfg, precision="bf16x3").to(DEV)
    model.load_state_dict(sd)
    gathered = ib.gathered()
    with torch.no_grad():
        model.eval()
        assert torch.equal(model(ib), model(gathered))
    model.train()
    a = model(ib)                                                         # pooled once per entity
    b = model(gathered)
    assert torch.equal(a, b)
    del table, text
    torch.cuda.empty_cache()


def test_mid_size_batch_tail_split_paths_agree():
    """256 WikiMEL-shaped mentions = 25 856 pairs = 303 output tiles of 256 x 256: the 47 tiles of the partly filled last
    round split K over the idle CUs in both split-bf16 NT kernels.  The folded path (planes kernel, tail split) against
    the layer-by-layer path (its own kernels and tail split) on the same device batch, and a 64-mention slice of the
    batch scored on its own (whole-product split-K instead): all within fp32 re-association."""
    cfg = DrinConfig(dataset_name="wikimel", num_candidates_data=100)
    B = 256
    sd = synth.make_state_dict(cfg, 5)
    batch = synth.make_device_batch(cfg, B, 77, DEV)[:14]
    with torch.no_grad():
        folded = Model(cfg, precision="bf16x3").to(DEV).eval()
        folded.load_state_dict(sd)
        generic = Model(cfg, precision="bf16x3", fused=False).to(DEV).eval()
        generic.load_state_dict(sd)
        a = folded(batch)
        b = generic(batch)
        c = folded([t[:64] for t in batch])
    assert torch.isfinite(a).all()
    assert (a - b).abs().max().item() <= 1e-5
    assert (a[:64] - c).abs().max().item() <= 2e-6
    ref = O.forward(sd, [t[:4].cpu() for t in batch])
    assert (a[:4].cpu() - ref).abs().max().item() <= 1e-5


def test_training_from_bf16_stored_features_pools_in_place():
    """A training step on a batch whose features are stored as bf16: the token blocks are pooled in place by the library
    (`drin_pool_fwd`, bf16 tokens, fp32 sums) instead of being widened first - scores and gradients equal those of the same
    step on the widened batch, bit for bit on the scores."""
    cfg = DrinConfig(dataset_name="wikimel", num_candidates_data=20, max_entity_attr_token_len=10, **TINY)
    sd = synth.make_state_dict(cfg, 8)
    batch = _to_dev(_bf16_features(synth.make_batch(cfg, 5, 72)))
    wide = [t.float() if t.dtype == torch.bfloat16 else t for t in batch]
    model = Model(cfg, precision="bf16x3").to(DEV)
    model.load_state_dict(sd)
    lib = _lib.load()
    _lib.profile_begin()
    a = model(batch[:14])
    prof = _lib.profile_end()
    assert prof["pool"][1] >= 1
    a.sum().backward()
    g1 = [p.grad.clone() if p.grad is not None else None for p in model.parameters()]
    model.zero_grad()
    b = model(wide[:14])
    b.sum().backward()
    assert torch.equal(a, b)
    for x, p in zip(g1, model.parameters()):
        assert (x is None) == (p.grad is None)
        if x is not None:
            assert torch.allclose(x, p.grad, rtol=1e-4, atol=1e-6)
    # the C entry point: bf16 tokens pooled in place == fp32 pooling of the widened tokens; the image means are refused
    from drin_amd.model import _pool_tokens
    assert torch.equal(_pool_tokens(batch[7], batch[8]), _pool_tokens(wide[7], wide[8]))
    ref = O.entity_token_mean(wide[7].cpu(), wide[8].cpu())
    np.testing.assert_allclose(_pool_tokens(batch[7], batch[8]).cpu().numpy(), ref.numpy(), atol=1e-6)
    c = _lib.DrinConfigC()
    lib.drin_default_config(C.byref(c))
    c.batch, c.num_candidates, c.embed_dim, c.entity_tokens, c.feature_dtype = 5, 21, cfg.bert_embed_dim, 10, _lib.FEAT_BF16
    bt = _lib.DrinBatchC()
    bt.entity_text, bt.entity_text_mask, bt.mention_image = batch[7].data_ptr(), batch[8].data_ptr(), batch[4].data_ptr()
    out = torch.empty(5, cfg.resnet_embed_dim, device=DEV)
    assert lib.drin_pool_fwd(C.byref(c), C.byref(bt), None, out.data_ptr(), None, torch.cuda.current_stream().cuda_stream) == _lib.E_UNSUPPORTED


def test_training_reads_entity_tables_through_the_index():
    """Table-form training at full widths (>= 1024 pairs): the pooled / token-0 / image / object TABLES go to `drin_forward` /
    `drin_backward` with `entity_index`, and the static-edge kernels and the vertex-encoder GEMMs (forward x W^T, backward
    dY^T x) address their rows through it - same scores (bit for bit) and gradients as the step on gathered rows."""
    from drin_amd.model import EntityTable, IndexedBatch
    cfg = DrinConfig(dataset_name="wikimel", num_candidates_data=100)
    E, B, N = 300, 12, cfg.num_candidates_model
    tab = synth.make_device_batch(cfg.with_(num_candidates_data=E - 1), 1, 31, DEV)
    table = EntityTable(tab[7][0], tab[8][0], tab[9][0], tab[10][0], tab[11][0])
    men = synth.make_device_batch(cfg, B, 32, DEV)
    g = torch.Generator(device=DEV)
    g.manual_seed(4)
    cand = torch.randint(0, E, (B, N), device=DEV, generator=g)
    ib = IndexedBatch(men[:7], table, cand, men[12], men[13])
    sd = synth.make_state_dict(cfg, 6)
    model = Model(cfg, precision="bf16x3").to(DEV)
    model.load_state_dict(sd)
    assert model._indexed_training_call(ib, True) is not None
    _lib.profile_begin()
    a = model(ib)
    a.sum().backward()
    prof = _lib.profile_end()
    assert prof["pool"][1] <= 2 + 1          # span + region means (+ the one-off table pooling): no per-pair token pooling
    g1 = [p.grad.clone() if p.grad is not None else None for p in model.parameters()]
    model.zero_grad()
    seq, cls = ib.gathered_pooled(cfg)
    from drin_amd.model import _Call, _DrinScore, _param_list
    call = _Call(cfg, seq, _lib.PREC_BF16X3, entity_text_cls=cls)
    b = _DrinScore.apply(call, None, True, *_param_list(model))
    b.sum().backward()
    assert torch.equal(a, b)
    for x, p in zip(g1, model.parameters()):
        assert (x is None) == (p.grad is None)
        if x is not None:
            assert torch.allclose(x, p.grad, rtol=1e-4, atol=1e-6)
    ref = O.forward(sd, [t[:2].cpu() for t in ib.gathered()])
    assert (a[:2].detach().cpu() - ref).abs().max().item() <= 1e-5
    # the C side refuses what it does not build (here: exact-fp32 precision)
    bad = _Call(cfg, ib.mention + [table.pooled_text(cfg)[0], torch.zeros(B, dtype=torch.int64, device=DEV), table.image, table.object,
                                   table.object_score, men[12], men[13]], _lib.PREC_F32, entity_index=cand,
                entity_text_cls=table.pooled_text(cfg)[1])
    lib = _lib.load()
    ws = bad.workspace(True)
    sc = torch.empty(B, N, device=DEV)
    pc = _lib.DrinParamsC()
    from drin_amd.model import _fill_params
    _fill_params(pc, tuple(p.detach().contiguous() for p in _param_list(model)), bad.per_layer)
    assert lib.drin_forward(C.byref(bad.cfg), C.byref(bad.batch), C.byref(pc), ws.data_ptr(), ws.numel(), sc.data_ptr(), 1, None,
                            torch.cuda.current_stream().cuda_stream) == _lib.E_UNSUPPORTED


@pytest.mark.parametrize("seed", range(12))
def test_full_width_random_batches_three_paths_agree(seed):
    """D = 768 / R = 2048 with random layout, batch and candidate counts (1 ... 20 000 pairs): every size lands on a different mix
    of tile shapes, whole-product K splits and tail splits of the split-bf16 kernels.  The folded split-bf16 path, the
    layer-by-layer split-bf16 path and the folded exact-fp32 path agree within fp32 re-association, and a slice of the batch
    agrees with the CPU oracle."""
    g = np.random.default_rng(1000 + seed)
    wm = bool(g.integers(0, 2))
    N = int(g.integers(1, 140))
    B = int(g.choice([1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233, 300]))
    if B * N > 20000:
        B = max(1, 20000 // N)
    cfg = DrinConfig(dataset_name="wikimel" if wm else "wikidiverse", num_candidates_data=N - 1,
                     max_entity_attr_token_len=int(g.integers(3, 20)))
    sd = synth.make_state_dict(cfg, 100 + seed)
    batch = synth.make_device_batch(cfg, B, 500 + seed, DEV)[:14]
    outs = {}
    with torch.no_grad():
        for name, kw in (("folded", dict(precision="bf16x3")), ("layerwise", dict(precision="bf16x3", fused=False)),
                         ("exact", dict(precision="f32"))):
            m = Model(cfg, **kw).to(DEV).eval()
            m.load_state_dict(sd)
            outs[name] = m(batch)
    a = outs["folded"]
    assert torch.isfinite(a).all()
    assert (a - outs["layerwise"]).abs().max().item() <= 1e-5, (B, N, wm)
    assert (a - outs["exact"]).abs().max().item() <= 1e-5, (B, N, wm)
    k = min(B, 2)
    ref = O.forward(sd, [t[:k].cpu() for t in batch])
    assert (a[:k].cpu() - ref).abs().max().item() <= 1e-5
