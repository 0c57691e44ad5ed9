"""-m gpu, round 5: the oracle as the yardstick where round 4 used the library's own exact-fp32 path, at the sizes `bench.py` times
(VERDICT r4 "missing" 3-4, "weak" 3):

* the bf16-stored-feature stream kernel (`k_entity_stream<..., __bf16>`: the flat two-rows-per-three-loads token walk) with
  whole-mention workgroups, B = 2 048, token counts 4 .. 27;
* the WikiDiverse-shaped leg's call size, B = 16 384, fp32- and bf16-stored features;
* TRAINED weights: the exact-fp32 and the default split-bf16 arithmetic against `O.forward` with the same `state_dict`;
* the per-entity cache's mixed-f16 rows on entity rows of extreme magnitude: both row formats against the fp64 oracle.

Every case compares with the CPU oracle (`oracle/drin_oracle.py`, pinned to the reference by `tests/golden`) on slices of the batch the HIP
path scored in ONE call of the timed size; `drin_workgroups_per_mention` / the library's launch profile assert that the timed
instantiation is the one that ran."""
import ctypes as C
import os

import pytest
import torch

from drin_amd import _lib, synth
from drin_amd.config import DrinConfig, wikimel_config
from drin_amd.metrics import TripletLoss
from drin_amd.model import Model
from oracle import drin_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
FEATS = (0, 4, 5, 7, 9, 10)          # the six feature tensors of the 14-sequence (drin/data.py:110-126)


def _threads():
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count() or 1)))


def _cfg_c(cfg: DrinConfig, B: int, precision: int, tokens: int = 0, feature_dtype: int = 0):
    c = _lib.DrinConfigC()
    _lib.check(_lib.load().drin_default_config(C.byref(c)))
    c.batch, c.num_candidates, c.embed_dim, c.image_dim = B, cfg.num_candidates_model, cfg.bert_embed_dim, cfg.resnet_embed_dim
    c.entity_tokens, c.precision, c.feature_dtype = tokens, precision, feature_dtype
    return c


def _slices(B, width=8):
    return (slice(0, width), slice(B // 2 - width // 2, B // 2 + width // 2), slice(B - width, B))


def _host(batch, rows):
    return [(t[rows].float() if t.dtype == torch.bfloat16 else t[rows]).cpu() for t in batch[:14]]


# ---- (a) bf16-stored features at the timed call size ---------------------------------------------------------------------
def test_bf16_stored_features_whole_mention_workgroups_against_the_oracle():
    """WikiMEL-shaped D = 768 / R = 2 048 / N = 101, B = 2 048 (one workgroup per mention, as `bench.py`'s `wikimel_bf16_features`
    leg at B = 4 096), the six feature tensors stored as bf16 and read in place; T = 27 with token counts ~ U{4..27}: kept-row
    counts 2 .. 25 of both parities, i.e. the flat walk's four-pairs-in-flight body, single pairs and the unpaired last row
    (`ghmfc.py:245-249`).  24 mentions (first / middle / last 8) against the oracle on the same stored values widened to fp32:
    <= 1e-5 (bar 1e-4), arg-max equal; the same bits on a second call."""
    T = 27
    cfg = wikimel_config(max_entity_attr_token_len=T)
    B = 2048
    lib = _lib.load()
    assert lib.drin_workgroups_per_mention(C.byref(_cfg_c(cfg, B, _lib.PREC_BF16X3, tokens=T, feature_dtype=1)), 0) == 1
    sd = synth.make_state_dict(cfg, 7)
    model = Model(cfg).to(DEV).eval()
    model.load_state_dict(sd)
    batch = synth.make_device_batch(cfg, B, 23, DEV, dtype=torch.bfloat16)[:14]
    assert all(batch[i].dtype == torch.bfloat16 for i in FEATS)
    ntok = batch[8].sum(-1)
    assert int(ntok.min()) == 4 and int(ntok.max()) == T
    for rows in _slices(B):                                           # the compared slices hold odd and even kept-row counts
        kept = (ntok[rows] - 2).flatten()
        assert bool((kept % 2 == 0).any()) and bool((kept % 2 == 1).any())
    _threads()
    with torch.no_grad():
        model([t[:8] for t in batch])                                 # folds the weights
        _lib.profile_begin()
        out = model(batch)
        prof = _lib.profile_end()
        assert prof["stream"][1] == 1 and prof["gemm_planes"][1] >= 3 and prof["gemm"][1] == 0   # x_i C_i^T on the planes kernel (bf16 rows in place)
        assert torch.isfinite(out).all() and torch.equal(out, model(batch))
        worst = 0.0
        for rows in _slices(B):
            ref = O.forward(sd, _host(batch, rows))
            worst = max(worst, (out[rows].cpu() - ref).abs().max().item())
            assert (out[rows, :-1].argmax(1).cpu() == ref[:, :-1].argmax(1)).all()
    print(f"bf16-stored features, B={B} whole-mention workgroups, ntok 4..{T}: max |score - oracle(widened)| {worst:.2e}")
    assert worst <= 1e-5


# ---- (b) the WikiDiverse-shaped leg at its call size -----------------------------------------------------------------------
@pytest.mark.parametrize("features", ["f32", "bf16"])
def test_wikidiverse_leg_call_size_against_the_oracle(features):
    """BASELINE config 2 as `bench.py`'s `wikidiverse` leg times it: 16 384 mentions x 11 candidates in ONE call (one workgroup per
    mention, the pooled-text instantiation of `k_entity_stream`; `model.py:71-92`), fp32- and bf16-stored features, default
    arithmetic.  48 mentions against the oracle <= 1e-5, arg-max equal."""
    cfg = DrinConfig()
    B = 16384
    lib = _lib.load()
    assert lib.drin_workgroups_per_mention(C.byref(_cfg_c(cfg, B, _lib.PREC_BF16X3, feature_dtype=int(features == "bf16"))), 0) == 1
    sd = synth.make_state_dict(cfg, 7)
    model = Model(cfg).to(DEV).eval()
    model.load_state_dict(sd)
    batch = synth.make_device_batch(cfg, B, 100, DEV, dtype=torch.bfloat16 if features == "bf16" else torch.float32)[:14]
    _threads()
    with torch.no_grad():
        model([t[:8] for t in batch])
        _lib.profile_begin()
        out = model(batch)
        prof = _lib.profile_end()
        assert prof["stream"][1] == 1 and prof["gemm"][1] == 0        # fused path, every contraction split-bf16 (mention-sized ones too: 16 384 rows)
        assert torch.isfinite(out).all() and torch.equal(out, model(batch))
        worst = 0.0
        for rows in _slices(B, 16):
            ref = O.forward(sd, _host(batch, rows))
            worst = max(worst, (out[rows].cpu() - ref).abs().max().item())
            assert (out[rows, :-1].argmax(1).cpu() == ref[:, :-1].argmax(1)).all()
    print(f"wikidiverse-shaped B={B}, {features} features: max |score - oracle| {worst:.2e}")
    assert worst <= 1e-5


# ---- (c) trained weights: the oracle, not the library's fp32 path, as the yardstick -----------------------------------------
def test_trained_weights_exact_and_default_arithmetic_against_the_oracle():
    """40 Adam steps of the reference's loop (`train.py:30-56`) on the learnable synthetic stream at the reference's width and batch
    (D = 768, R = 2 048, N = 101, B = 64; the loop of `test_precision_modes_on_trained_weights`), then 512 held-out mentions scored
    in ONE call by the exact-fp32 and by the default split-bf16 arithmetic; the first 32 of them against `O.forward` with the SAME
    trained `state_dict` (fp32, and fp64 as the arbiter).  Training steepens the vertex -> score map, so every distance grows
    against initialisation (3e-7 / 1.3e-6): the bound asserted is 2e-5 (bar 1e-4)."""
    from drin_amd.train import make_adam
    cfg = wikimel_config(max_entity_attr_token_len=8, batch_size=64)
    torch.manual_seed(0)
    model = Model(cfg).to(DEV)
    opt = make_adam(model, cfg.learning_rate)
    loss_fn = TripletLoss(cfg.triplet_margin)
    first = last = None
    for i in range(40):
        b = [t.to(DEV) for t in synth.plant_gold_signal(cfg, synth.make_device_batch(cfg, 64, 50 + i, "cpu"), 0.15)]
        opt.zero_grad(set_to_none=True)
        loss = loss_fn(b[14], model(b[:14]))
        loss.backward()
        opt.step()
        first, last = (float(loss.detach()) if first is None else first), float(loss.detach())
    assert last < 0.35 * first, (first, last)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    held = synth.plant_gold_signal(cfg, synth.make_device_batch(cfg, 512, 999, "cpu"), 0.15)
    dheld = [t.to(DEV) for t in held]
    _threads()
    head = [t[:32] for t in held[:14]]
    with torch.no_grad():
        ref32 = O.forward(sd, head)
        ref64 = O.forward({k: v.double() for k, v in sd.items()}, head, dtype=torch.float64)
    errs = {}
    for prec in ("f32", "bf16x3"):
        m = Model(cfg, precision=prec).to(DEV).eval()
        m.load_state_dict(sd)
        with torch.no_grad():
            got = m(dheld[:14])[:32].cpu()
        errs[prec] = ((got - ref32).abs().max().item(), (got.double() - ref64).abs().max().item())
        assert (got[:, :-1].argmax(1) == ref32[:, :-1].argmax(1)).all()
    own = (ref32.double() - ref64).abs().max().item()
    spread = (ref32[:, :-1].max(1).values - ref32[:, :-1].median(1).values).mean().item()
    print(f"trained weights (loss {first:.3f} -> {last:.3f}, top-minus-median score {spread:.2f}); 32 held-out mentions, max |score - oracle|: "
          f"HIP f32 {errs['f32'][0]:.2e} (vs fp64 oracle {errs['f32'][1]:.2e}), HIP bf16x3 {errs['bf16x3'][0]:.2e} (vs fp64 {errs['bf16x3'][1]:.2e}); "
          f"the fp32 oracle itself vs fp64 {own:.2e}")
    assert errs["f32"][0] <= 2e-5 and errs["bf16x3"][0] <= 2e-5, errs
    assert errs["f32"][1] <= 2e-5 and errs["bf16x3"][1] <= 2e-5, errs


# ---- (d) mixed-f16 cache rows on entity rows of extreme magnitude: both formats against the fp64 oracle ---------------------
def test_mixed_f16_cache_rows_of_extreme_magnitude_against_the_fp64_oracle():
    """`test_mixed_f16_cache_rows_hold_rows_of_any_magnitude` (round 4) compares the two row formats with EACH OTHER on tables whose
    image rows are scaled by 1e-30 .. 3e37 and argued - without a yardstick - about the mentions where they differed.  Here BOTH
    formats meet the oracle on the gathered 14-sequence (`model.py:121-153`), in fp32 (what the reference computes) and in fp64 (the
    arbiter of rounding), in three groups of mentions:
      A  candidates with image rows x 1e-6, x 1e-30, an all-zero image row, an all-zero object score next to ordinary ones:
         both formats within 1e-5 of the fp64 oracle;
      B  additionally three candidates with rows x 1e6 (W_v1(ei0) ~ 3e5: inf as plain fp16, held by the field's power-of-two scale;
         their dynamic edges saturate the sigmoid in either format).  What parts the formats there is the STATIC image-image edge
         (`model.py:84-92`): the fp16 object row holds it to ~1e-5, and that edge multiplies the candidate's layer-1 image vertex in the
         mention aggregate mean_n(ii ei) (`model.py:143-144`) - a vertex 1e6 times the others' DOMINATES the mean, so the aggregate
         inherits the relative error of ONE edge instead of averaging 101 of them.  Measured: the fp32 rows stay within 1e-6 of the
         fp64 oracle, the mixed rows are 1e-5 .. 7e-5 off on half of such mentions (inside the 1e-4 bar, outside the 1e-5 guard) -
         which is why `EntityTable` does not USE the format for such a table (round 6: the build scans the image rows and falls back
         to fp32 rows; asserted here: <= 1e-5 from the fp64 oracle); the fp16 fields are then forced (`force=True`) and
         The test pins the cause by EMULATION: the fp64 oracle with exactly the three fp16 fields passed through the format's
         rounding (`oracle/precision_emulation.py`) reproduces the HIP scores of the mixed rows to 1e-5 - the deviation is the
         storage format's and nothing else's (`include/drin_hip.h`, drin_cache_format, states the limit);
      C  additionally a row scaled to 3e37: its layer-1 pre-activations are ~1e37 and their LayerNorm variance overflows fp32 in the
         reference itself (`model.py:128`) - fp64 is no yardstick there; the formats must agree with each other and stay finite."""
    from drin_amd.model import EntityTable, IndexedBatch
    cfg = wikimel_config(max_entity_attr_token_len=4)
    sd = synth.make_state_dict(cfg, 7)
    E, B, N = 600, 96, cfg.num_candidates_model
    g = torch.Generator(device=DEV).manual_seed(3)
    img = torch.randn(E, cfg.resnet_embed_dim, device=DEV, generator=g)
    img[0:50] *= 1e6
    img[150] *= 3e37 / img[150].abs().max()
    img[50:100] *= 1e-6
    img[100:149] *= 1e-30
    img[151] = 0.0
    score = torch.rand(E, 1, device=DEV, generator=g)
    score[152] = 0.0
    table = EntityTable(torch.randn(E, 4, cfg.bert_embed_dim, device=DEV, generator=g), torch.ones(E, 4, dtype=torch.int64, device=DEV),
                        img, torch.randn(E, 1, cfg.resnet_embed_dim, device=DEV, generator=g), score)
    men = synth.make_device_batch(cfg.with_(num_candidates_data=0), B, 12, DEV)
    cand = torch.randint(50, E, (B, N), device=DEV, generator=g)                       # rows 0-49 (x 1e6) only where planted
    cand[cand == 150] = 153
    cand[:, :5] = torch.tensor([60, 120, 151, 152, 300], device=DEV)                   # every mention: small, tiny, zero rows
    third = B // 3
    cand[third:, 5:8] = torch.tensor([0, 17, 49], device=DEV)                          # groups B, C: rows x 1e6
    cand[2 * third:, 8] = 150                                                          # group C: the 3e37 row
    sims = 20.0 + 5.0 * torch.randn(2, B, N, device=DEV, generator=g)
    ib = IndexedBatch(men[:7], table, cand, sims[0], sims[1])
    model = Model(cfg).to(DEV).eval()
    model.load_state_dict(sd)
    with torch.no_grad():
        table.enable_cache(True)
        full = model(ib).cpu()
        # round 6: the format meets the guard BY CONSTRUCTION - the build's scan of the image table finds the 50 rows x 1e6 and the 3e37
        # row and gives this table fp32 rows (saying so); the fp16 fields below are FORCED, to keep the limit of the format pinned
        table.enable_cache(True, format="mixed_f16")
        with pytest.warns(UserWarning, match="mixed_f16: 51 of 600 entity image rows.*gets fp32 cache rows"):
            fell_back = model(ib).cpu()
        assert table.cache_format_used == "f32" and torch.equal(fell_back, full)
        table.enable_cache(True, format="mixed_f16", force=True)
        with pytest.warns(UserWarning, match=r"mixed_f16 \(forced\): 51 of 600 entity image rows"):
            mixed = model(ib).cpu()
        assert table.cache_format_used == "mixed_f16"
    table.enable_cache(False)
    assert torch.isfinite(mixed).all() and torch.isfinite(full).all()
    _threads()
    host = [t.cpu() for t in ib.gathered()]
    ab = slice(0, 2 * third)
    trace = {}
    with torch.no_grad():
        ref32 = O.forward(sd, [t[ab] for t in host])
        ref64 = O.forward({k: v.double() for k, v in sd.items()}, [t[ab] for t in host], dtype=torch.float64, trace=trace)
    own = (ref32.double() - ref64).abs()
    e32, e16 = (full[ab].double() - ref64).abs(), (mixed[ab].double() - ref64).abs()
    d = (mixed - full).abs()
    a_rows, b_rows = slice(0, third), slice(third, 2 * third)
    print(f"group A: fp32 oracle vs fp64 {own[a_rows].max().item():.2e}; fp32 rows {e32[a_rows].max().item():.2e}, mixed-f16 rows "
          f"{e16[a_rows].max().item():.2e} from the fp64 oracle; formats apart {d[a_rows].max().item():.2e}")
    assert own[a_rows].max().item() <= 1e-5 and e32[a_rows].max().item() <= 1e-5 and e16[a_rows].max().item() <= 1e-5
    # group B: what the format's rounding of its three fp16 fields does to the fp64 forward - and nothing else - is what the HIP path shows
    from oracle.precision_emulation import MIXED_F16_FIELDS, scores_with_rounded_cache_fields
    with torch.no_grad():
        emul = scores_with_rounded_cache_fields({k: v.double() for k, v in sd.items()}, [t[b_rows] for t in host], MIXED_F16_FIELDS)
        only_ohat = scores_with_rounded_cache_fields({k: v.double() for k, v in sd.items()}, [t[b_rows] for t in host], ("ohat",))
    fmt = (emul - ref64[b_rows]).abs()                                   # the format's own effect, emulated
    off = e16[b_rows].max(1).values > 1e-5
    print(f"group B: fp32 oracle vs fp64 {own[b_rows].max().item():.2e}; fp32 rows {e32[b_rows].max().item():.2e} from the fp64 oracle; mixed-f16 rows: "
          f"{int(off.sum())} of {third} mentions > 1e-5 (max {e16[b_rows].max().item():.2e}); emulated format effect max {fmt.max().item():.2e} "
          f"(the object row alone: {(only_ohat - ref64[b_rows]).abs().max().item():.2e}); HIP mixed rows vs the emulation {(mixed[b_rows].double() - emul).abs().max().item():.2e}")
    assert own[b_rows].max().item() <= 1e-5 and e32[b_rows].max().item() <= 1e-5          # the fp32 rows are NOT noise there
    assert (mixed[b_rows].double() - emul).abs().max().item() <= 1e-5                      # the deviation IS the storage format's rounding
    assert (only_ohat - emul).abs().max().item() <= 1e-5                                   # ... of the object row (the static ii edge)
    assert e16[b_rows].max().item() <= 1e-4                                                # the FORCED format: inside the bar (measured 6.7e-5)
    assert (fell_back[ab].double() - ref64).abs().max().item() <= 1e-5                     # what a caller gets: the guard, by construction
    c = slice(2 * third, B)
    print(f"group C (a 3e37 row among the candidates): finite; formats apart max {d[c].max().item():.2e}, {(d[c] <= 1e-5).float().mean().item():.4f} within 1e-5")
    assert (d[c] <= 1e-5).float().mean().item() >= 0.95
