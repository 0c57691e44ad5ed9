"""bench.py - mention x candidate pairs scored per second on WikiMEL-shaped synthetic batches.

    python bench.py [--gpus N --steps K --warmup W] [--workload wikimel|wikidiverse] [--batch B]

One "step" = one pass of the DRIN scoring path (Model.forward, drin/model.py:164-209) over one
batch of B mentions x N candidates of seeded synthetic features that are already resident in HBM.
N>1: `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`; every rank scores
its own shard of mentions (weak scaling, no data-path collective - mentions are independent,
SURVEY.md 8e); the timed region is bracketed by barrier + synchronize and the max over ranks is used.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     : dominant kernel class, HIP-event timed inside this process
  cpu_baseline : the CPU oracle (oracle/drin_oracle.py) timed on this host's cores on a bounded sample
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from drin_amd import _lib, synth  # noqa: E402
from drin_amd.config import DrinConfig, wikimel_config  # noqa: E402
from drin_amd.model import Model  # noqa: E402

PEAK_F32_MATRIX_TFLOPS = 157.3   # MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA" (dense)
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md "HBM3E peak BW" (spec)


def path_flops_per_pair(D, R, layers, dynamic, fused):
    """Pair-sized contraction FLOPs per pair the HIP path executes.

    generic path: dead work of the last layer removed (SURVEY.md 8a tail) -> 10.22 MFLOP;
    fused path (csrc/fused_forward.hip): x_t C_t^T, x_i C_i^T, et' W_h2^T -> 5.51 MFLOP.
    """
    if fused and layers == 2:
        return 2.0 * D * D + 2.0 * R * D + 2.0 * D * D
    f = 2.0 * D * D + 2.0 * R * D                       # W_et, W_ei
    for l in range(layers):
        last = l == layers - 1
        f += 2.0 * D * D * (1 if last else 2)           # W_h on et (+ ei unless last)
        if dynamic and not last:
            f += 2.0 * D * D * 2                        # W_v on et, ei
    return f


def mention_flops(D, R, fused):
    """Mention-sized contraction FLOPs per mention (they ride in the same kernel class)."""
    if fused:  # mt0, mi0, [hm|fu], q, T (two), W_h1 x2, hm2 x2, mt2
        return 2.0 * D * D + 2.0 * R * D + 2 * (2.0 * D * 2 * D) + 2 * (2.0 * D * (D + R)) + 2 * (2.0 * D * D + 2.0 * R * D) \
            + 2 * 2.0 * D * D + 2 * 2.0 * D * D + 2.0 * D * D
    return 2.0 * D * D + 2.0 * R * D + 2 * 2.0 * D * D * 2 + 2.0 * D * D


def algorithmic_bytes_per_pair(cfg, batch):
    """Compulsory input bytes per pair: entity-side rows that are actually needed
    + the mention-side bytes amortised over the N candidates + the 4-byte score."""
    if not isinstance(batch, (list, tuple)):          # table form: 8-byte candidate index + the gathered pooled row
        D, R, N = cfg.bert_embed_dim, cfg.resnet_embed_dim, cfg.num_candidates_model
        men = 3 * D * 4 + cfg.resnet_num_region * R * 4 + cfg.object_topk_mention * (R + 1) * 4 + 16
        return 8 + D * 4 + R * 4 + cfg.object_topk_entity * (R + 1) * 4 + 8 + men / N + 4
    B, N = batch[0].shape[0], cfg.num_candidates_model
    D, R = cfg.bert_embed_dim, cfg.resnet_embed_dim
    es = batch[0].element_size()                         # feature storage: 4 (fp32) or 2 (bf16) bytes
    if cfg.token_level_entities:
        ntok = batch[8].sum(-1).float()
        rows = (ntok - 2).clamp(min=0) + 1              # pooled tokens 1..ntok-2 plus the CLS row
        ent_text = float(rows.mean()) * D * es + batch[8].shape[-1] * 8
    else:
        ent_text = D * es
    ent = ent_text + R * es + cfg.object_topk_entity * R * es + (cfg.object_topk_entity + 2) * 4
    span = float((batch[3] - batch[2]).float().mean())
    men = span * D * es + cfg.resnet_num_region * R * es + cfg.object_topk_mention * (R * es + 4) + 16
    return ent + men / N + 4


def measured_traffic(kernel_prefix, B, precision, fused, features="f32", section="kernels"):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (tools/collect_pmc.py),
    valid only for the configuration they were taken on (default workload, or the cached-table one); None otherwise."""
    path = os.path.join(REPO, "profiles", "r1_hbm_traffic.json")
    want_b = 4096 if section == "kernels" else 256
    if not (os.path.exists(path) and B == want_b and precision == "bf16x3" and fused and features == "f32"):
        return None
    try:
        for k, v in json.load(open(path)).get(section, {}).items():
            if kernel_prefix in k:
                return v["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def host_cores() -> int:
    """CPU cores this process may actually use (affinity and cgroup quota, not the node's total)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("DRIN_CPU_THREADS", "16"))))


def cpu_baseline(cfg, sd, seconds=12.0, model=None, dev=None, features="f32"):
    """The CPU oracle timed on the host cores this process may use; with `model`, the same sample batches are also
    scored by the HIP path and compared (max |score error|, top-1 agreement) - the checker, never the thing measured."""
    from oracle import drin_oracle as O

    cores = host_cores()
    torch.set_num_threads(cores)
    B = 8 if cfg.token_level_entities else 64
    batch = synth.make_batch(cfg, B, 3)
    parity = None
    if model is not None:
        err, agree, total, err32, agree32 = 0.0, 0, 0, 0.0, 0
        feat_slots = (0, 4, 5, 7, 9, 10)    # the six feature tensors of the 14-sequence
        with torch.no_grad():
            for seed in range(3, 3 + (8 if cfg.token_level_entities else 2)):   # 64 / 128 mentions
                b = synth.make_batch(cfg, B, seed)
                ref32 = O.forward(sd, b)
                if features == "bf16":   # stored as bf16 on the device; the oracle scores the same values widened
                    stored = [t.to(torch.bfloat16) if i in feat_slots else t for i, t in enumerate(b[:14])]
                    ref = O.forward(sd, [t.float() if t.dtype == torch.bfloat16 else t for t in stored])
                else:
                    stored, ref = b[:14], ref32
                got = model([t.to(dev) for t in stored]).cpu()
                err = max(err, (got - ref).abs().max().item())
                agree += int((got[:, :-1].argmax(1) == ref[:, :-1].argmax(1)).sum())
                err32 = max(err32, (got - ref32).abs().max().item())
                agree32 += int((got[:, :-1].argmax(1) == ref32[:, :-1].argmax(1)).sum())
                total += B
        parity = {"max_abs_score_err": err, "top1_agreement": agree / total, "mentions": total,
                  "against": "CPU oracle (pinned to the reference by tests/golden), same synthetic inputs"
                             + (" as stored (bf16 values widened)" if features == "bf16" else "")}
        if features == "bf16":           # what storing the features as bf16 costs against the fp32 inputs
            parity["vs_fp32_features"] = {"max_abs_score_err": err32, "top1_agreement": agree32 / total}
    with torch.no_grad():
        O.forward(sd, batch)  # warm-up
        t0 = time.perf_counter()
        it = 0
        while True:
            O.forward(sd, batch)
            it += 1
            el = time.perf_counter() - t0
            if el >= seconds or it >= 2000:
                break
    pairs = it * B * cfg.num_candidates_model
    # the same arithmetic with the reference's own Python loops over mentions and candidates kept
    # (baselines/ghmfc.py:58-59,246-249; model.py:86-91): its cost profile, ~5 s sample
    with torch.no_grad():
        O.reference_style_forward(sd, batch)  # warm-up
        t1 = time.perf_counter()
        it_ref = 0
        while True:
            O.reference_style_forward(sd, batch)
            it_ref += 1
            el_ref = time.perf_counter() - t1
            if el_ref >= 5.0 or it_ref >= 500:
                break
    ref_style = it_ref * B * cfg.num_candidates_model / el_ref
    out = {"value": pairs / el, "unit": "pairs/s", "cores": cores, "kind": "port",
           "reference_style_loops_value": ref_style,
           "sample": f"{it} forwards of the CPU oracle on {cfg.dataset_name}-shaped B={B} N={cfg.num_candidates_model} fp32, "
                     f"torch {torch.get_num_threads()} threads, {el:.1f} s"}
    return out, parity


def train_roofline(cfg, B, N, prof, args):
    """The dominant kernel class of a training step is the split-bf16 GEMM family (k_gemm_bf16x3 + k_gemm_tn_bf16x3):
    algorithmic FLOPs of the pair-sized contractions (forward with the dead work of the last layer removed, dX and dW of
    backward: 17 D^2 + 2 . 2 R D multiply-adds per pair for two dynamic layers) over its summed launch time."""
    D, R, nl = cfg.bert_embed_dim, cfg.resnet_embed_dim, cfg.num_gcn_layers
    dyn = cfg.gcn_edge_type == "dynamic"
    # forward: W_et, W_ei, per layer W_h on et (+ ei unless last), W_v on (et, ei) unless last / static
    fwd = 2.0 * D * D + 2.0 * R * D
    bwd = 2.0 * D * D + 2.0 * R * D          # dW of the two vertex encoders (inputs carry no gradient)
    for l in range(nl):
        last = l == nl - 1
        wh = (1 if last else 2) * 2.0 * D * D
        wv = 0.0 if (last or not dyn) else 2 * 2.0 * D * D
        fwd += wh + wv
        bwd += 2 * (wh + wv)                 # dX and dW
    ms, launches = prof.get("gemm_x3", (0.0, 0))
    if ms <= 0:
        return None
    tf = (fwd + bwd) * B * N * args.steps / (ms * 1e-3) / 1e12
    x3 = args.precision in ("bf16x3", "bf16")
    peak = PEAK_BF16_MFMA_TFLOPS if x3 else PEAK_F32_MATRIX_TFLOPS
    out = {"bound": "mfma", "kernel": "k_gemm_bf16x3 + k_gemm_tn_bf16x3", "achieved": tf, "peak": peak, "unit": "TFLOP/s",
           "frac": tf / peak, "traffic": None, "launches": int(launches), "avg_launch_ms": ms / max(launches, 1),
           "flops_per_pair": fwd + bwd}
    if x3:
        out["executed_bf16_tflops"], out["executed_frac"] = 3 * tf, 3 * tf / peak
    return out


def bench_train(args, cfg, model, dev, world, rank, B, barrier):
    """One optimisation step of train.py:30-56 per "step": forward (intermediates kept), TripletLoss, backward
    through the HIP kernels, one RCCL all-reduce of the flat gradient bucket (world > 1), Adam."""
    from drin_amd.metrics import DeviceLossMetric
    from drin_amd.train import GradBucket

    model.train()
    full = synth.make_device_batch(cfg, B, 200 + rank, dev, dtype=torch.bfloat16 if args.features == "bf16" else torch.float32)
    batch, y = full[:14], full[14]
    if args.train_form != "gathered":
        # table form: the entity tables live on the device, a step carries candidate indices.
        # "table": the library pools every entity's tokens once and gathers pooled rows; "table-tokens": the token
        # blocks are gathered with torch indexing every step (what the reference's loader does on the host)
        from drin_amd.model import EntityTable, IndexedBatch
        E = args.train_entities
        tab = synth.make_device_batch(cfg.with_(num_candidates_data=E - 1), 1, 300 + rank, dev,
                                      dtype=torch.bfloat16 if args.features == "bf16" else torch.float32)
        table = EntityTable(tab[7][0], tab[8][0] if cfg.token_level_entities else None, tab[9][0], tab[10][0], tab[11][0])
        del tab
        g = torch.Generator(device=dev)
        g.manual_seed(17 + rank)
        cand = torch.randint(0, E, (B, cfg.num_candidates_model), device=dev, generator=g)
        ib = IndexedBatch(batch[:7], table, cand, batch[12], batch[13])
        full = None
        batch = ib if args.train_form == "table" else None
    loss_fn = DeviceLossMetric(cfg.triplet_margin, cfg.metrics_topk, dev)   # loss + top-k counters in one library call, as MELRunner
    if args.torch_loss:
        from drin_amd.metrics import TripletLoss
        loss_fn = TripletLoss(cfg.triplet_margin)
    opt = torch.optim.Adam(model.parameters(), lr=cfg.learning_rate, capturable=args.graph, **({"fused": True} if args.fused_adam else {}))   # default: as train.py:55-56
    bucket = GradBucket(list(model.parameters()))

    def eager_step():
        opt.zero_grad(set_to_none=True)
        loss = loss_fn(y, model(batch if batch is not None else ib.gathered()))
        loss.backward()
        bucket.allreduce_mean()
        opt.step()
        return loss

    step = eager_step
    if args.graph:
        # the library allocates nothing and never synchronises, so the step is capture-safe: ~100 launches
        # (forward, loss, backward, Adam) become one hipGraph replay and the host drops out of the loop
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                eager_step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(graph):
            static_loss = eager_step()

        def step():
            graph.replay()
            return static_loss

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    _lib.profile_begin(1 << 16)
    for _ in range(args.steps):
        eager_step()
    prof = _lib.profile_end()
    if rank == 0:
        N = cfg.num_candidates_model
        print(json.dumps({
            "metric": "mention x candidate pairs trained/sec (forward + backward + Adam)" + (" [hipGraph replay]" if args.graph else ""),
            "value": B * N * world * args.steps / elapsed, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"{cfg.dataset_name}-shaped training step, {N - 1}-cand, per-GPU batch {B} (args.py:118)"
                                   + (", features stored as bf16 (token blocks pooled in place, the rest widened)" if args.features == "bf16" else "")
                                   + ({"gathered": "", "table": f", candidates indexed into a device-resident table of {args.train_entities} entities (tokens pooled once per entity)",
                                       "table-tokens": f", candidates gathered from a device-resident table of {args.train_entities} entities with torch indexing every step"}[args.train_form]),
                       "global_batch": B * world, "parallelism": f"dp{world}, one flat-bucket RCCL all-reduce per step"},
            "kernel_ms_per_step": {k: v[0] / args.steps for k, v in prof.items()},
            "roofline": train_roofline(cfg, B, N, prof, args),
            "final_loss": float(loss)}))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=None,
                    help="untimed steps (default 3; 30 in train mode: a 10 ms step needs ~0.3 s before the clocks have ramped - "
                         "the first run after an idle or lightly loaded GPU otherwise measures 15 instead of 9.3 ms at batch 512)")
    ap.add_argument("--workload", default="wikimel", choices=["wikimel", "wikidiverse", "table"],
                    help="wikimel: 100-cand token-level (headline); wikidiverse: 10-cand pooled; table: BASELINE config 5 - "
                         "1000 candidates per mention gathered on the device from a table of --entities random entities")
    ap.add_argument("--entities", type=int, default=1_000_000)
    ap.add_argument("--entity-cache", action="store_true",
                    help="table workload: score from the per-entity precompute cache (SURVEY.md 8f-2; built during warm-up)")
    ap.add_argument("--batch", type=int, default=0, help="mentions per step per GPU (default 4096 wikimel / 16384 wikidiverse / 512 table)")
    ap.add_argument("--precision", default="bf16x3", choices=["f32", "bf16x3", "bf16"],
                    help="contraction arithmetic: exact fp32 MFMA, or split-bf16 (3 bf16 MFMAs, fp32 accumulate; "
                         "max score error vs the fp32 reference 1.4e-6, tests/test_gpu_parity.py)")
    ap.add_argument("--features", default="f32", choices=["f32", "bf16"],
                    help="storage type of the six feature tensors (BASELINE configs 2-3 name bf16): bf16 halves the bytes "
                         "of the HBM-bound pass; arithmetic stays fp32-equivalent and the scores equal the reference "
                         "forward on the same features widened to fp32 (tests/test_gpu_parity.py::test_bf16_stored_features)")
    ap.add_argument("--generic", action="store_true", help="use the layer-by-layer path instead of the fused one")
    ap.add_argument("--mode", default="score", choices=["score", "train"],
                    help="score: the scoring forward (headline metric); train: forward + TripletLoss + backward + "
                         "gradient all-reduce + Adam step (BASELINE configs 3-4), reported under the same unit")
    ap.add_argument("--train-form", default="gathered", choices=["gathered", "table", "table-tokens"],
                    help="train mode: per-pair tensors as the reference's loader delivers them (default), or candidate indices into device-resident entity tables")
    ap.add_argument("--train-entities", type=int, default=50_000, help="rows of the entity tables of --train-form table")
    ap.add_argument("--graph", action="store_true",
                    help="capture the whole step (score: the scoring call; train: forward, loss, backward, Adam) in one hipGraph and replay it")
    ap.add_argument("--fused-adam", action="store_true", help="train mode: torch's fused single-kernel Adam instead of its default multi-tensor one (7 launches): -0.27 ms per step, different rounding")
    ap.add_argument("--torch-loss", action="store_true", help="train mode: the torch TripletLoss instead of the library's loss/metric call")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.warmup is None:
        args.warmup = 30 if args.mode == "train" else 3

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    if args.workload == "table":
        cfg = DrinConfig(num_candidates_data=1000)       # pooled-text entity rows: 19.7 KB per entity in fp32
    else:
        cfg = wikimel_config() if args.workload == "wikimel" else DrinConfig()
    # WikiMEL: 4096 mentions = 413 696 pairs and 92 GB of resident inputs per step (of 288 GB): large steps
    # amortise the latency-bound mention-side kernels and the GEMM tile quantisation (22.6 vs 20.0 M pairs/s
    # at 1024 mentions)
    default_b = {"wikimel": 4096, "wikidiverse": 16384, "table": 512}[args.workload]
    B = args.batch or (default_b if args.mode == "score" else 64)
    sd = synth.make_state_dict(cfg, 7)
    model = Model(cfg, precision=args.precision, fused=not args.generic).to(dev).eval()
    model.load_state_dict(sd)
    N = cfg.num_candidates_model
    if args.workload == "table":
        from drin_amd.model import EntityTable, IndexedBatch
        E, D, R = args.entities, cfg.bert_embed_dim, cfg.resnet_embed_dim
        g = torch.Generator(device=dev)
        g.manual_seed(7)
        table = EntityTable(torch.randn(E, D, device=dev, generator=g), None, torch.randn(E, R, device=dev, generator=g),
                            torch.randn(E, 1, R, device=dev, generator=g), torch.rand(E, 1, device=dev, generator=g))
        men = synth.make_device_batch(cfg.with_(num_candidates_data=0), B, 100 + rank, dev)
        cand = torch.randint(0, E, (B, N), device=dev, generator=g)
        sims = 20.0 + 5.0 * torch.randn(2, B, N, device=dev, generator=g)
        batch = IndexedBatch(men[:7], table, cand, sims[0], sims[1])
        if args.entity_cache:
            table.enable_cache()
    else:
        batch = synth.make_device_batch(cfg, B, 100 + rank, dev,
                                        dtype=torch.bfloat16 if args.features == "bf16" else torch.float32)[:14]
    pairs_per_step = B * N

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize(dev)

    if args.mode == "train":
        return bench_train(args, cfg, model, dev, world, rank, B, barrier)

    with torch.no_grad():
        run = lambda: model(batch)  # noqa: E731
        if args.graph:
            # small batches are a chain of ~25 short launches: the library allocates nothing and never synchronises,
            # so the whole scoring call replays as one hipGraph (the weights' folded products are built before capture)
            for _ in range(3):
                model(batch)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                model(batch)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = model(batch)

            def run():
                graph.replay()
                return static_out
        for _ in range(args.warmup):
            run()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = run()
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        assert torch.isfinite(out).all()

        # instrumented pass: per-kernel-class GPU time from HIP events on the launch stream
        _lib.profile_begin(1 << 16)
        for _ in range(args.steps):
            model(batch)
        prof = _lib.profile_end()

    if rank == 0:
        D, R = cfg.bert_embed_dim, cfg.resnet_embed_dim
        # dominant kernel of the instrumented pass.  stream / gemm_planes / gemm_x3 are single kernels
        # (k_entity_stream, k_gemm_x3_planes, k_gemm_bf16x3); "gemm" is k_gemm_f32
        dom = max(prof, key=lambda k: prof[k][0])
        ms, launches = prof[dom]
        per_launch_ms = ms / max(launches, 1)
        fused = not args.generic and cfg.num_gcn_layers == 2
        x3 = args.precision in ("bf16x3", "bf16")
        flops_pair = path_flops_per_pair(D, R, cfg.num_gcn_layers, cfg.gcn_edge_type == "dynamic", fused)
        bytes_pair = algorithmic_bytes_per_pair(cfg, batch)
        cached = args.workload == "table" and args.entity_cache
        stream_bytes_pair = bytes_pair
        if cached:
            # k_cached_pairs: one gathered cache row in (h_t, h_i, c^, o^, sg and, with dynamic edges, fv_t, fv_i),
            # the candidate index and two similarities; et' planes and the four layer-2 edges out
            row = ((5 if cfg.gcn_edge_type == "dynamic" else 3) * D + R + 4) * 4
            stream_bytes_pair = row + 8 + 8 + D * 4 + 16
            flops_pair = 2.0 * D * D
        kernel_names = {"stream": "k_cached_pairs" if cached else "k_entity_stream", "gemm_planes": "k_gemm_x3_planes", "gemm_x3": "k_gemm_bf16x3",
                        "gemm": "k_gemm_f32"}
        if dom == "stream":
            # algorithmic = compulsory input bytes of the step; traffic = PMC-measured HBM bytes
            work = stream_bytes_pair * pairs_per_step * args.steps / max(launches, 1)
            achieved = work / (per_launch_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": kernel_names[dom], "achieved": achieved, "peak": PEAK_HBM_GBS,
                    "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS,
                    "traffic": (measured_traffic("k_cached_pairs", B, args.precision, fused, args.features, "kernels_table_cache")
                                if cached else (measured_traffic("k_entity_stream", B, args.precision, fused, args.features)
                                                if args.workload == "wikimel" else None)),
                    "launches": int(launches), "avg_launch_ms": per_launch_ms}
        else:
            # algorithmic FLOPs the kernel's launches cover in one step / their summed time.  In split-bf16
            # precision each algorithmic multiply-add is three bf16 MFMAs: `executed_*` is the matrix-core rate
            if dom == "gemm_planes":
                step_flops = pairs_per_step * (1 if cached else 2) * (2.0 * D * D)   # (x_t C_t^T and) et' W_h2^T
                if not cached and (args.workload == "table" or args.features == "bf16"):
                    step_flops += pairs_per_step * 2.0 * R * D   # x_i C_i^T runs on this kernel too (gathered / bf16 image planes)
            elif dom == "gemm_x3" and fused:
                # x_i C_i^T plus the mention-sized products, which run on the same kernel from 256 rows up
                step_flops = pairs_per_step * 2.0 * R * D + (mention_flops(D, R, fused) * B if (x3 and B >= 256) else 0.0)
            else:
                step_flops = flops_pair * pairs_per_step + mention_flops(D, R, fused) * B
            achieved = step_flops * args.steps / (ms * 1e-3) / 1e12
            peak = PEAK_F32_MATRIX_TFLOPS if dom == "gemm" else PEAK_BF16_MFMA_TFLOPS
            roof = {"bound": "mfma", "kernel": kernel_names.get(dom, dom), "achieved": achieved, "peak": peak,
                    "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None, "launches": int(launches),
                    "avg_launch_ms": per_launch_ms}
            if dom != "gemm":
                passes = 1 if args.precision == "bf16" else 3
                roof["executed_bf16_tflops"] = passes * achieved
                roof["executed_frac"] = passes * achieved / peak
        value = pairs_per_step * world * args.steps / elapsed
        line = {
            "metric": "mention x candidate pairs scored/sec",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"{cfg.dataset_name}-shaped scoring forward: {N - 1}-cand (+1 answer slot), D={D}, R={R}, "
                                   f"L={cfg.max_mention_sentence_len}, P={cfg.resnet_num_region}"
                                   + (f", T={cfg.max_entity_attr_token_len} token-level entity text" if cfg.token_level_entities else ""),
                       "mentions_per_step_per_gpu": B, "pairs_per_step": pairs_per_step * world,
                       "parallelism": f"dp{world} (mentions sharded, no collective)"},
            "roofline": roof,
            "kernel_ms_per_step": {k: v[0] / args.steps for k, v in prof.items()},
            "hbm_fraction_whole_path": bytes_pair * value / world / (PEAK_HBM_GBS * 1e9),
            # matrix-core work the path executes (split-bf16: three bf16 MFMA passes per algorithmic product) and the
            # reference-faithful operation count at the same rate, both against the dense bf16 peak
            "mfma_fraction_whole_path": ((1 if args.precision == "bf16" else 3) if x3 else 1) * flops_pair * value / world
                                        / ((PEAK_BF16_MFMA_TFLOPS if x3 else PEAK_F32_MATRIX_TFLOPS) * 1e12),
            "mfma_fraction_reference_flops": (2.0 * D * D + 2.0 * R * D + cfg.num_gcn_layers * 8.0 * D * D) * value / world
                                             / ((PEAK_BF16_MFMA_TFLOPS if x3 else PEAK_F32_MATRIX_TFLOPS) * 1e12),
            "launch": "hipGraph replay" if args.graph else "eager",
            "path": ("per-entity cache + layer 2" if cached else "fused two-layer" if fused else "layer-by-layer") + ", " + args.precision
                    + (", features stored as bf16" if args.features == "bf16" else ""),
            "algorithmic": {"bytes_per_pair": bytes_pair, "dominant_kernel_bytes_per_pair": stream_bytes_pair, "flops_per_pair_executed": flops_pair,
                            "flops_per_pair_reference": 2.0 * D * D + 2.0 * R * D + cfg.num_gcn_layers * 8.0 * D * D},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"], line["parity"] = cpu_baseline(cfg, sd, model=model if args.workload != "table" else None, dev=dev,
                                                                   features=args.features)
            if line["parity"] is None:
                del line["parity"]
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
