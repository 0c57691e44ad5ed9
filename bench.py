"""bench.py - mention x candidate pairs scored per second on WikiMEL-shaped synthetic batches.

    python bench.py [--gpus N --steps K --warmup W] [--workload wikimel|wikidiverse|table] [--mode score|train] ...

One "step" = one pass of the DRIN scoring path (Model.forward, drin/model.py:164-209) over one batch of B mentions
x N candidates of seeded synthetic features that are already resident in HBM.

N > 1: started as a plain process (`python bench.py --gpus N ...`, WORLD_SIZE unset) this file spawns
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` of ITSELF before it
makes any GPU call and relays rank 0's JSON line and the children's exit code; started by torch.distributed.run it
is a rank.  Every rank scores its own shard of mentions (weak scaling, no data-path collective - mentions are
independent); the timed region is bracketed by barrier + synchronize and the max over ranks is used.

Prints ONE JSON line (rank 0) of at most 4 KB, strict JSON: the contract fields of the headline (BASELINE.json's metric on its WikiMEL-100
config) plus
  roofline     : dominant kernel of the headline, HIP-event timed inside this process on the launch stream
  cpu_baseline : the CPU oracle (oracle/drin_oracle.py) timed on this host's cores on a bounded sample (N = 1)
  parity       : slices of the TIMED batch re-scored by the oracle (N = 1)
  legs         : value / ms_per_step / roofline fraction / parity error of every other leg, <= 150 bytes each
  legs_file    : where the FULL record went (--legs-file, default bench_legs.json next to this file): every leg with its own roofline,
                 the scaling model, step floors and provenance sentences - the other BASELINE configs under the same clock: f32_exact,
                 wikimel_mixed_f16 (only the entity-image contraction in one fp16 pass), wikimel_bf16_features (the headline batch with
                 bf16-stored features), wikidiverse_b4 (configs[0]: the reference's CPU-runnable case), train_step (configs 3 / 4; the
                 only leg that also runs at N > 1: one RCCL all-reduce of the flat gradient bucket per step), train_b512 (the step at
                 a rate-bound batch, gathered and table form), wikidiverse (config 2, fp32- and bf16-stored features), table_cache
                 (config 5: 1 M-entity table, 1000 candidates gathered on the device, mention chunks streamed)
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from drin_amd import synth  # noqa: E402
from drin_amd.config import DrinConfig, wikimel_config  # noqa: E402

PEAK_F32_MATRIX_TFLOPS = 157.3   # MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA" (dense)
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md "HBM3E peak BW" (spec)
def _latest(name):
    """The newest round's committed counter file (profiles/rN_<name>, highest N), as a repo-relative path."""
    import glob
    import re
    here = os.path.dirname(os.path.abspath(__file__))
    best = None
    for path in glob.glob(os.path.join(here, "profiles", f"r*_{name}")):
        m = re.fullmatch(rf"r(\d+)_{re.escape(name)}", os.path.basename(path))
        if m and (best is None or int(m.group(1)) > best):
            best = int(m.group(1))
    return os.path.join("profiles", f"r{best}_{name}") if best is not None else os.path.join("profiles", f"r0_{name}")


TRAFFIC_FILE = _latest("hbm_traffic.json")
MFMA_PMC_FILE = _latest("mfma_pmc.json")
XGMI_LINKS, XGMI_GBS_PER_LINK = 7, 153.0   # MI355X_MICROARCH.md / SURVEY.md section 5: point-to-point links per GPU
FEAT_SLOTS = (0, 4, 5, 7, 9, 10)  # the six feature tensors of the 14-sequence


# =====================================================================================================================
# launcher: N > 1 from a plain process
# =====================================================================================================================
def launch_children(args) -> int:
    """`python bench.py --gpus N` without torch.distributed.run around it: start the N ranks as child processes of a
    torch.distributed.run we spawn (this process has made no GPU call and makes none), relay rank 0's JSON line, return the
    children's exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env, cwd=REPO)
    lines = []
    for ln in proc.stdout:                                 # children print little: rank 0's one JSON line
        if ln.lstrip().startswith("{"):
            lines.append(ln.rstrip("\n"))
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    for ln in lines:
        print(ln, flush=True)
    if rc == 0 and not lines:
        sys.stderr.write("bench.py: the ranks exited 0 without printing a JSON line\n")
        rc = 1
    return rc


class Ctx:
    """Rank bookkeeping + the contract's timing bracket."""

    def __init__(self, args):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.stub = args.stub
        self.force = bool(getattr(args, "force_collective", False))
        self.pg_error = None
        if self.world != args.gpus:
            raise SystemExit(f"bench.py: WORLD_SIZE={self.world} but --gpus {args.gpus}: start N ranks with "
                             f"`python bench.py --gpus N` (self-launching) or torch.distributed.run --nproc-per-node N ... --gpus N")
        # rehearsal knobs for a ONE-GPU box (tools/dp_rehearsal.sh): all ranks on device 0 and gloo instead of RCCL (which
        # refuses two ranks on one device) - everything but the collective library itself is then the N > 1 code path
        share = os.environ.get("DRIN_BENCH_SHARE_GPU") == "1"
        backend = os.environ.get("DRIN_BENCH_BACKEND", "nccl")
        if self.stub:
            self.dev = torch.device("cpu")
        else:
            self.dev = torch.device("cuda", 0 if share else self.local_rank)
            torch.cuda.set_device(self.dev)
        self.backend = "gloo" if (self.stub or backend == "gloo") else "nccl"
        if self.world > 1 or self.force:
            self.init_group()

    def init_group(self):
        """The process group: always at N > 1; at N = 1 only for --force-collective / the `train_step_rccl` leg (a world of
        ONE rank - what a one-GPU box can run of the RCCL path).  Idempotent; returns whether a group is up."""
        import torch.distributed as dist
        if dist.is_initialized():
            return True
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if self.world == 1 and "MASTER_PORT" not in os.environ:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        import datetime
        limit = datetime.timedelta(seconds=300)              # a collective that never completes fails the run instead of hanging it
        try:
            if self.backend == "gloo":
                dist.init_process_group("gloo", timeout=limit, rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group("nccl", device_id=self.dev, timeout=limit, rank=self.rank, world_size=self.world)
        except Exception as e:  # noqa: BLE001 - at N = 1 the group is an extra: reported, never fatal; at N > 1 it is the run
            if self.world > 1:
                raise
            self.pg_error = f"{type(e).__name__}: {e}"
            return False
        return True

    def sync(self):
        if not self.stub:
            torch.cuda.synchronize(self.dev)

    def barrier(self):
        self.sync()
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()
        self.sync()

    def timed(self, run, steps, warmup):
        """W untimed steps, then EXACTLY K steps between barrier + synchronize pairs; returns (max over ranks of the
        bracketed seconds, [per-rank seconds until the rank's own work had drained], last result)."""
        out = None
        for _ in range(warmup):
            out = run()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = run()
        self.sync()
        local = time.perf_counter() - t0
        self.barrier()
        elapsed = time.perf_counter() - t0
        if self.world > 1:
            import torch.distributed as dist
            t = torch.tensor([elapsed, local], dtype=torch.float64, device=self.dev)
            parts = [torch.empty_like(t) for _ in range(self.world)]
            dist.all_gather(parts, t)
            elapsed = max(float(p[0]) for p in parts)
            per_rank = [float(p[1]) for p in parts]
        else:
            per_rank = [local]
        return elapsed, per_rank, out

    def finish(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()


# =====================================================================================================================
# algorithmic work
# =====================================================================================================================
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


def algorithmic_bytes(cfg, batch):
    """Compulsory input bytes per pair, split by who reads them:
      entity  - the entity-side rows one pair needs: pooled tokens 1..ntok-2 plus the CLS row (token form) or the pooled
                row, the mask, image and object rows, object scores, the two similarities (+ the 8-byte candidate index in
                table form);
      mention_stream - mention-side rows the entity pass itself reads once per mention (object rows + scores), / N;
      mention_pool   - mention-side rows the pooling kernels read (span rows, image regions), / N.
    whole path = entity + mention_stream + mention_pool + the 4-byte score."""
    D, R, N = cfg.bert_embed_dim, cfg.resnet_embed_dim, cfg.num_candidates_model
    Km, Ke = cfg.object_topk_mention, cfg.object_topk_entity
    if not isinstance(batch, (list, tuple)):              # table form (IndexedBatch): gathered pooled rows
        es = batch.table.text.element_size()
        ent = 8 + D * es + R * es + Ke * R * es + Ke * 4 + 8
        span, mes = 3.0, batch.mention[0].element_size()
    else:
        es = mes = batch[0].element_size()                # feature storage: 4 (fp32) or 2 (bf16) bytes
        if cfg.token_level_entities:
            ntok = batch[8].sum(-1).float()
            rows = (ntok - 2).clamp(min=0) + 1            # pooled tokens 1..ntok-2 plus the CLS row
            ent_text = float(rows.mean()) * D * es + batch[8].shape[-1] * 8
        else:
            ent_text = D * es
        ent = ent_text + R * es + Ke * R * es + Ke * 4 + 8
        span = float((batch[3] - batch[2]).float().mean())
    men_stream = (Km * R * mes + Km * 4) / N
    men_pool = (span * D * mes + cfg.resnet_num_region * R * mes + 16) / N
    return {"entity": ent, "mention_stream": men_stream, "mention_pool": men_pool,
            "whole_path": ent + men_stream + men_pool + 4}


def measured_traffic(kernel_prefix, section, matches):
    """HBM bytes per launch of a kernel from the committed PMC passes (tools/collect_pmc.py) - NOT re-measured in this
    run: valid only for the configuration the passes were taken on (`matches`), None otherwise."""
    path = os.path.join(REPO, TRAFFIC_FILE)
    if not (matches and os.path.exists(path)):
        return None
    try:
        for k, v in json.load(open(path)).get(section, {}).items():
            if kernel_prefix in k:
                return v["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def measured_whole_path_traffic(section, matches):
    """HBM bytes of ALL kernels of one scoring call (launches per call x bytes per launch, summed; tools/collect_pmc.py
    `whole_path`) from the committed PMC passes of this configuration - replayed, not re-measured; None without a pass."""
    path = os.path.join(REPO, TRAFFIC_FILE)
    if not (matches and os.path.exists(path)):
        return None
    try:
        return json.load(open(path))["whole_path"][section]["hbm_bytes_per_call"]
    except (OSError, KeyError, ValueError, TypeError):
        return None


def measured_mfma_busy(section, family):
    """Fraction of the kernel family's SIMD-cycles with a matrix instruction executing, from the committed SQ counter pass of
    this configuration (tools/pmc_mfma.sh -> tools/collect_mfma_pmc.py: SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES)) -
    replayed, NOT re-measured in this run; None when there is no pass for the configuration."""
    path = os.path.join(REPO, MFMA_PMC_FILE)
    if not (section and os.path.exists(path)):
        return None, None
    try:
        sec = json.load(open(path))["sections"][section]
        v = sec["families"][family]["mfma_busy"]
        return v, (f"{MFMA_PMC_FILE} [{section}].families.{family}: rocprofv3 --pmc SQ pass of `{sec['command']}`, replayed - not re-measured in this run")
    except (OSError, KeyError, ValueError, TypeError):
        return None, None


KERNEL_NAMES = {"stream": "k_entity_stream", "gemm_planes": "k_gemm_x3_planes", "gemm_x3": "k_gemm_bf16x3", "gemm": "k_gemm_f32",
                "gcn": "row kernels (k_pair_layer1, k_pair_final, ...)", "pool": "pooling kernels", "edge": "edge kernels"}


def score_roofline(cfg, batch, B, prof, steps, precision, features, workload, cached, fused, cache_format="f32"):
    """Roofline object of the dominant kernel class of an instrumented pass + the whole-path fractions' inputs."""
    D, R, N = cfg.bert_embed_dim, cfg.resnet_embed_dim, cfg.num_candidates_model
    pairs = B * N
    dom = max(prof, key=lambda k: prof[k][0])
    ms, launches = prof[dom]
    per_launch_ms = ms / max(launches, 1)
    x3 = precision in ("bf16x3", "bf16x3_if16")
    flops_pair = path_flops_per_pair(D, R, cfg.num_gcn_layers, cfg.gcn_edge_type == "dynamic", fused)
    ab = algorithmic_bytes(cfg, batch)
    stream_bytes_pair = ab["entity"] + ab["mention_stream"]          # what k_entity_stream itself has to read
    if cached:
        # k_cached_pairs: one gathered cache row in (h_t, h_i, c^, o^, sg and, with dynamic edges, fv_t, fv_i),
        # the candidate index and two similarities; et' planes and the four layer-2 edges out
        dyn = cfg.gcn_edge_type == "dynamic"
        row = ((5 if dyn else 3) * D + R + 4) * 4
        if cache_format == "mixed_f16":    # h_t, h_i, c^ fp32; o^ (and fv_t, fv_i) scaled fp16; sg + the three scales (drin_hip.h)
            row = 3 * D * 4 + ((2 if dyn else 0) * D + R) * 2 + 16
        stream_bytes_pair = row + 8 + 8 + D * 4 + 16
        flops_pair = 2.0 * D * D
    names = dict(KERNEL_NAMES)
    if cached:
        names["stream"] = "k_cached_pairs"
    if dom == "stream":
        work = stream_bytes_pair * pairs * steps / max(launches, 1)
        achieved = work / (per_launch_ms * 1e-3) / 1e9
        default_cfg = B == 4096 and precision == "bf16x3" and fused and features == "f32" and workload == "wikimel"
        traffic = (measured_traffic("k_cached_pairs", "kernels_table_cache_mixed_f16" if cache_format == "mixed_f16" else "kernels_table_cache",
                                    B == 4096 and features == "f32") if cached
                   else measured_traffic("k_entity_stream", "kernels", default_cfg))
        roof = {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": achieved / PEAK_HBM_GBS, "traffic": traffic,
                # `traffic` is NOT measured in this run: PMC passes need their own rocprofv3 invocation
                "traffic_replayed": bool(traffic), "traffic_file": (TRAFFIC_FILE if traffic else None),
                "traffic_source": (f"{TRAFFIC_FILE}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this configuration "
                                   "(tools/collect_pmc.py), replayed - not re-measured in this run") if traffic else None,
                "launches": int(launches), "avg_launch_ms": per_launch_ms, "algorithmic_bytes_per_launch": work}
    else:
        # algorithmic FLOPs the kernel's launches cover in one step / their summed time.  In split-bf16
        # precision each algorithmic multiply-add is three bf16 MFMAs: `executed_*` is the matrix-core rate
        if dom == "gemm_planes":
            step_flops = pairs * (1 if cached else 2) * (2.0 * D * D)   # (x_t C_t^T and) et' W_h2^T
            if not cached and (workload == "table" or features == "bf16"):
                step_flops += pairs * 2.0 * R * D   # x_i C_i^T runs on this kernel too (gathered / bf16 image planes)
        elif dom == "gemm_x3" and fused:
            # x_i C_i^T plus the mention-sized products, which run on the same kernel from 256 rows up
            step_flops = pairs * 2.0 * R * D + (mention_flops(D, R, fused) * B if (x3 and B >= 256) else 0.0)
        else:
            step_flops = flops_pair * pairs + mention_flops(D, R, fused) * B
        achieved = step_flops * steps / (ms * 1e-3) / 1e12
        peak = PEAK_F32_MATRIX_TFLOPS if dom == "gemm" else PEAK_BF16_MFMA_TFLOPS
        roof = {"bound": "mfma", "kernel": names.get(dom, dom), "achieved": achieved, "peak": peak,
                "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None, "launches": int(launches),
                "avg_launch_ms": per_launch_ms}
        if dom != "gemm":
            passes = 1 if (dom == "gemm_x3" and if16_taken(precision, cfg, workload, B, features)) else 3
            roof["executed_bf16_tflops"] = passes * achieved
            roof["executed_frac"] = passes * achieved / peak
            section = None
            if precision == "bf16x3" and features == "f32" and fused and not cached:
                section = {("wikimel", 4096): "wikimel_b4096", ("wikidiverse", 16384): "wikidiverse_b16384"}.get((workload, B))
            roof["mfma_busy"], roof["mfma_busy_source"] = measured_mfma_busy(section, {"gemm_x3": "k_gemm_bf16x3", "gemm_planes": "k_gemm_x3_planes"}.get(dom, ""))
    return roof, ab, stream_bytes_pair, flops_pair


def if16_taken(precision, cfg, workload, B, features="f32"):
    """Whether a call takes the one-pass fp16 image contraction: asked of the LIBRARY (`drin_image_contraction_passes`, the gate
    `csrc/fused_forward.hip` itself applies - N >= 64, the exact widths, per-pair fp32-stored rows, at least 128 tiles of
    256 x 256): a `bf16x3_if16` line of any other shape ran split-bf16 and says so."""
    if precision != "bf16x3_if16":
        return False
    import ctypes as C
    from drin_amd import _lib
    c = _lib.DrinConfigC()
    _lib.check(_lib.load().drin_default_config(C.byref(c)))
    c.batch, c.num_candidates, c.embed_dim, c.image_dim = B, cfg.num_candidates_model, cfg.bert_embed_dim, cfg.resnet_embed_dim
    c.entity_tokens = cfg.max_entity_attr_token_len if cfg.token_level_entities else 0
    c.precision = _lib.PREC_BF16X3_IF16
    c.feature_dtype = _lib.FEAT_BF16 if features == "bf16" else _lib.FEAT_F32
    return _lib.load().drin_image_contraction_passes(C.byref(c), 1 if workload == "table" else 0) == 1


def whole_path_fractions(cfg, ab, flops_pair, rate_per_gpu, precision, workload="wikimel", B=1 << 20, features="f32"):
    D, R = cfg.bert_embed_dim, cfg.resnet_embed_dim
    x3 = precision in ("bf16x3", "bf16x3_if16")
    peak = (PEAK_BF16_MFMA_TFLOPS if x3 else PEAK_F32_MATRIX_TFLOPS) * 1e12
    ref_flops = 2.0 * D * D + 2.0 * R * D + cfg.num_gcn_layers * 8.0 * D * D
    executed = (3 if x3 else 1) * flops_pair
    if if16_taken(precision, cfg, workload, B, features):
        executed -= 2 * 2.0 * R * D                       # the image contraction in one pass instead of three
    return {
        "hbm_fraction_whole_path": ab["whole_path"] * rate_per_gpu / (PEAK_HBM_GBS * 1e9),
        # matrix-core work the path executes (split-bf16: three bf16 MFMA passes per algorithmic product) and the
        # reference-faithful operation count at the same rate, both against the dense peak of the arithmetic type
        "mfma_fraction_whole_path": executed * rate_per_gpu / peak,
        "mfma_fraction_reference_flops": ref_flops * rate_per_gpu / peak,
    }, ref_flops


# =====================================================================================================================
# host baseline + parity
# =====================================================================================================================
def host_threads():
    """Threads the CPU baseline uses and where the number comes from: the cores this process may run on
    (affinity intersected with the cgroup quota), unless DRIN_CPU_THREADS overrides it - then that is stated."""
    info = {"os_cpu_count": os.cpu_count() or 1}
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else info["os_cpu_count"]
    info["affinity"] = n
    info["cgroup_quota_cpus"] = None
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            info["cgroup_quota_cpus"] = int(quota) / int(period)
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    info["capped_by"] = None
    env = os.environ.get("DRIN_CPU_THREADS")
    if env:
        n, info["capped_by"] = max(1, min(n, int(env))), "DRIN_CPU_THREADS"
    info["threads_used"] = max(1, n)
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                info["cpu_model"] = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return info


def _time_oracle(fn, seconds, max_iter):
    fn()                                                   # warm-up
    t0 = time.perf_counter()
    it = 0
    while True:
        fn()
        it += 1
        el = time.perf_counter() - t0
        if el >= seconds or it >= max_iter:
            return it, el


def cpu_baseline(cfg, sd, seconds=10.0):
    """The CPU oracle (the reference restated, pinned by tests/golden) timed on this host's cores on bounded samples:
    the headline shape at B=8 (20 MB of fp32 tokens per mention make larger host batches pointless), the same with the
    reference's own Python loops kept, and the WikiDiverse B=64 sample SURVEY.md 8d names."""
    from oracle import drin_oracle as O

    host = host_threads()
    torch.set_num_threads(host["threads_used"])
    B = 8 if cfg.token_level_entities else 64
    batch = synth.make_batch(cfg, B, 3)
    N = cfg.num_candidates_model
    with torch.no_grad():
        it, el = _time_oracle(lambda: O.forward(sd, batch), seconds, 2000)
        # the same arithmetic with the reference's own Python loops over mentions and candidates kept
        # (baselines/ghmfc.py:58-59,246-249; model.py:86-91): its cost profile, ~4 s sample
        it_r, el_r = _time_oracle(lambda: O.reference_style_forward(sd, batch), 4.0, 500)
        wd = DrinConfig()
        wd_sd = synth.make_state_dict(wd, 7)
        wd_batch = synth.make_batch(wd, 64, 3)
        it_w, el_w = _time_oracle(lambda: O.forward(wd_sd, wd_batch), 4.0, 2000)
        b64 = None
        if cfg.token_level_entities:
            # the headline shape at the B = 64 SURVEY.md 8d names (1.4 GB of host features): one warm-up + up to three forwards
            del batch
            big = synth.make_batch(cfg, 64, 4)
            it_b, el_b = _time_oracle(lambda: O.forward(sd, big), 3.0, 3)
            b64 = it_b * 64 * N / el_b
            del big
    return {"value": it * B * N / el, "unit": "pairs/s", "cores": host["threads_used"], "kind": "port",
            "host": host,
            "reference_style_loops_value": it_r * B * N / el_r,
            "wikidiverse_b64_value": it_w * 64 * wd.num_candidates_model / el_w,
            "headline_shape_b64_value": b64,
            "b64_value": b64,                      # SURVEY.md 8d's sample size (B = 64) beside `value` (B = 8: the CPU's better figure)
            "sample": f"{it} CPU-oracle forwards, {cfg.dataset_name}-shaped B={B} N={N} fp32, {torch.get_num_threads()} torch threads, {el:.1f} s",
            "other_samples": f"reference-style loops {it_r} forwards {el_r:.1f} s; wikidiverse-shaped B=64 N={wd.num_candidates_model}: "
                             f"{it_w} forwards {el_w:.1f} s"}


def slice_batch(batch, rows):
    """Mentions `rows` (a slice) of a 14-sequence or an IndexedBatch, as the equivalent CPU 14-sequence."""
    if isinstance(batch, (list, tuple)):
        return [t[rows].cpu() for t in batch[:14]]
    from drin_amd.model import IndexedBatch
    sub = IndexedBatch([t[rows] for t in batch.mention], batch.table, batch.candidates[rows],
                       batch.miet_similarity[rows], batch.mtei_similarity[rows])
    return [t.cpu() for t in sub.gathered()]


def parity_of_timed_batch(cfg, sd, batch, out, n_slices=8, width=8, fp32_batch=None):
    """Slices of the TIMED batch (mentions are independent: the timed output rows are the scores of those mentions) scored
    by the CPU oracle on the same values - the checker, never the thing measured."""
    from oracle import drin_oracle as O

    B = out.shape[0]
    width = min(width, B)
    starts = sorted({int(round(i * (B - width) / max(n_slices - 1, 1))) for i in range(n_slices)})
    err, agree, total, err32, agree32 = 0.0, 0, 0, 0.0, 0
    with torch.no_grad():
        for s in starts:
            rows = slice(s, s + width)
            host = [t.float() if t.dtype == torch.bfloat16 else t for t in slice_batch(batch, rows)]
            ref = O.forward(sd, host)
            got = out[rows].float().cpu()
            err = max(err, (got - ref).abs().max().item())
            agree += int((got[:, :-1].argmax(1) == ref[:, :-1].argmax(1)).sum())
            if fp32_batch is not None:               # what storing the features as bf16 costs against the fp32 inputs
                ref32 = O.forward(sd, slice_batch(fp32_batch, rows))
                err32 = max(err32, (got - ref32).abs().max().item())
                agree32 += int((got[:, :-1].argmax(1) == ref32[:, :-1].argmax(1)).sum())
            total += width
    par = {"max_abs_score_err": err, "top1_agreement": agree / total, "mentions": total,
           "against": "CPU oracle (pinned to the reference by tests/golden) on slices of the timed batch, same stored values"}
    if fp32_batch is not None:
        par["vs_fp32_features"] = {"max_abs_score_err": err32, "top1_agreement": agree32 / total}
    return par


# =====================================================================================================================
# scoring legs
# =====================================================================================================================
def make_model(cfg, sd, dev, precision, fused=True):
    from drin_amd.model import Model
    model = Model(cfg, precision=precision, fused=fused).to(dev).eval()
    model.load_state_dict(sd)
    return model


def run_score(ctx, model, batch, steps, warmup, graph=False):
    """Timed K steps of model(batch) + an instrumented pass (HIP events on the launch stream, per kernel class)."""
    from drin_amd import _lib
    with torch.no_grad():
        run = lambda: model(batch)  # noqa: E731
        if graph:
            # small batches are a chain of ~25 short launches: the library allocates nothing and never synchronises,
            # so the whole scoring call replays as one hipGraph (the weights' folded products are built before capture)
            for _ in range(3):
                model(batch)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                model(batch)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                static_out = model(batch)

            def run():
                g.replay()
                return static_out
        elapsed, per_rank, out = ctx.timed(run, steps, warmup)
        assert torch.isfinite(out).all()
        _lib.profile_begin(1 << 16)
        for _ in range(steps):
            model(batch)
        prof = _lib.profile_end()
    return elapsed, per_rank, out, prof


def score_line(ctx, cfg, args, B, batch, elapsed, per_rank, prof, steps, warmup, precision, features, workload, cached, fused, graph=False,
               cache_format=None):
    D, R, N = cfg.bert_embed_dim, cfg.resnet_embed_dim, cfg.num_candidates_model
    pairs = B * N
    cache_format = cache_format or getattr(args, "cache_format", "f32")
    roof, ab, stream_bytes_pair, flops_pair = score_roofline(cfg, batch, B, prof, steps, precision, features, workload, cached, fused, cache_format)
    value = pairs * ctx.world * steps / elapsed
    fr, ref_flops = whole_path_fractions(cfg, ab, flops_pair, value / ctx.world, precision, workload, B, features)
    # the one HBM-bound pass over the entity bytes, whichever class is the longest of this leg
    s_ms, s_n = prof.get("stream", (0.0, 0))
    hbm_kernel = None
    if s_n:
        s_bytes = stream_bytes_pair * pairs * steps / s_n
        hbm_kernel = {"kernel": "k_cached_pairs" if cached else "k_entity_stream", "avg_launch_ms": s_ms / s_n,
                      "algorithmic_bytes_per_launch": s_bytes, "achieved": s_bytes / (s_ms / s_n * 1e-3) / 1e9, "unit": "GB/s",
                      "frac": s_bytes / (s_ms / s_n * 1e-3) / 1e9 / PEAK_HBM_GBS}
    default_cfg = B == 4096 and precision == "bf16x3" and fused and features == "f32" and workload == "wikimel"
    whole_traffic = (measured_whole_path_traffic("kernels_table_cache_mixed_f16" if cache_format == "mixed_f16" else "kernels_table_cache",
                                                 B == 4096 and features == "f32") if cached
                     else measured_whole_path_traffic("kernels", default_cfg))
    line = {
        "metric": "mention x candidate pairs scored/sec",
        "value": value, "unit": "pairs/s", "n_gpus": ctx.world, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": precision, "data": "synthetic",
        "config": {"workload": f"{cfg.dataset_name}-shaped scoring forward: {N - 1}-cand (+1 answer slot), D={D}, R={R}, "
                               f"L={cfg.max_mention_sentence_len}, P={cfg.resnet_num_region}"
                               + (f", T={cfg.max_entity_attr_token_len} token-level entity text" if cfg.token_level_entities else "")
                               + (", candidates gathered on the device from an entity table" if workload == "table" else ""),
                   "mentions_per_step_per_gpu": B, "pairs_per_step": pairs * ctx.world,
                   "parallelism": f"dp{ctx.world} (mentions sharded, no collective)"},
        "rank_ms_per_step": [t / steps * 1e3 for t in per_rank],
        "roofline": roof,
        "hbm_kernel": hbm_kernel,
        "kernel_ms_per_step": {k: v[0] / steps for k, v in prof.items()},
        **fr,
        # every kernel of the step, not the dominant one alone: HBM bytes per step from the committed PMC passes (replayed)
        # against the compulsory input bytes - intermediates that round-trip through HBM show up here
        "hbm_traffic_whole_path": whole_traffic,
        # (compulsory bytes of THIS path: through the per-entity cache a pair needs its cache row, not its raw feature rows)
        "hbm_traffic_whole_path_over_algorithmic": (whole_traffic / (((stream_bytes_pair + ab["mention_pool"] + 4) if cached
                                                                      else ab["whole_path"]) * pairs)) if whole_traffic else None,
        "hbm_traffic_whole_path_source": (f"{TRAFFIC_FILE} whole_path: launches per call x FETCH_SIZE / WRITE_SIZE bytes per launch of every "
                                          "kernel of the call (tools/collect_pmc.py), replayed - not re-measured in this run") if whole_traffic else None,
        "launch": "hipGraph replay" if graph else "eager",
        "path": ((f"per-entity cache ({cache_format} rows) + layer 2") if cached else "fused two-layer" if fused else "layer-by-layer") + ", "
                + ("bf16x3 (the fp16 image contraction is not taken at this shape)" if (precision == "bf16x3_if16" and not if16_taken(precision, cfg, workload, B, features))
                   else precision) + (", features stored as bf16" if features == "bf16" else ""),
        "algorithmic": {"bytes_per_pair": ab["whole_path"], "dominant_kernel_bytes_per_pair": stream_bytes_pair,
                        "bytes_per_pair_split": ab, "flops_per_pair_executed": flops_pair, "flops_per_pair_reference": ref_flops},
    }
    return line


def compact(line, keep=("value", "unit", "ms_per_step", "steps", "dtype", "path", "roofline", "hbm_kernel", "kernel_ms_per_step",
                        "hbm_fraction_whole_path", "hbm_traffic_whole_path", "hbm_traffic_whole_path_over_algorithmic",
                        "mfma_fraction_whole_path", "parity")):
    """A secondary leg's entry of the one JSON line."""
    out = {"workload": line["config"]["workload"], "mentions_per_step": line["config"]["mentions_per_step_per_gpu"]}
    out.update({k: line[k] for k in keep if k in line})
    return out


def build_table(cfg, E, dev, seed=7):
    from drin_amd.model import EntityTable
    D, R = cfg.bert_embed_dim, cfg.resnet_embed_dim
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    return EntityTable(torch.randn(E, D, device=dev, generator=g), None, torch.randn(E, R, device=dev, generator=g),
                       torch.randn(E, 1, R, device=dev, generator=g), torch.rand(E, 1, device=dev, generator=g)), g


def make_table_chunk(cfg, table, B, seed, dev, g):
    from drin_amd.model import IndexedBatch
    N, E = cfg.num_candidates_model, table.num_entities
    men = synth.make_device_batch(cfg.with_(num_candidates_data=0), B, seed, dev)
    cand = torch.randint(0, E, (B, N), device=dev, generator=g)
    sims = 20.0 + 5.0 * torch.randn(2, B, N, device=dev, generator=g)
    return IndexedBatch(men[:7], table, cand, sims[0], sims[1])


def stream_table(ctx, model, cfg, table, g, mentions, chunk, seed0=1000):
    """BASELINE config 5 as SURVEY.md 8d words it: `mentions` mentions x N candidates streamed through the scoring path in
    chunks of `chunk` mentions; each chunk's mention side and candidate rows are drawn on the device right before it is
    scored (1.1 ms of generator time against 41 ms of scoring per 4096-mention chunk - inside the timed region; drawing the
    next chunk on a second stream was measured and dropped: nothing to hide, and cross-stream blocks made the caching
    allocator grow by 1 GB per chunk).  Only the current chunk is resident.  Returns (seconds, chunks, last scores, last chunk)."""
    n_chunks = (mentions + chunk - 1) // chunk
    with torch.no_grad():
        # untimed: two chunks' worth of blocks into the caching allocator (the loop holds the previous chunk while it draws the
        # next one); a first hipMalloc of 3.4 GB costs a few hundred ms and would otherwise sit in the first timed steps
        warm = [make_table_chunk(cfg, table, min(chunk, mentions), seed0 - 1 - i, ctx.dev, g) for i in range(2)]
        model(warm[0])
        del warm
        ctx.barrier()
        t0 = time.perf_counter()
        out = ib = None
        for c in range(n_chunks):
            ib = make_table_chunk(cfg, table, min(chunk, mentions - c * chunk), seed0 + c, ctx.dev, g)
            out = model(ib)
        ctx.barrier()
        return time.perf_counter() - t0, n_chunks, out, ib


# =====================================================================================================================
# training leg (BASELINE configs 3 / 4)
# =====================================================================================================================
def train_roofline(cfg, B, N, prof, steps, precision):
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
    tf = (fwd + bwd) * B * N * steps / (ms * 1e-3) / 1e12
    x3 = precision == "bf16x3"
    peak = PEAK_BF16_MFMA_TFLOPS if x3 else PEAK_F32_MATRIX_TFLOPS
    out = {"bound": "mfma", "kernel": "k_gemm_bf16x3 + k_gemm_tn_bf16x3", "achieved": tf, "peak": peak, "unit": "TFLOP/s",
           "frac": tf / peak, "traffic": None, "launches": int(launches), "avg_launch_ms": ms / max(launches, 1),
           "flops_per_pair": fwd + bwd}
    if x3:
        out["executed_bf16_tflops"], out["executed_frac"] = 3 * tf, 3 * tf / peak
        section = {64: "train_b64", 512: "train_b512"}.get(B) if (precision == "bf16x3" and cfg.token_level_entities) else None
        out["mfma_busy"], out["mfma_busy_source"] = measured_mfma_busy(section, "split_bf16_family")
    return out


def step_floor(cfg, B, N, bytes_per_pair, flops_per_pair, precision):
    """What a training step cannot go below on this machine, from two measured rates: its compulsory input bytes at the
    6.3 TB/s MI355X_MICROARCH.md measures for streaming reads, and its executed matrix-core FLOPs (three bf16 passes per
    product in split-bf16 precision) at the 1.3 PF/s this library's own K-loops sustain.  `ms` = the
    larger of the two (they could overlap at best); `serial_ms` = their sum (what back-to-back kernels could reach)."""
    passes = 3 if precision in ("bf16x3",) else 1
    hbm_ms = bytes_per_pair * B * N / 6.3e12 * 1e3
    mfma_ms = passes * flops_per_pair * B * N / (1.3e15 if passes == 3 else 0.82 * PEAK_F32_MATRIX_TFLOPS * 1e12) * 1e3
    return {"ms": max(hbm_ms, mfma_ms), "serial_ms": hbm_ms + mfma_ms, "hbm_ms": hbm_ms, "mfma_ms": mfma_ms,
            "rates": "6.3 TB/s streaming reads (MI355X_MICROARCH.md); 1.3 PF/s executed bf16 (this library's K-loop, DESIGN.md 4.2)"}


def scaling_model(step_ms, allreduce_bytes, hideable_ms, worlds=(2, 4, 8)):
    """What a data-parallel step should cost at N GPUs of one node, stated BEFORE anybody has measured it (no 8-GPU node has
    been available to this repository; the driver computes the measured efficiency from its own runs): the per-rank step
    without a collective (`step_ms`, measured here), one all-reduce of the flat gradient bucket over xGMI - a ring moves
    2 (N - 1) / N of the bucket through ONE link per hop; reduce-scatter + all-gather with every peer directly uses N - 1 of
    the 7 links at once - of which the parameter-free head of the next forward (`hideable_ms`: pooling + static edges,
    measured here) hides its length (drin_amd.train.OverlappedStep).  Mention shards are independent: nothing else couples."""
    out = {"step_ms_without_collective": step_ms, "allreduce_bytes": allreduce_bytes, "hideable_ms": hideable_ms,
           "xgmi": {"links_per_gpu": XGMI_LINKS, "GB_per_s_per_link": XGMI_GBS_PER_LINK}, "by_world": {}}
    for n in worlds:
        ring = 2.0 * (n - 1) / n * allreduce_bytes / (XGMI_GBS_PER_LINK * 1e9) * 1e3
        direct = 2.0 * (allreduce_bytes / n) / (XGMI_GBS_PER_LINK * 1e9) * 1e3       # each peer's share over its own link, both phases
        exposed = max(0.0, ring - hideable_ms)
        out["by_world"][str(n)] = {"ring_allreduce_ms": ring, "direct_allreduce_ms": direct, "expected_exposed_ms": exposed,
                                   "expected_step_ms": step_ms + exposed,
                                   "expected_weak_scaling_efficiency": step_ms / (step_ms + exposed)}
    return out


def gather_ranks(ctx, value):
    """[value of rank 0, rank 1, ...] (floats)."""
    if ctx.world == 1:
        return [float(value)]
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=ctx.dev)
    parts = [torch.empty_like(t) for _ in range(ctx.world)]
    dist.all_gather(parts, t)
    return [float(x[0]) for x in parts]


def bench_train(ctx, cfg, sd, B, steps, warmup, precision="bf16x3", features="f32", train_form="gathered", train_entities=50_000,
                graph=False, fused_adam=False, torch_loss=False, library_adam=True, force_collective=False):
    """One optimisation step of train.py:30-56 per "step": forward (intermediates kept), TripletLoss, backward
    through the HIP kernels, the RCCL all-reduce of the flat gradient bucket (world > 1, or a forced world of one), Adam.
    With collectives the all-reduce and Adam run on a side stream under the next step's parameter-free head
    (drin_amd.train.OverlappedStep; DRIN_OVERLAP selects another mode).  `allreduce_ms` = the one-piece collective when it
    runs serially after backward (HIP events), `allreduce_exposed_ms` = the timed step minus the same step without any
    collective: what the overlap leaves on the critical path."""
    from drin_amd import _lib
    from drin_amd.metrics import DeviceLossMetric
    from drin_amd.model import Model
    from drin_amd.train import GradBucket, make_adam

    dev, world, rank = ctx.dev, ctx.world, ctx.rank
    model = Model(cfg, precision=precision).to(dev)
    model.load_state_dict(sd)
    model.train()
    fdt = torch.bfloat16 if features == "bf16" else torch.float32
    full = synth.make_device_batch(cfg, B, 200 + rank, dev, dtype=fdt)
    batch, y = full[:14], full[14]
    ib = None
    if train_form != "gathered":
        # table form: the entity tables live on the device, a step carries candidate indices.
        # "table": the library pools every entity's tokens once and gathers pooled rows; "table-tokens": the token
        # blocks are gathered with torch indexing every step (what the reference's loader does on the host)
        from drin_amd.model import EntityTable, IndexedBatch
        E = train_entities
        tab = synth.make_device_batch(cfg.with_(num_candidates_data=E - 1), 1, 300 + rank, dev, dtype=fdt)
        table = EntityTable(tab[7][0], tab[8][0] if cfg.token_level_entities else None, tab[9][0], tab[10][0], tab[11][0])
        del tab
        g = torch.Generator(device=dev)
        g.manual_seed(17 + rank)
        cand = torch.randint(0, E, (B, cfg.num_candidates_model), device=dev, generator=g)
        ib = IndexedBatch(batch[:7], table, cand, batch[12], batch[13])
        full = None
        batch = ib if train_form == "table" else None
    loss_fn = DeviceLossMetric(cfg.triplet_margin, cfg.metrics_topk, dev)   # loss + top-k counters in one library call, as MELRunner
    if torch_loss:
        from drin_amd.metrics import TripletLoss
        loss_fn = TripletLoss(cfg.triplet_margin)
    opt = make_adam(model, cfg.learning_rate, library=library_adam and not graph and not fused_adam, capturable=graph, fused=fused_adam)
    import torch.distributed as dist
    from drin_amd.train import OverlappedStep
    collectives = world > 1 or (force_collective and dist.is_available() and dist.is_initialized())
    # how the all-reduce leaves the critical path (drin_amd/train.py): "forward" - collective + Adam on a side stream under
    # the next step's parameter-free head (default); "backward" - the GCN layers' piece started inside the staged backward
    mode = os.environ.get("DRIN_OVERLAP", "forward") if (collectives and not graph) else "none"
    if mode not in ("none", "forward", "backward", "both"):
        raise SystemExit(f"DRIN_OVERLAP={mode!r}: none | forward | backward | both")
    plain_bucket = GradBucket(list(model.parameters()), force=force_collective)
    staged = mode in ("backward", "both")
    bucket = GradBucket(list(model.parameters()), force=force_collective, overlap=True, model=model) if staged else plain_bucket
    pipe = OverlappedStep(model, bucket, opt) if mode in ("forward", "both") else None
    ar_events = []

    def eager_step(record=False, how="default"):
        """how: "default" (the configured overlap), "serial" (one-piece collective after backward, then Adam), "none" (no collective)"""
        opt.zero_grad(set_to_none=True)
        loss = loss_fn(y, model(batch if batch is not None else ib.gathered()))
        loss.backward()
        if how == "default" and pipe is not None:
            pipe.run()
            return loss
        use = bucket if how == "default" else plain_bucket
        if how == "none":
            pass
        elif record and collectives:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            use.allreduce_mean()
            e1.record()
            ar_events.append((e0, e1))
        else:
            use.allreduce_mean()
        opt.step()
        return loss

    step = eager_step
    if graph:
        # the library allocates nothing and never synchronises, so the step is capture-safe: ~100 launches
        # (forward, loss, backward, Adam) become one hipGraph replay and the host drops out of the loop
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                eager_step()
        torch.cuda.current_stream().wait_stream(side)
        cg = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(cg):
            static_loss = eager_step()

        def step():
            cg.replay()
            return static_loss

    elapsed, per_rank, loss = ctx.timed(step, steps, warmup)
    if pipe is not None:
        pipe.finish()
    serial_ms = none_ms = None
    hook = model._layers_ready_hook
    if collectives:
        # the same step (a) with the one-piece collective after backward and Adam behind it, (b) without any collective:
        # what the collective costs when nothing hides it, and what of it the configured overlap leaves exposed
        model._layers_ready_hook = None
        e_s, _pr, _l = ctx.timed(lambda: eager_step(how="serial"), steps, 5)
        e_n, _pr, _l = ctx.timed(lambda: eager_step(how="none"), steps, 5)
        serial_ms, none_ms = e_s / steps * 1e3, e_n / steps * 1e3
    _lib.profile_begin(1 << 16)
    for _ in range(steps):
        eager_step(record=True, how="serial")
    prof = _lib.profile_end()
    ctx.sync()
    model._layers_ready_hook = hook
    ar_ms = sum(a.elapsed_time(b) for a, b in ar_events) / max(len(ar_events), 1) if ar_events else 0.0
    exposed_ms = max(0.0, elapsed / steps * 1e3 - none_ms) if none_ms is not None else 0.0
    N = cfg.num_candidates_model
    ab = algorithmic_bytes(cfg, full[:14]) if full is not None else None
    roof = train_roofline(cfg, B, N, prof, steps, precision)
    floor = step_floor(cfg, B, N, ab["whole_path"], roof["flops_per_pair"], precision) if (ab and roof) else None
    coll = None
    if collectives:
        coll = {"backend": dist.get_backend(), "world": dist.get_world_size(), "forced_world_of_one": world == 1,
                "overlap": mode, "pieces": 2 if staged else 1, "in_place": bool(plain_bucket.in_place or bucket.in_place),
                "collectives_issued": bucket.collectives + plain_bucket.collectives * (plain_bucket is not bucket),
                "steps_overlapped_under_next_forward": pipe.steps if pipe is not None else 0,
                "steps_with_early_piece": bucket.overlapped,
                "serial_ms_per_step": serial_ms, "no_collective_ms_per_step": none_ms}
    bucket.close()
    step_ms = elapsed / steps * 1e3
    hideable = (prof.get("pool", (0.0, 0))[0] + prof.get("edge", (0.0, 0))[0]) / steps
    # (the loss kernels count under "edge" too: ~0.03 ms of the class belongs to the end of the step, not to the next forward's head)
    model_of_scaling = scaling_model(none_ms if none_ms is not None else step_ms, plain_bucket.nbytes(), max(0.0, hideable - 0.03))
    rank_gemm_ms = gather_ranks(ctx, roof["avg_launch_ms"] if roof else 0.0)
    if coll is not None:
        coll["rank_gemm_avg_launch_ms"] = rank_gemm_ms
    return {
        "metric": "mention x candidate pairs trained/sec (forward + backward + Adam)" + (" [hipGraph replay]" if graph else ""),
        "value": B * N * world * steps / elapsed, "unit": "pairs/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": precision, "data": "synthetic",
        "config": {"workload": f"{cfg.dataset_name}-shaped training step, {N - 1}-cand, per-GPU batch {B} (args.py:118)"
                               + (", features stored as bf16 (token blocks pooled in place, the rest widened)" if features == "bf16" else "")
                               + ({"gathered": "", "table": f", candidates indexed into a device-resident table of {train_entities} entities (tokens pooled once per entity)",
                                   "table-tokens": f", candidates gathered from a device-resident table of {train_entities} entities with torch indexing every step"}[train_form]),
                   "global_batch": B * world, "parallelism": f"dp{world}, one flat-bucket RCCL all-reduce per step"},
        "rank_ms_per_step": [t / steps * 1e3 for t in per_rank],
        "allreduce_ms": ar_ms, "allreduce_exposed_ms": exposed_ms, "allreduce_bytes": plain_bucket.nbytes(), "collective": coll,
        "scaling_model": model_of_scaling, "rank_roofline_avg_launch_ms": rank_gemm_ms,
        "step_floor_ms": floor["ms"] if floor else None, "step_floor": floor,
        "optimizer": opt.describe() if hasattr(opt, "describe") else type(opt).__name__,
        "library_launches_per_step": sum(v[1] for v in prof.values()) / steps,
        "kernel_ms_per_step": {k: v[0] / steps for k, v in prof.items()},
        "roofline": roof,
        "final_loss": float(loss.detach())}


def train_parity(dev, steps=10):
    """The checker of the train_step leg (outside every timed region): `steps` optimisation steps of `train.py:30-56` at the
    reference's width and batch (D = 768, R = 2048, N = 101, B = 64; T = 8: the token count is a free dimension, and eight
    tokens keep the CPU side at ~2 s per step) with the HIP `Model` in the leg's arithmetic (split-bf16 forward and backward
    contractions) + the one-launch Adam, against the CPU oracle's fp32 forward / autograd + torch.optim.Adam from the same
    seed-0 weights over the same learnable synthetic stream (`oracle/trajectory.py`; tests/test_gpu_round4.py runs 30 steps)."""
    from oracle.trajectory import trajectory
    torch.set_num_threads(host_threads()["threads_used"])
    cfg = wikimel_config(max_entity_attr_token_len=8, batch_size=64)
    res = trajectory(cfg, steps, 0.15, dev, held_out=64)
    curve = res["curve"]
    return {"steps": steps, "shape": "wikimel-shaped D=768 R=2048 N=101 B=64 T=8, learnable synthetic stream (gold planted at 0.15)",
            "max_abs_step_loss_diff": max(abs(a - b) for a, b in curve), "loss_first_step": curve[0], "loss_last_step": curve[-1],
            "held_out": {"mentions": res["held_out_mentions"], "hip": res["hip"], "oracle": res["oracle"],
                         "max_abs_score_diff": res["max_abs_held_out_score_diff"]},
            "against": "CPU oracle fp32 autograd + torch.optim.Adam (same seed-0 init, same batches)", "optimizer": res["optimizer"]}


# =====================================================================================================================
# stub worker: the launcher / timing / reporting plumbing on CPU + gloo (tests/test_bench_launcher.py)
# =====================================================================================================================
def stub_worker(ctx, args):
    """No GPU, no library: a "step" sleeps (rank r: (r + 1) ms) so that the barrier / max-over-ranks / per-rank
    reporting and - in train mode - the flat-bucket all-reduce can be exercised with gloo."""
    from drin_amd.train import GradBucket
    if os.environ.get("DRIN_BENCH_STUB_FAIL_RANK") == str(ctx.rank):    # tests: a rank that dies must fail the parent
        raise SystemExit(3)
    params = [torch.nn.Parameter(torch.zeros(257)), torch.nn.Parameter(torch.zeros(31, 3))]
    bucket = GradBucket(params, force=ctx.force)

    def step():
        time.sleep(1e-3 * (ctx.rank + 1))
        if args.mode == "train":
            for p in params:
                p.grad = torch.full_like(p, float(ctx.rank + 1))
            bucket.allreduce_mean()
        return None

    elapsed, per_rank, _ = ctx.timed(step, args.steps, args.warmup)
    if args.mode == "train":
        want = sum(range(1, ctx.world + 1)) / ctx.world
        assert all(torch.allclose(p.grad, torch.full_like(p, want)) for p in params), "stub all-reduce mean is wrong"
    if ctx.rank == 0:
        emit({"metric": "stub steps/sec (launcher plumbing only, no GPU work)", "stub": True,
                          "value": args.steps * ctx.world / elapsed, "unit": "steps/s", "n_gpus": ctx.world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "synthetic",
                          "config": {"workload": f"stub {args.mode}"}, "collectives_issued": bucket.collectives,
                          "rank_ms_per_step": [t / args.steps * 1e3 for t in per_rank]})
    ctx.finish()


# =====================================================================================================================
def parse_args(argv=None):
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
    ap.add_argument("--cache-format", default="f32", choices=["f32", "mixed_f16"],
                    help="--entity-cache: row format of the per-entity cache (drin_cache_format): every field fp32, or the fields that "
                         "reach the score through an per-pair scalars' operands stored as scaled fp16 (16.4 KB instead of 23.5 KB per pair)")
    ap.add_argument("--entity-cache", action="store_true",
                    help="table workload: score from the per-entity precompute cache (SURVEY.md 8f-2; built during warm-up)")
    ap.add_argument("--mentions", type=int, default=0,
                    help="table workload: stream this many mentions through the path in chunks of --chunk (SURVEY.md 8d config 5: "
                         "--mentions 1000000 --chunk 4096); a step is one chunk")
    ap.add_argument("--chunk", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=0, help="mentions per step per GPU (default 4096 wikimel / 16384 wikidiverse / 512 table)")
    ap.add_argument("--precision", default="bf16x3", choices=["f32", "bf16x3", "bf16x3_if16"],
                    help="contraction arithmetic: exact fp32 MFMA, split-bf16 (3 bf16 MFMAs, fp32 accumulate; max score error vs the fp32 "
                         "reference 1.4e-6, tests/test_gpu_parity.py), or split-bf16 with the entity-image contraction in one fp16 pass "
                         "(<= 2e-5 on trained weights; N >= 64 only)")
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
    ap.add_argument("--fused-adam", action="store_true", help="train mode: torch's fused single-kernel Adam (different rounding)")
    ap.add_argument("--torch-adam", action="store_true", help="train mode: torch.optim.Adam (multi-tensor, ~9 launches) instead of the library's one-launch Adam over the flat parameter bucket")
    ap.add_argument("--torch-loss", action="store_true", help="train mode: the torch TripletLoss instead of the library's loss/metric call")
    ap.add_argument("--force-collective", action="store_true",
                    help="N = 1: initialise a process group of ONE rank (RCCL unless DRIN_BENCH_BACKEND=gloo) and run the training step's "
                         "gradient all-reduce in it - the real collective code path on a one-GPU box; allreduce_ms is then non-zero")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--legs", default="auto",
                    help="secondary legs of the default run, comma separated: f32_exact,wikimel_mixed_f16,wikimel_bf16_features,wikidiverse_b4,train_step,train_b512,wikidiverse,table_cache | all | none "
                         "(auto: all for the default headline at N = 1, train_step at N > 1, none when a non-default workload / mode / batch is asked for)")
    ap.add_argument("--legs-file", default=LEGS_FILE,
                    help="where the FULL record goes (every leg, scaling model, step floors, provenance); stdout carries the <= 4 KB headline "
                         "line only; '' = do not write the file")
    ap.add_argument("--stub", action="store_true", help="CPU + gloo stand-in step (no GPU, no library): exercises the launcher and the timing plumbing only")
    args = ap.parse_args(argv)
    if args.warmup is None:
        args.warmup = 30 if args.mode == "train" else 3
    return args


def wanted_legs(args, world):
    names = ("f32_exact", "wikimel_mixed_f16", "wikimel_bf16_features", "wikidiverse_b4", "train_step", "train_b512", "wikidiverse", "table_cache")
    default_headline = (args.workload == "wikimel" and args.mode == "score" and not args.batch and not args.generic
                        and args.precision == "bf16x3" and args.features == "f32" and not args.graph)
    if args.legs == "auto":
        if not default_headline:
            return ()
        return names if world == 1 else ("train_step",)
    if args.legs == "none":
        return ()
    if args.legs == "all":
        return names
    legs = tuple(x for x in args.legs.split(",") if x)
    for x in legs:
        if x not in names:
            raise SystemExit(f"unknown leg {x!r}; choose from {names}")
    return legs


def leg_guard(name, fn):
    """A secondary leg must never take the headline line down with it."""
    try:
        t0 = time.perf_counter()
        out = fn()
        out["leg_wall_s"] = time.perf_counter() - t0
        return out
    except Exception as e:  # noqa: BLE001 - reported in the line, not swallowed silently
        import traceback
        traceback.print_exc(file=sys.stderr)
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        if torch.cuda.is_available():
            torch.cuda.empty_cache()


_REAL_STDOUT = None


def quiet_stdout():
    """The contract is ONE JSON line on stdout: whatever else writes to file descriptor 1 - RCCL prints a version banner
    there when its communicator comes up - is sent to stderr; `emit` writes the line to the saved descriptor."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


LEGS_FILE = os.path.join(REPO, "bench_legs.json")
LINE_LIMIT = 4096                 # bytes of the stdout line (the driver keeps an 8 KB tail; round 4's 32 KB line was not parsed)

_ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_replayed", "traffic_file", "launches", "avg_launch_ms",
              "algorithmic_bytes_per_launch", "executed_frac", "mfma_busy")
_HEAD_KEYS = ("metric", "stub", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "rank_ms_per_step", "roofline", "cpu_baseline", "parity", "hbm_fraction_whole_path",
              "mfma_fraction_whole_path", "hbm_traffic_whole_path_over_algorithmic", "path", "launch", "collectives_issued",
              "allreduce_ms", "allreduce_exposed_ms", "allreduce_bytes", "collective", "collective_error",
              "rank_roofline_avg_launch_ms", "scaling_model", "step_floor_ms", "final_loss", "kernel_ms_per_step")


def _strict(x, digits=9):
    """JSON-safe copy: non-finite floats -> None (strict JSON has no NaN / Infinity token), floats to `digits` significant."""
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{digits}g}") if digits else x
    if isinstance(x, dict):
        return {str(k): _strict(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_strict(v, digits) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if k in d} if isinstance(d, dict) else d


def _leg_summary(leg, out, name):
    """<= ~150 bytes per leg: value, ms_per_step, the dominant kernel's roofline fraction (nested legs flattened as a.b)."""
    if not isinstance(leg, dict):
        return
    if "error" in leg:
        out[name] = {"error": str(leg["error"])[:100]}
        return
    if "value" in leg:
        s = {"value": leg.get("value"), "ms_per_step": leg.get("ms_per_step", leg.get("ms_per_call"))}
        roof = leg.get("roofline")
        if isinstance(roof, dict):
            s["frac"], s["bound"] = roof.get("frac"), roof.get("bound")
        par = leg.get("parity") or (leg.get("train_parity") or {}).get("held_out")
        if isinstance(par, dict):
            err = par.get("max_abs_score_err", par.get("max_abs_score_diff"))
            if err is not None:
                s["err"] = err
        out[name] = s
    for k, v in leg.items():
        if isinstance(v, dict) and ("value" in v or "error" in v) and k not in ("roofline", "cpu_baseline", "parity"):
            _leg_summary(v, out, f"{name}.{k}")


def headline_only(line, legs_file=None):
    """The stdout line: the contract fields of the headline, its `roofline`, `cpu_baseline`, `parity`, a one-entry summary of
    every other leg, and where the full record went - nothing else (everything this drops is in `legs_file`)."""
    head = _pick(line, _HEAD_KEYS)
    if isinstance(head.get("roofline"), dict):
        head["roofline"] = _pick(head["roofline"], _ROOF_KEYS)
    if isinstance(head.get("cpu_baseline"), dict):
        head["cpu_baseline"] = _pick(head["cpu_baseline"], ("value", "b64_value", "unit", "cores", "kind", "sample"))
    if isinstance(head.get("parity"), dict):
        head["parity"] = _pick(head["parity"], ("max_abs_score_err", "top1_agreement", "mentions"))
    if isinstance(head.get("collective"), dict):
        head["collective"] = _pick(head["collective"], ("backend", "world", "forced_world_of_one", "overlap", "pieces", "collectives_issued",
                                                        "serial_ms_per_step", "no_collective_ms_per_step", "rank_gemm_avg_launch_ms"))
    sm = head.get("scaling_model")
    if isinstance(sm, dict):
        small = _pick(sm, ("collectives_in_the_scoring_path", "step_ms_without_collective", "allreduce_bytes", "hideable_ms"))
        if "by_world" in sm:
            small["expected_weak_scaling_efficiency"] = {n: v["expected_weak_scaling_efficiency"] for n, v in sm["by_world"].items()}
        head["scaling_model"] = small
    legs = {}
    for name, leg in (line.get("legs") or {}).items():
        _leg_summary(leg, legs, name)
    if legs:
        head["legs"] = legs
    if legs_file:
        head["legs_file"] = os.path.relpath(legs_file, REPO) if legs_file.startswith(REPO) else legs_file
    head = _strict(head)
    # never past the limit: drop the optional entries, least important first, until the line fits
    for k in ("kernel_ms_per_step", "scaling_model", "collective", "legs", "rank_roofline_avg_launch_ms", "path", "launch"):
        if len(json.dumps(head, allow_nan=False)) <= LINE_LIMIT:
            break
        head.pop(k, None)
    return head


def emit(line, legs_file=""):
    """ONE strict-JSON line <= LINE_LIMIT bytes on stdout (the headline only); the full record - every leg with its roofline,
    scaling model, step floor and provenance sentences - goes to `legs_file` (one short pointer line on stderr)."""
    legs_file = LEGS_FILE if legs_file == "" else legs_file
    full = json.dumps(_strict(line, digits=0), allow_nan=False)
    wrote = None
    if legs_file:
        try:
            with open(legs_file, "w") as f:
                f.write(full + "\n")
            wrote = legs_file
        except OSError as e:
            sys.stderr.write(f"bench.py: could not write {legs_file}: {e}\n")
    # (the full record is NOT copied to stderr: the driver keeps one 8 KB tail of stdout followed by stderr, and 30 KB of stderr
    #  would push the stdout line out of it)
    sys.stderr.write(f"[bench.py] full record ({len(full)} bytes): {wrote or 'not written'}\n")
    sys.stderr.flush()
    text = json.dumps(headline_only(line, wrote), allow_nan=False)
    if len(text) > LINE_LIMIT:
        # never no line at all: the contract fields alone, strings cut short (an `assert` here would end the run without a JSON
        # line - an INCOMPLETE record - and vanishes under `python -O`)
        short = {k: v for k, v in _strict(_pick(line, _HEAD_KEYS[:14])).items()}
        for k, v in list(short.items()):
            if isinstance(v, str):
                short[k] = v[:120]
            elif isinstance(v, dict):
                short[k] = {a: (b[:80] if isinstance(b, str) else b) for a, b in v.items() if not isinstance(b, (dict, list))}
        short["truncated"] = True
        if wrote:
            short["legs_file"] = os.path.relpath(wrote, REPO) if wrote.startswith(REPO) else wrote
        text = json.dumps(short, allow_nan=False)
        sys.stderr.write(f"[bench.py] headline line was {len(text)} bytes after truncation to the contract fields\n")
    out = _REAL_STDOUT or sys.stdout
    out.write(text + "\n")
    out.flush()


def main(argv=None):
    global LEGS_FILE
    args = parse_args(argv)
    LEGS_FILE = os.path.abspath(args.legs_file) if args.legs_file else None
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_children(args))                     # before anything touches the GPU in this process
    quiet_stdout()
    ctx = Ctx(args)
    if args.stub:
        return stub_worker(ctx, args)
    dev, world, rank = ctx.dev, ctx.world, ctx.rank

    if args.workload == "table":
        cfg = DrinConfig(num_candidates_data=1000)       # pooled-text entity rows: 19.7 KB per entity in fp32
    else:
        cfg = wikimel_config() if args.workload == "wikimel" else DrinConfig()
    sd = synth.make_state_dict(cfg, 7)

    if args.mode == "train":
        line = bench_train(ctx, cfg, sd, args.batch or 64, args.steps, args.warmup, args.precision, args.features, args.train_form,
                           args.train_entities, args.graph, args.fused_adam, args.torch_loss, not args.torch_adam,
                           force_collective=args.force_collective)
        if ctx.pg_error:
            line["collective_error"] = ctx.pg_error
        if rank == 0:
            emit(line)
        return ctx.finish()

    # WikiMEL: 4096 mentions = 413 696 pairs and 92 GB of resident inputs per step (of 288 GB): large steps
    # amortise the latency-bound mention-side kernels and the GEMM tile quantisation (22.6 vs 20.0 M pairs/s
    # at 1024 mentions)
    default_b = {"wikimel": 4096, "wikidiverse": 16384, "table": 512}[args.workload]
    B = args.batch or default_b
    fused = not args.generic and cfg.num_gcn_layers == 2
    model = make_model(cfg, sd, dev, args.precision, fused=not args.generic)
    legs = wanted_legs(args, world)
    extra = {}

    if args.workload == "table":
        table, g = build_table(cfg, args.entities, dev)
        if args.entity_cache:
            table.enable_cache(True, format=args.cache_format)
        if args.mentions:
            # config 5 streamed: a step is one chunk; K is fixed by --mentions / --chunk
            with torch.no_grad():
                model(make_table_chunk(cfg, table, min(args.chunk, 64), 99, dev, g))    # folds the weights / builds the cache
            el, n_chunks, out, last = stream_table(ctx, model, cfg, table, g, args.mentions, args.chunk)
            assert torch.isfinite(out).all()
            N = cfg.num_candidates_model
            line = {"metric": "mention x candidate pairs scored/sec", "value": args.mentions * N * world / el, "unit": "pairs/s",
                    "n_gpus": world, "steps": n_chunks, "warmup": 1, "ms_per_step": el / n_chunks * 1e3, "higher_is_better": True,
                    "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
                    "config": {"workload": f"BASELINE config 5: {args.mentions} mentions x {N - 1} candidates (+1 answer slot) gathered on the device from a "
                                           f"{args.entities}-entity table, streamed in chunks of {args.chunk} mentions (each drawn on the device right before it is scored)",
                               "mentions_per_step_per_gpu": args.chunk, "pairs_per_step": args.chunk * N * world,
                               "parallelism": f"dp{world} (mentions sharded, no collective)"},
                    "path": (f"per-entity cache ({args.cache_format} rows) + layer 2" if args.entity_cache else "fused two-layer") + ", " + args.precision}
            if rank == 0:
                line["parity"] = parity_of_timed_batch(cfg, sd, last, out, n_slices=2, width=1)
                emit(line)
            return ctx.finish()
        batch = make_table_chunk(cfg, table, B, 100 + rank, dev, g)
    else:
        batch = synth.make_device_batch(cfg, B, 100 + rank, dev, dtype=torch.bfloat16 if args.features == "bf16" else torch.float32)[:14]

    elapsed, per_rank, out, prof = run_score(ctx, model, batch, args.steps, args.warmup, args.graph)
    cached = args.workload == "table" and args.entity_cache
    line = score_line(ctx, cfg, args, B, batch, elapsed, per_rank, prof, args.steps, args.warmup, args.precision, args.features,
                      args.workload, cached, fused, args.graph)
    line["rank_roofline_avg_launch_ms"] = gather_ranks(ctx, line["roofline"]["avg_launch_ms"])
    line["scaling_model"] = {"collectives_in_the_scoring_path": 0,
                             "expected": "every rank scores its own mentions with the N = 1 schedule: ms_per_step and the dominant kernel's "
                                         "launch time of each rank equal the N = 1 run's (the stream kernel's +-4 % placement band per "
                                         "process apart), value = N x the N = 1 value; the timed bracket is the max over ranks"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["parity"] = parity_of_timed_batch(cfg, sd, batch, out, n_slices=(2 if args.workload == "table" else 8),
                                               width=(1 if args.workload == "table" else 8 if cfg.token_level_entities else 16))

    # ---- secondary legs: the other BASELINE configs under the same clock -------------------------------------------------
    if "f32_exact" in legs:
        def f32_leg():
            m = make_model(cfg, sd, dev, "f32")
            st = max(2, min(args.steps, 5))
            e, pr, o, pf = run_score(ctx, m, batch, st, 1)
            ln = score_line(ctx, cfg, args, B, batch, e, pr, pf, st, 1, "f32", args.features, args.workload, False, True)
            ln["parity"] = parity_of_timed_batch(cfg, sd, batch, o, n_slices=4, width=8)
            ln["max_abs_diff_vs_headline_scores"] = float((o - out).abs().max())
            return compact(ln, keep=("value", "unit", "ms_per_step", "steps", "dtype", "path", "roofline", "kernel_ms_per_step",
                                     "parity", "max_abs_diff_vs_headline_scores"))
        extra["f32_exact"] = leg_guard("f32_exact", f32_leg)
    if "wikimel_mixed_f16" in legs and world == 1:
        def f16_leg():
            # BASELINE's "bf16 inference" INSIDE the 1e-4 bar: precision by contraction (`bf16x3_if16`) - only the folded entity-image
            # contraction x_i (W_h1 W_ei)^T, 57 % of the path's FLOPs, runs ONE pass of the FP16 matrix instruction (11-bit operands) on
            # the fp16 plane the stream kernel writes under a power-of-two scale per row; its result reaches the score through a mean
            # over the 101 candidates alone (model.py:124-129,143-144).  Same batch as the headline; every score against the
            # headline's (split-bf16, itself within 1.4e-6 of the oracle), slices against the oracle
            m = make_model(cfg, sd, dev, "bf16x3_if16")
            st = min(args.steps, 10)
            e, pr, o, pf = run_score(ctx, m, batch, st, 2)
            ln = score_line(ctx, cfg, args, B, batch, e, pr, pf, st, 2, "bf16x3_if16", args.features, args.workload, False, fused)
            ln["parity"] = parity_of_timed_batch(cfg, sd, batch, o, n_slices=8, width=8)
            ln["parity"]["max_abs_diff_vs_headline_scores_all"] = float((o - out).abs().max())
            ln["parity"]["top1_agreement_vs_headline_all_mentions"] = float((o[:, :-1].argmax(1) == out[:, :-1].argmax(1)).float().mean())
            res = compact(ln)
            res["note"] = ("parity above is at the benchmark's random-init weights; with TRAINED weights 0.6-2.0e-5 against the exact-fp32 path "
                           "(split-bf16: 4-6e-6) - inside the 1e-4 bar (profiles/r4_precision_on_trained_weights.txt)")
            return res
        extra["wikimel_mixed_f16"] = leg_guard("wikimel_mixed_f16", f16_leg)
    if "wikimel_bf16_features" in legs and world == 1:
        def bf16_leg():
            # BASELINE configs 2-3 say "bf16": the headline batch with its six feature tensors stored as bf16 (read in place by
            # the fused path; arithmetic unchanged, so the scores equal the oracle's on the same stored values)
            b16 = [t.to(torch.bfloat16) if i in FEAT_SLOTS else t for i, t in enumerate(batch)]
            st = min(args.steps, 10)
            e, pr, o, pf = run_score(ctx, model, b16, st, 2)
            ln = score_line(ctx, cfg, args, B, b16, e, pr, pf, st, 2, args.precision, "bf16", args.workload, False, fused)
            ln["parity"] = parity_of_timed_batch(cfg, sd, b16, o, n_slices=4, width=8, fp32_batch=batch)
            res = compact(ln)
            return res
        extra["wikimel_bf16_features"] = leg_guard("wikimel_bf16_features", bf16_leg)
    del batch, out, model
    if torch.cuda.is_available():
        torch.cuda.empty_cache()

    if "wikidiverse_b4" in legs and world == 1:
        def b4_leg():
            # BASELINE configs[0]: the reference's own CPU-runnable case (WikiDiverse, batch 4) - one scoring call on the GPU
            # next to the CPU oracle on the same batch
            from oracle import drin_oracle as O
            wd = DrinConfig()
            wsd = synth.make_state_dict(wd, 7)
            m = make_model(wd, wsd, dev, "bf16x3")
            host = synth.make_batch(wd, 4, 1)
            b = [t.to(dev) for t in host[:14]]
            e, pr, o, pf = run_score(ctx, m, b, 200, 20)
            torch.set_num_threads(host_threads()["threads_used"])
            with torch.no_grad():
                it, el = _time_oracle(lambda: O.forward(wsd, host), 2.0, 2000)
                ref = O.forward(wsd, host)
            return {"workload": "BASELINE configs[0]: wikidiverse-shaped, batch 4, 11 candidates - one scoring call",
                    "ms_per_call": e / 200 * 1e3, "value": 44 * 200 / e, "unit": "pairs/s",
                    "cpu_oracle_ms_per_call": el / it * 1e3, "cpu_threads": torch.get_num_threads(),
                    "parity": {"max_abs_score_err": float((o.cpu() - ref).abs().max()), "mentions": 4}}
        extra["wikidiverse_b4"] = leg_guard("wikidiverse_b4", b4_leg)

    if "train_step" in legs:
        extra["train_step"] = leg_guard("train_step", lambda: bench_train(ctx, cfg, sd, 64, 20, 30))
        if world == 1 and rank == 0 and not args.no_cpu_baseline and "error" not in extra["train_step"]:
            extra["train_step"]["train_parity"] = leg_guard("train_parity", lambda: train_parity(dev))

        if world == 1:
            def rccl_leg():
                # BASELINE config 4's collective code path as far as ONE GPU can run it: an RCCL process group of one rank, the
                # flat gradient bucket all-reduced (ReduceOp.AVG, in place) in two pieces, the first started inside backward
                if not ctx.init_group():
                    return {"error": "process group of one rank: " + str(ctx.pg_error)}
                return bench_train(ctx, cfg, sd, 64, 20, 30, force_collective=True)
            extra["train_step_rccl_world1"] = leg_guard("train_step_rccl_world1", rccl_leg)

    if "train_b512" in legs and world == 1:
        # the same step at a rate-bound batch, and with the candidates indexed into a device-resident entity table
        extra["train_b512"] = leg_guard("train_b512", lambda: bench_train(ctx, cfg, sd, 512, 10, 20))
        extra["train_b512_table"] = leg_guard("train_b512_table", lambda: bench_train(ctx, cfg, sd, 512, 10, 20, train_form="table"))

    if "wikidiverse" in legs and world == 1:
        def wd_leg():
            wd = DrinConfig()
            wsd = synth.make_state_dict(wd, 7)
            m = make_model(wd, wsd, dev, "bf16x3")
            Bw, st = 16384, min(args.steps, 20)
            b32 = synth.make_device_batch(wd, Bw, 100, dev)[:14]
            e, pr, o, pf = run_score(ctx, m, b32, st, 3)
            l32 = score_line(ctx, wd, args, Bw, b32, e, pr, pf, st, 3, "bf16x3", "f32", "wikidiverse", False, True)
            l32["parity"] = parity_of_timed_batch(wd, wsd, b32, o, n_slices=8, width=16)
            b16 = [t.to(torch.bfloat16) if i in FEAT_SLOTS else t for i, t in enumerate(b32)]
            e, pr, o, pf = run_score(ctx, m, b16, st, 3)
            l16 = score_line(ctx, wd, args, Bw, b16, e, pr, pf, st, 3, "bf16x3", "bf16", "wikidiverse", False, True)
            l16["parity"] = parity_of_timed_batch(wd, wsd, b16, o, n_slices=8, width=16, fp32_batch=b32)
            # (the one-pass image contraction is gated on N >= 64: at N = 11 the fp16 pass measured 28.2-28.6 M pairs/s and 8e-6 at
            #  initialisation, but 1.2e-4 on trained weights - profiles/r4_precision_on_trained_weights.txt - so this shape runs split-bf16)
            return {"fp32_features": compact(l32), "bf16_features": compact(l16)}
        extra["wikidiverse"] = leg_guard("wikidiverse", wd_leg)

    if "table_cache" in legs and world == 1:
        def table_leg():
            tc = DrinConfig(num_candidates_data=1000)
            tsd = synth.make_state_dict(tc, 7)
            m = make_model(tc, tsd, dev, "bf16x3")
            table, g = build_table(tc, 1_000_000, dev)
            table.enable_cache()
            with torch.no_grad():
                m(make_table_chunk(tc, table, 64, 99, dev, g))              # folds the weights, builds the 23.5 GB cache
            chunk, mentions = 4096, 8 * 4096
            el, n_chunks, o, last = stream_table(ctx, m, tc, table, g, mentions, chunk)
            N = tc.num_candidates_model
            # one resident chunk, instrumented: the dominant kernel's roofline
            st = 5
            e, pr, o2, pf = run_score(ctx, m, last, st, 1)
            ln = score_line(ctx, tc, args, chunk, last, e, pr, pf, st, 1, "bf16x3", "f32", "table", True, True)
            res = compact(ln)
            res.update({"workload": f"BASELINE config 5 (SURVEY.md 8d): {N - 1} candidates (+1) per mention gathered on the device from a 1 000 000-entity table, "
                                    f"per-entity cache, {mentions} mentions streamed in chunks of {chunk} (python bench.py --workload table --entity-cache "
                                    f"--mentions 1000000 --chunk 4096 runs the full 1 M)",
                        "value": mentions * N / el, "ms_per_step": el / n_chunks * 1e3, "steps": n_chunks,
                        "resident_chunk_value": ln["value"],
                        "parity": parity_of_timed_batch(tc, tsd, last, o2, n_slices=8, width=1)})
            # the same chunks from DRIN_CACHE_MIXED_F16 rows (precision by storage: the operands of per-pair scalars as scaled
            # fp16, 16.4 KB per pair instead of 23.5): every score of the resident chunk against the fp32 rows'
            table.enable_cache(True, format="mixed_f16")
            with torch.no_grad():
                m(make_table_chunk(tc, table, 64, 99, dev, g))              # builds the 16.4 GB cache
            el_m, n_m, _, _ = stream_table(ctx, m, tc, table, g, mentions, chunk)
            e, pr, o3, pf = run_score(ctx, m, last, st, 1)
            lm = score_line(ctx, tc, args, chunk, last, e, pr, pf, st, 1, "bf16x3", "f32", "table", True, True, cache_format="mixed_f16")
            rm = compact(lm)
            rm.update({"value": mentions * N / el_m, "ms_per_step": el_m / n_m * 1e3, "steps": n_m, "resident_chunk_value": lm["value"],
                       "parity": parity_of_timed_batch(tc, tsd, last, o3, n_slices=8, width=1),
                       "vs_f32_rows": {"max_abs_score_diff": float((o3 - o2).abs().max()), "scores": int(o3.numel()),
                                       "top1_agreement": float((o3[:, :-1].argmax(1) == o2[:, :-1].argmax(1)).float().mean())}})
            res["mixed_f16_rows"] = rm
            return res
        extra["table_cache"] = leg_guard("table_cache", table_leg)

    if rank == 0:
        if extra:
            line["legs"] = extra
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(cfg, sd)
        emit(line)
    ctx.finish()


if __name__ == "__main__":
    main()
