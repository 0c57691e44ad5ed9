"""bench.py started as a plain process with --gpus N > 1 (no torch.distributed.run around it) launches its own ranks;
exercised here with the CPU + gloo stand-in step (`--stub`): the launcher, the barrier / max-over-ranks bracket, the
per-rank report and - in train mode - the flat-bucket all-reduce.  No GPU, no library call."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *argv], capture_output=True, text=True, env=e,
                          timeout=300)


@pytest.mark.parametrize("mode", ["score", "train"])
def test_plain_process_launches_its_own_ranks(mode):
    r = _run("--gpus", "2", "--stub", "--steps", "4", "--warmup", "1", "--mode", mode)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                                   # ONE JSON line, from rank 0
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["warmup"] == 1 and line["stub"] is True
    per_rank = line["rank_ms_per_step"]
    assert len(per_rank) == 2
    if mode == "score":
        assert per_rank[1] > per_rank[0] >= 1.0                        # rank r sleeps (r + 1) ms per step, no collective
    else:
        assert min(per_rank) >= 1.9                                    # the per-step all-reduce paces both ranks to the slower
    assert line["ms_per_step"] >= max(per_rank) * 0.999                # the bracket is the MAX over ranks
    assert line["value"] == pytest.approx(4 * 2 / (line["ms_per_step"] * 4e-3), rel=1e-6)


def test_single_rank_stub_line_unchanged_shape():
    r = _run("--stub", "--steps", "2", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and len(line["rank_ms_per_step"]) == 1


def test_world_size_mismatch_is_an_error():
    # started by a launcher with 3 ranks but asked for 1: refuse instead of silently reporting n_gpus = 3
    r = _run("--gpus", "1", "--stub", env={"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stderr + r.stdout)


def test_a_failing_rank_fails_the_parent():
    r = _run("--gpus", "2", "--stub", "--steps", "1", "--warmup", "0", env={"DRIN_BENCH_STUB_FAIL_RANK": "1"})
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_force_collective_runs_the_allreduce_in_a_world_of_one():
    """--force-collective at N = 1: a process group of one rank is initialised and the train step's flat-bucket all-reduce
    runs in it (gloo here; RCCL on the GPU box: tests/test_gpu_round3.py)."""
    r = _run("--stub", "--mode", "train", "--steps", "3", "--warmup", "1", "--force-collective")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["collectives_issued"] == 4
    r = _run("--stub", "--mode", "train", "--steps", "3", "--warmup", "1")
    assert json.loads(r.stdout.strip().splitlines()[-1])["collectives_issued"] == 0


def _reject_constant(tok):
    raise ValueError(f"non-strict JSON token {tok}")


@pytest.mark.parametrize("argv", [("--stub", "--steps", "2", "--warmup", "0"),
                                  ("--gpus", "2", "--stub", "--steps", "2", "--warmup", "0", "--mode", "train")])
def test_stdout_line_is_small_strict_json_and_the_full_record_goes_to_the_legs_file(argv, tmp_path):
    """The driver parses ONE stdout line and keeps an 8 KB tail: the line carries the headline only (<= 4 KB, no NaN /
    Infinity tokens); everything else is in --legs-file (round 4's 32 KB line left BENCH_r04.json with parsed = null)."""
    legs = tmp_path / "full.json"
    r = _run(*argv, "--legs-file", str(legs))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    assert len(lines[0].encode()) < 4096
    line = json.loads(lines[0], parse_constant=_reject_constant)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data", "config"):
        assert k in line, k
    assert "workload" in line["config"] and "model" not in line["config"]
    full = json.loads(legs.read_text(), parse_constant=_reject_constant)
    assert full["value"] == pytest.approx(line["value"], rel=1e-5) and line["legs_file"] == str(legs)


def test_headline_of_a_full_round4_record_fits_and_keeps_roofline_cpu_baseline_parity():
    """headline_only() on the largest record this repository has produced (round 4's 32 KB default line with 11 legs)."""
    sys.path.insert(0, REPO)
    import bench
    path = os.path.join(REPO, "profiles", "r4_wikimel_b4096_bench_all_legs.json")
    rec = json.loads(open(path).readline())
    rec["roofline"]["traffic"] = float("nan")                        # a non-finite value must come out as null, never NaN
    rec["legs"]["f32_exact"]["value"] = float("inf")
    text = json.dumps(bench.headline_only(rec, os.path.join(REPO, "bench_legs.json")), allow_nan=False)
    assert len(text) <= bench.LINE_LIMIT < 8192
    head = json.loads(text, parse_constant=_reject_constant)
    assert head["roofline"]["traffic"] is None and head["legs"]["f32_exact"]["value"] is None
    assert {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launches", "avg_launch_ms",
            "algorithmic_bytes_per_launch"} <= set(head["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(head["cpu_baseline"])
    assert {"max_abs_score_err", "top1_agreement", "mentions"} == set(head["parity"])
    assert head["legs_file"] == "bench_legs.json" and set(head["legs"]) >= {"train_step", "table_cache", "wikidiverse.fp32_features"}
    assert all(len(json.dumps(v)) <= 150 for v in head["legs"].values())
    # a record ten times fatter still fits: optional entries are dropped, the contract fields never
    rec["legs"] = {f"leg{i}": dict(rec["legs"]["train_step"]) for i in range(200)}
    text = json.dumps(bench.headline_only(rec, None), allow_nan=False)
    assert len(text) <= bench.LINE_LIMIT and "roofline" in json.loads(text) and "cpu_baseline" in json.loads(text)


def test_latest_counter_file_is_found_by_round_number_not_a_hard_coded_list(tmp_path, monkeypatch):
    sys.path.insert(0, REPO)
    import bench
    assert bench._latest("hbm_traffic.json").startswith("profiles/r")
    n = int(bench._latest("hbm_traffic.json").split("/r")[1].split("_")[0])
    existing = sorted(int(f.split("_")[0][1:]) for f in os.listdir(os.path.join(REPO, "profiles"))
                      if f.endswith("_hbm_traffic.json") and f[1:].split("_")[0].isdigit())
    assert n == existing[-1]


def test_eight_ranks_rendezvous_and_report_on_the_stub_path():
    """The first `bench.py --gpus 8` on an 8-GPU node must not also be the first 8-process rendezvous: eight stub ranks
    (CPU + gloo, no GPU, no library) launched by bench.py itself - one JSON line from rank 0, eight per-rank times, the bracket the
    MAX over ranks, whole-job value; train mode paces every rank to the slowest through the per-step all-reduce."""
    r = _run("--gpus", "8", "--stub", "--steps", "3", "--warmup", "1", "--mode", "train")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    assert len(lines[0].encode()) < 4096
    line = json.loads(lines[0], parse_constant=_reject_constant)
    assert line["n_gpus"] == 8 and len(line["rank_ms_per_step"]) == 8 and line["stub"] is True
    assert min(line["rank_ms_per_step"]) >= 7.9                        # rank 7 sleeps 8 ms per step: the all-reduce paces all eight
    assert line["ms_per_step"] >= max(line["rank_ms_per_step"]) * 0.999
    assert line["collectives_issued"] == 4
    r = _run("--gpus", "8", "--stub", "--steps", "2", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 8 and line["scaling"] == "weak"
    assert line["value"] == pytest.approx(2 * 8 / (line["ms_per_step"] * 2e-3), rel=1e-6)


def test_a_line_that_cannot_be_shrunk_still_comes_out_as_contract_fields(capsys, monkeypatch):
    """`emit()` never ends a run without a JSON line (the former `assert` did, and vanished under `python -O`): a record
    whose headline cannot be brought under the limit leaves the contract fields, strings cut short, marked `truncated`."""
    sys.path.insert(0, REPO)
    import bench
    monkeypatch.setattr(bench, "_REAL_STDOUT", None)
    rec = {"metric": "m" * 300, "value": 1.5, "unit": "pairs/s", "n_gpus": 1, "steps": 2, "warmup": 1, "ms_per_step": 3.0,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16x3", "data": "synthetic",
           "config": {"workload": "w" * 9000}, "roofline": {"bound": "hbm", "kernel": "k" * 9000}, "parity": {"mentions": "x" * 9000}}
    bench.emit(rec, legs_file=None)
    out = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")]
    assert len(out) == 1 and len(out[0]) <= bench.LINE_LIMIT
    line = json.loads(out[0], parse_constant=_reject_constant)
    assert line["truncated"] is True and line["value"] == 1.5 and line["unit"] == "pairs/s" and line["n_gpus"] == 1
    assert len(line["config"]["workload"]) == 80


def test_roofline_of_the_line_says_that_its_traffic_is_replayed():
    sys.path.insert(0, REPO)
    import bench
    assert {"traffic_replayed", "traffic_file"} <= set(bench._ROOF_KEYS)
    head = bench.headline_only({"metric": "m", "value": 1.0, "roofline": {"bound": "hbm", "traffic": 5e10, "traffic_replayed": True,
                                                                         "traffic_file": "profiles/r6_hbm_traffic.json", "traffic_source": "long sentence"},
                                "cpu_baseline": {"value": 4.5e4, "b64_value": 1.9e4, "unit": "pairs/s", "cores": 16, "kind": "port", "sample": "s", "host": {}}})
    assert head["roofline"]["traffic_replayed"] is True and head["roofline"]["traffic_file"].endswith("hbm_traffic.json")
    assert "traffic_source" not in head["roofline"] and head["cpu_baseline"]["b64_value"] == 1.9e4


def test_the_fp16_pass_gate_bench_reports_is_the_librarys():
    """`bench.if16_taken` asks `drin_image_contraction_passes` (it used to re-implement the gate)."""
    sys.path.insert(0, REPO)
    import bench
    from drin_amd.config import wikidiverse_config, wikimel_config
    wm, wd = wikimel_config(), wikidiverse_config()
    assert bench.if16_taken("bf16x3_if16", wm, "wikimel", 4096) is True
    assert bench.if16_taken("bf16x3_if16", wm, "wikimel", 64) is False          # fewer than 128 tiles of 256 x 256
    assert bench.if16_taken("bf16x3_if16", wm, "wikimel", 4096, "bf16") is False
    assert bench.if16_taken("bf16x3_if16", wm, "table", 4096) is False
    assert bench.if16_taken("bf16x3_if16", wd, "wikidiverse", 16384) is False   # N = 11 < 64
    assert bench.if16_taken("bf16x3", wm, "wikimel", 4096) is False
