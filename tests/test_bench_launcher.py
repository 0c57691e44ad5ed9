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
