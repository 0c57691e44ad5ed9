#!/bin/bash
# The N > 1 code path of bench.py on a ONE-GPU box: two ranks share device 0 and talk over gloo (RCCL refuses two ranks on
# one device).  Exercises the self-launcher, the barrier / max-over-ranks bracket on device tensors, the HIP Model's flat
# gradient bucket all-reduced in place and the one-launch Adam behind it.  Numbers are NOT scaling numbers.
set -e
export DRIN_BENCH_SHARE_GPU=1 DRIN_BENCH_BACKEND=gloo
O=gpurun_out/dp_rehearsal
mkdir -p $O
python bench.py --gpus 2 --batch 512 --steps 5 --warmup 2 --legs train_step --no-cpu-baseline --legs-file $O/score2_full.json > $O/score2.json 2> $O/score2.err
python bench.py --gpus 2 --mode train --batch 64 --steps 10 --warmup 10 > $O/train2.json 2> $O/train2.err
python bench.py --gpus 1 --mode train --batch 64 --steps 10 --warmup 10 > $O/train1.json 2> $O/train1.err
python - <<'PY'
import json
for f in ("score2", "train2", "train1"):
    l = json.load(open(f"gpurun_out/dp_rehearsal/{f}.json"))
    print(f, "n_gpus", l["n_gpus"], "ms/step", round(l["ms_per_step"], 3), "per rank", [round(x, 3) for x in l["rank_ms_per_step"]],
          "value", round(l["value"] / 1e6, 3), "allreduce_ms", l.get("allreduce_ms"), "loss", l.get("final_loss"),
          "train leg", (l.get("legs", {}).get("train_step", {}) or {}).get("ms_per_step"), "| stdout line bytes", len(json.dumps(l)),
          "collective", l.get("collective"), "scaling_model", l.get("scaling_model"))
PY
