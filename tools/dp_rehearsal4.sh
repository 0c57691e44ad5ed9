#!/bin/bash
# Rehearsal of the N = 4 code path of bench.py on a ONE-GPU box: four ranks launched by bench.py itself, all on device 0, gloo instead of
# RCCL (which refuses several ranks on one device).  Scoring line with the train_step leg (overlap mode "forward" = the N > 1 default), then
# train mode in every overlap mode.  Not scaling numbers: the collective goes through the host.
O=gpurun_out/dp_rehearsal4
rm -rf $O; mkdir -p $O
export DRIN_BENCH_SHARE_GPU=1 DRIN_BENCH_BACKEND=gloo
python bench.py --gpus 4 --batch 256 --steps 5 --warmup 2 --legs train_step --no-cpu-baseline --legs-file $O/score_n4_full.json > $O/score_n4.json 2> $O/score_n4.err
echo "score rc=$?"
for m in forward none backward both; do
  DRIN_OVERLAP=$m python bench.py --gpus 4 --mode train --batch 16 --steps 5 --warmup 3 --legs-file $O/train_n4_$m.json > $O/train_n4_${m}_stdout.json 2> $O/train_n4_$m.err
  echo "train $m rc=$?"
done
python - <<'PY'
import json, glob
l = json.load(open("gpurun_out/dp_rehearsal4/score_n4_full.json"))
t = l["legs"]["train_step"]
print("score n_gpus", l["n_gpus"], "value", round(l["value"] / 1e6, 2), "rank ms", [round(x, 2) for x in l["rank_ms_per_step"]], "| train_step leg:", round(t["ms_per_step"], 3), "ms",
      "allreduce_ms", round(t["allreduce_ms"], 3), "exposed", round(t["allreduce_exposed_ms"], 3), t["collective"])
for f in sorted((f for f in glob.glob("gpurun_out/dp_rehearsal4/train_n4_*.json") if not f.endswith("_stdout.json"))):
    l = json.load(open(f))
    c = l["collective"]
    print(f.split("/")[-1], "n", l["n_gpus"], round(l["ms_per_step"], 3), "ms | serial", round(c["serial_ms_per_step"], 3), "none", round(c["no_collective_ms_per_step"], 3),
          "allreduce", round(l["allreduce_ms"], 3), "exposed", round(l["allreduce_exposed_ms"], 3), c["overlap"], c["pieces"], c["in_place"], "loss", round(l["final_loss"], 5))
PY
