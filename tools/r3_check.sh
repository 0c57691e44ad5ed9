#!/bin/bash
# Round-3 check: the whole -m gpu suite, then the training legs (B = 64 three times, B = 512 both forms) and the forced RCCL
# world of one in each overlap mode (DRIN_OVERLAP).
O=gpurun_out/r3_check
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1
echo "pytest rc=$?" >> $O/gpu_tests.log
tail -3 $O/gpu_tests.log
bash tools/train_quick.sh > $O/train_quick.txt 2>&1 && cat $O/train_quick.txt
for m in forward none backward both; do
  DRIN_OVERLAP=$m python bench.py --mode train --batch 64 --force-collective > $O/b64_rccl_world1_$m.json 2>> $O/err.log
done
python - <<'PY'
import json
for m in ("forward", "none", "backward", "both"):
    try:
        l = json.load(open(f"gpurun_out/r3_check/b64_rccl_world1_{m}.json"))
        c = l["collective"]
        print(m, "ms/step", round(l["ms_per_step"], 4), "serial", round(c["serial_ms_per_step"], 4), "no collective", round(c["no_collective_ms_per_step"], 4),
              "allreduce_ms", round(l["allreduce_ms"], 4), "exposed", round(l["allreduce_exposed_ms"], 4), "floor", round(l["step_floor_ms"], 3))
    except Exception as e:
        print(m, "unreadable", e)
PY
