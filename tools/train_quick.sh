#!/bin/bash
# The training legs three times (B = 64) and once (B = 512, per-pair and table form): ms per step + kernel classes.
set -e
O=gpurun_out/train_quick
mkdir -p $O
for i in 1 2 3; do python bench.py --mode train --batch 64 > $O/b64_$i.json 2>> $O/err.log; done
python bench.py --mode train --batch 512 > $O/b512.json 2>> $O/err.log
python bench.py --mode train --batch 512 --train-form table > $O/b512_table.json 2>> $O/err.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/train_quick/*.json")):
    try:
        l = json.load(open(f))
        print(f.split("/")[-1], round(l["ms_per_step"], 4), l["library_launches_per_step"], {k: round(v, 3) for k, v in l["kernel_ms_per_step"].items() if v})
    except Exception as e:
        print(f, "unreadable", e)
PY
