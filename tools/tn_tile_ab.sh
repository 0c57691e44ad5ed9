#!/bin/bash
# Same-box A/B of the weight-gradient (TN) tile: 128 x 128 two-per-CU (default at M ~ 10^4) against 256 x 256 (DRIN_TN_TILE).
set -e
O=gpurun_out/tn_ab
mkdir -p $O
for i in 1 2 3; do
  for t in 128 256; do
    DRIN_TN_TILE=$t python bench.py --mode train --batch 64 > $O/b64_t${t}_$i.json 2>> $O/err.log
  done
done
for t in 128 256; do
  DRIN_TN_TILE=$t python bench.py --mode train --batch 512 > $O/b512_t${t}.json 2>> $O/err.log
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/tn_ab/*.json")):
    try:
        l = json.load(open(f))
        print(f.split("/")[-1], round(l["ms_per_step"], 4), l["library_launches_per_step"], {k: round(v, 3) for k, v in l["kernel_ms_per_step"].items() if v})
    except Exception as e:
        print(f, "unreadable", e)
PY
