#!/bin/bash
O=gpurun_out/ablate_ab; rm -rf $O; mkdir -p $O
for i in 1 2; do
  python bench.py --legs none --no-cpu-baseline --steps 10 > $O/score_0_$i.json 2>> $O/err.log
  DRIN_LIB_PATH=$PWD/drin_amd/libdrin_hip_ablate.so python bench.py --legs none --no-cpu-baseline --steps 10 > $O/score_ablate_$i.json 2>> $O/err.log
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ablate_ab/*.json")):
    l = json.load(open(f))
    print(f.split("/")[-1], round(l["ms_per_step"], 3), {k: round(v, 3) for k, v in l["kernel_ms_per_step"].items() if v})
PY
