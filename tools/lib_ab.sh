#!/bin/bash
# Same-box A/B of two builds of the library on the scoring headline and the training step: DRIN_LIB_PATH selects drin_amd/libdrin_hip_prev.so
# (built from the previous commit) against the current drin_amd/libdrin_hip.so; three alternating runs per batch size.
O=gpurun_out/lib_ab
mkdir -p $O
for i in 1 2; do
  python bench.py --legs none --no-cpu-baseline --steps 10 > $O/score_new_$i.json 2>> $O/err.log
  DRIN_LIB_PATH=$PWD/drin_amd/libdrin_hip_prev.so python bench.py --legs none --no-cpu-baseline --steps 10 > $O/score_prev_$i.json 2>> $O/err.log
done
for i in 1 2 3; do
  for b in 64 512; do
    python bench.py --mode train --batch $b > $O/new_${b}_$i.json 2>> $O/err.log
    DRIN_LIB_PATH=$PWD/drin_amd/libdrin_hip_prev.so python bench.py --mode train --batch $b > $O/prev_${b}_$i.json 2>> $O/err.log
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/lib_ab/*.json")):
    try:
        l = json.load(open(f))
        print(f.split("/")[-1], round(l["ms_per_step"], 4), l.get("library_launches_per_step"), {k: round(v, 3) for k, v in l["kernel_ms_per_step"].items() if v}, "loss", l.get("final_loss"))
    except Exception as e:
        print(f, "unreadable", e)
PY
