#!/bin/bash
# Timing ablations of the mid-sized split-bf16 NT kernel inside the B = 64 / 128 training step (wrong results by construction):
# the K-loop without its global loads, and without its MFMAs, against the product build.  Builds: DRIN_EXTRA_FLAGS=-DDRIN_ABLATE_*.
O=gpurun_out/ablate_mid
rm -rf $O; mkdir -p $O
for v in product NO_LOADS NO_MFMA product2; do
  lib=$PWD/drin_amd/libdrin_hip_ablate_$v.so
  case $v in product*) lib=$PWD/drin_amd/libdrin_hip.so;; esac
  for b in 64 128; do
    DRIN_LIB_PATH=$lib timeout -k 10 300 python bench.py --mode train --batch $b > $O/b${b}_$v.json 2>> $O/err.log || echo "b$b $v failed" 
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ablate_mid/*.json")):
    try:
        l = json.load(open(f))
        print(f.split("/")[-1], round(l["ms_per_step"], 4), {k: round(v, 3) for k, v in l["kernel_ms_per_step"].items() if v})
    except Exception as e:
        print(f, "unreadable", e)
PY
