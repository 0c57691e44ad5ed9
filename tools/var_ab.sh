#!/bin/bash
# Same-box A/B of a VARIANT build (drin_amd/libdrin_hip_var.so, DRIN_LIB_PATH) against the current library on the headline.
set -e
O=gpurun_out/var_ab
rm -rf $O; mkdir -p $O
for i in 1 2 3 4; do
  for v in cur var; do
    if [ $v = var ]; then export DRIN_LIB_PATH=$PWD/drin_amd/libdrin_hip_var.so; else unset DRIN_LIB_PATH; fi
    python bench.py --legs none --no-cpu-baseline --steps 10 --warmup 3 $EXTRA > $O/${v}_$i.json 2>> $O/err.log
    python - "$O/${v}_$i.json" "$v $i" <<'PY'
import json, sys
l = json.load(open(sys.argv[1]))
print(sys.argv[2], round(l["ms_per_step"], 3), {k: round(v, 3) for k, v in l["kernel_ms_per_step"].items() if v}, flush=True)
PY
  done
done
