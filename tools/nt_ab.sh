#!/bin/bash
# A/B on ONE box: non-temporal vs default-policy loads in k_entity_stream (HBM-bound kernel times move +-8 % between boxes)
set -e
for V in nt default nt default; do
  touch drin_amd/csrc/fused_kernels.hip
  if [ $V = default ]; then F="-DDRIN_NO_NT_LOADS"; else F=""; fi
  DRIN_EXTRA_FLAGS="$F" python -m drin_amd.build > gpurun_out/nt_build.log 2>&1
  timeout -k 10 300 python bench.py --no-cpu-baseline --legs none --steps 10 ${EXTRA} > gpurun_out/nt_$V.json 2> gpurun_out/nt_$V.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/nt_$V.json").read().strip().splitlines()[-1])
print("$V: %.2f M pairs/s  %.3f ms/step  stream %.3f ms" % (d["value"]/1e6, d["ms_per_step"], d["kernel_ms_per_step"]["stream"]))
PY
done
