#!/bin/bash
# k_cached_pairs occupancy sweep on the GPU box: rebuilds entity_cache.hip with 2 / 3 / 4 workgroups per CU
set -e
for W in 2 3 4; do
  touch drin_amd/csrc/entity_cache.hip
  DRIN_EXTRA_FLAGS="-DDRIN_CACHED_PAIRS_WG_PER_CU=$W" python -m drin_amd.build > gpurun_out/cs_build_$W.log 2>&1
  timeout -k 10 300 python bench.py --workload table --batch ${B:-256} --entity-cache --no-cpu-baseline --legs none > gpurun_out/cs_$W.json 2> gpurun_out/cs_$W.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/cs_$W.json").read().strip().splitlines()[-1])
print("WG/CU=$W  %.2f M pairs/s  %.3f ms/step  stream %.3f ms  (%.0f GB/s)" % (d["value"]/1e6, d["ms_per_step"], d["kernel_ms_per_step"]["stream"], d["roofline"]["achieved"]))
PY
done
