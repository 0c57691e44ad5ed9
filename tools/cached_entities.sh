#!/bin/bash
# k_cached_pairs gather rate against the table size (TLB reach / Infinity Cache)
set -e
for E in 10000 100000 1000000 4000000; do
  timeout -k 10 300 python bench.py --workload table --batch ${B:-256} --entities $E --entity-cache --no-cpu-baseline --legs none > gpurun_out/ce_$E.json 2> gpurun_out/ce_$E.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/ce_$E.json").read().strip().splitlines()[-1])
print("E=$E  %.2f M pairs/s  %.3f ms/step  stream %.3f ms  (%.0f GB/s)" % (d["value"]/1e6, d["ms_per_step"], d["kernel_ms_per_step"]["stream"], d["roofline"]["achieved"]))
PY
done
