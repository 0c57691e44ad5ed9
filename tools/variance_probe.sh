#!/bin/bash
# Run-to-run spread of the headline on ONE box: fresh processes (16 by default; env assignments as arguments select variants,
# "-" = none); prints the step, the stream kernel's ms per launch and the other kernel classes.
set -e
O=gpurun_out/variance
rm -rf $O; mkdir -p $O
N=${RUNS:-16}
[ $# -eq 0 ] && set -- -
for i in $(seq 1 $N); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then e=""; else e="$v"; fi
    env $e python bench.py --legs none --no-cpu-baseline --steps 8 --warmup 2 > $O/run_${i}_$v.json 2>> $O/err.log
    python - "$O/run_${i}_$v.json" "$i $v" <<'PY'
import json, sys
l = json.load(open(sys.argv[1]))
k = l["kernel_ms_per_step"]
print(sys.argv[2], "step", round(l["ms_per_step"], 3), "M pairs/s", round(l["value"] / 1e6, 2), "stream", round(k["stream"], 3), "frac", round(l["roofline"]["frac"], 3),
      "gemm_x3", round(k["gemm_x3"], 3), "gemm_planes", round(k["gemm_planes"], 3), "gcn", round(k["gcn"], 3), flush=True)
PY
  done
done
