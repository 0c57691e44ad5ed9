#!/bin/bash
# Run-to-run spread of the headline on ONE box: fresh processes, alternating variants (env assignments given as arguments,
# "-" = none); prints the stream kernel's ms per launch and the step.
set -e
O=gpurun_out/variance
rm -rf $O; mkdir -p $O
for i in 1 2 3 4 5 6; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then e=""; else e="$v"; fi
    env $e python bench.py --legs none --no-cpu-baseline --steps 8 --warmup 2 > $O/run_${i}_$v.json 2>> $O/err.log
    python - "$O/run_${i}_$v.json" "$i $v" <<'PY'
import json, sys
l = json.load(open(sys.argv[1]))
print(sys.argv[2], round(l["ms_per_step"], 3), "stream", round(l["kernel_ms_per_step"]["stream"], 3), "parity", l.get("parity", {}).get("max_abs_err"), flush=True)
PY
  done
done
