#!/bin/bash
# Same-box A/B of the headline arithmetic (bf16x3) against the one-pass fp16 image contraction on single planes (bf16x3_if16), alternating
# (with --features bf16 the mode is gated off since round 5: both arms then run the same kernels).   tools/if16_ab.sh [runs]   -> gpurun_out/if16_ab.txt
N=${1:-3}
O=gpurun_out/if16_ab.txt
: > $O
for i in $(seq $N); do
  for f in f32 bf16; do
    for p in bf16x3 bf16x3_if16; do
      timeout -k 10 300 python bench.py --precision $p --features $f --steps 10 --warmup 3 --no-cpu-baseline --legs none --legs-file /tmp/x.json > /dev/null 2>> gpurun_out/if16_ab.err
      python - "$p" "$f" <<'PY' >> $O
import json, sys
l = json.load(open('/tmp/x.json'))
print(sys.argv[1], sys.argv[2], 'ms/step', round(l['ms_per_step'], 3), 'M pairs/s', round(l['value'] / 1e6, 2),
      {k: round(v, 3) for k, v in l['kernel_ms_per_step'].items() if v}, 'err', (l.get('parity') or {}).get('max_abs_score_err'))
PY
    done
  done
done
cat $O
