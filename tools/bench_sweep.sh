#!/bin/bash
# usage: tools/bench_sweep.sh "256 1024" "f32 bf16x3"   (GPU box)
for b in $1; do for p in $2; do
  timeout -k 10 300 python bench.py --precision $p --batch $b --steps 10 --no-cpu-baseline --legs none 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('B=$b $p', round(d['value']/1e6,2), 'Mpairs/s', round(d['ms_per_step'],3), 'ms', {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})"
done; done
