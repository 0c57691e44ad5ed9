#!/bin/bash
# BASELINE config 2: WikiDiverse-shaped inference swept over the batch size, fp32- and bf16-stored features,
# with the parity leg (max |score - oracle| and top-1 agreement on the same synthetic inputs).  Run on the GPU box.
for b in 4 64 1024 16384; do for f in f32 bf16; do
  timeout -k 10 300 python3 bench.py --workload wikidiverse --features $f --batch $b --steps 50 --warmup 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
p=d.get('parity') or {}
v=p.get('vs_fp32_features') or {}
print('B=%-6s features=%-4s %8.3f ms  %7.3f M pairs/s   max|err| %s  top-1 agreement %s   vs fp32 inputs: %s %s' % ('$b', '$f', d['ms_per_step'], d['value']/1e6, p.get('max_abs_score_err'), p.get('top1_agreement'), v.get('max_abs_score_err'), v.get('top1_agreement')))"
done; done
