#!/bin/bash
# candidates per workgroup in the per-entity-cache path (config 5, 4096 mentions x 1001 candidates; edit --batch for other call
# sizes), one box; the full-width cached-path tests run under every setting first
O=gpurun_out/cached_chunk_ab.txt
: > $O
run() { env "$@" timeout -k 10 300 python bench.py --workload table --batch 4096 --entity-cache --steps 5 --warmup 2 --no-cpu-baseline --legs none 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('$*', round(l['ms_per_step'],3), 'ms', round(l['value']/1e6,2), 'M pairs/s', {k:round(v,3) for k,v in l['kernel_ms_per_step'].items() if v}, (l.get('parity') or {}).get('max_abs_score_err'))" >> $O; }
for c in 16 64 128 256; do
  echo "tests DRIN_CACHED_CHUNK=$c: $(DRIN_CACHED_CHUNK=$c timeout -k 10 600 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -x -q -m gpu -k 'cache or config5 or table' 2>&1 | tail -1)" >> $O
done
for rep in 1 2; do
  for c in 16 32 64 128 256 512; do run DRIN_CACHED_CHUNK=$c; done
done
cat $O
