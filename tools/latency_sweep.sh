#!/bin/bash
# Small-batch scoring latency, eager launches against hipGraph replay (run on the GPU box).
for wl in wikimel wikidiverse; do
  for b in 1 4 64 256; do
    for g in "" "--graph"; do
      python3 bench.py --workload $wl --batch $b --steps 200 --warmup 20 --no-cpu-baseline --legs none $g 2>/dev/null | \
        python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl B=$b', d['launch'], '%.3f ms' % d['ms_per_step'], '%.2f M pairs/s' % (d['value']/1e6))"
    done
  done
done
