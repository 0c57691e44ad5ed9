#!/bin/bash
# Batch-size sweeps of the scoring and training paths on one box (round 2): written to gpurun_out/sweeps.txt
O=gpurun_out/sweeps.txt
: > $O
echo "== WikiMEL-shaped scoring, split-bf16, batch sweep (bench.py --batch B --legs none)" >> $O
bash tools/bench_sweep.sh "1 4 16 64 256 512 1024 2048 4096" "bf16x3" >> $O 2>&1
echo "== exact fp32" >> $O
bash tools/bench_sweep.sh "64 1024 4096" "f32" >> $O 2>&1
echo "== small-batch latency, eager vs hipGraph replay" >> $O
bash tools/latency_sweep.sh >> $O 2>&1
echo "== BASELINE config 2: WikiDiverse-shaped, fp32- and bf16-stored features, with parity" >> $O
bash tools/config2_sweep.sh >> $O 2>&1
echo "== training step, batch sweep (bench.py --mode train --batch B)" >> $O
for b in 16 64 128 256 512 1024; do
  python bench.py --mode train --batch $b --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('train B=$b', round(d['ms_per_step'],3), 'ms', round(d['value']/1e6,3), 'M pairs/s', 'launches', d['library_launches_per_step'], {k:round(v,3) for k,v in d['kernel_ms_per_step'].items() if v})" >> $O
done
echo "== training step in table form (candidates indexed into a 50 000-entity device table)" >> $O
for b in 64 512; do
  python bench.py --mode train --batch $b --train-form table --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('train table B=$b', round(d['ms_per_step'],3), 'ms', round(d['value']/1e6,3), 'M pairs/s')" >> $O
done
cat $O
