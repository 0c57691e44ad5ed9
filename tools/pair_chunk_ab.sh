#!/bin/bash
# candidates per workgroup of the two pair kernels (DRIN_PAIR_CHUNK probe switch), one box: headline + WikiDiverse + small batches
O=gpurun_out/pair_chunk_ab.txt
: > $O
for rep in 1 2; do
  for pc in 0 32 48 101; do
    echo "== DRIN_PAIR_CHUNK=$pc (rep $rep)" >> $O
    DRIN_PAIR_CHUNK=$pc timeout -k 10 300 python tools/pipe_probe.py 4096 wikimel 0 >> $O 2>&1 || exit 1
  done
done
for pc in 0 32 48 101; do
  echo "== DRIN_PAIR_CHUNK=$pc B=64 / B=1 wikimel" >> $O
  DRIN_PAIR_CHUNK=$pc timeout -k 10 300 python tools/pipe_probe.py 64 wikimel 0 >> $O 2>&1 || exit 1
  DRIN_PAIR_CHUNK=$pc timeout -k 10 300 python tools/pipe_probe.py 1 wikimel 0 >> $O 2>&1 || exit 1
done
grep -v amdgpu $O
