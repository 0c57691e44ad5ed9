#!/bin/bash
# candidates per workgroup of k_entity_stream (DRIN_STREAM_CHUNK probe switch), one box, alternating, headline batch
O=gpurun_out/stream_chunk_ab.txt
: > $O
for rep in 1 2 3; do
  for sc in 0 32 48 101; do
    echo "== DRIN_STREAM_CHUNK=$sc (rep $rep)" >> $O
    DRIN_STREAM_CHUNK=$sc timeout -k 10 300 python tools/pipe_probe.py 4096 wikimel 0 >> $O 2>&1 || exit 1
  done
done
grep -v amdgpu $O
