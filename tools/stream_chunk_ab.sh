#!/bin/bash
# candidates per workgroup of k_entity_stream and the pair kernels (DRIN_STREAM_CHUNK probe switch), one box, alternating
# usage: tools/stream_chunk_ab.sh [mentions ...]   (default 4096)
O=gpurun_out/stream_chunk_ab.txt
: > $O
[ $# -eq 0 ] && set -- 4096
for B in "$@"; do
  for rep in 1 2 3; do
    for sc in 16 32 48 101; do
      echo "== B=$B DRIN_STREAM_CHUNK=$sc (rep $rep)" >> $O
      DRIN_STREAM_CHUNK=$sc timeout -k 10 300 python tools/pipe_probe.py $B wikimel 0 >> $O 2>&1 || exit 1
    done
  done
done
grep -v amdgpu $O
