#!/bin/bash
# Timing ablations (wrong results by construction) of the scoring headline in the given precision: the shipped library, a
# build without the GEMMs' global loads / weight DMA, one with a fraction of their MFMAs (variants built in the container).
P=${1:-bf16x3_i1}
for v in "" noloads nomfma; do
  for i in 1 2; do
    if [ -z "$v" ]; then python bench.py --legs none --no-cpu-baseline --steps 10 --precision $P 2>/dev/null > /tmp/x.json; else
      DRIN_LIB_PATH=$PWD/drin_amd/libdrin_hip_$v.so python bench.py --legs none --no-cpu-baseline --steps 10 --precision $P 2>/dev/null > /tmp/x.json; fi
    python -c "import json;l=json.load(open('/tmp/x.json'));print('${v:-shipped}', round(l['ms_per_step'],3), {k: round(v,3) for k,v in l['kernel_ms_per_step'].items() if v})"
  done
done
