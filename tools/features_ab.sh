#!/bin/bash
# Same-box A/B of the scoring headline with bf16-stored features (--features bf16): shipped library against a variant, alternating.
V=${1:?variant}; N=${2:-4}
for i in $(seq $N); do
  for v in "" $V; do
    if [ -z "$v" ]; then python bench.py --legs none --no-cpu-baseline --steps 10 --features bf16 2>/dev/null > /tmp/x.json; else
      DRIN_LIB_PATH=$PWD/drin_amd/libdrin_hip_$v.so python bench.py --legs none --no-cpu-baseline --steps 10 --features bf16 2>/dev/null > /tmp/x.json; fi
    python -c "import json;l=json.load(open('/tmp/x.json'));print('${v:-shipped}', round(l['ms_per_step'],3), {k: round(v,3) for k,v in l['kernel_ms_per_step'].items() if v})"
  done
done
