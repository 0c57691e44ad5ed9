#!/bin/bash
# Same-box A/B of the scoring headline: the shipped library against a variant build (python -m drin_amd.build --variant NAME, or a copy of an older
# build as drin_amd/libdrin_hip_NAME.so), alternating fresh processes.   tools/headline_ab.sh NAME [runs] [extra bench args...]
V=${1:?variant}; N=${2:-3}; shift; shift
O=gpurun_out/headline_ab_$V.txt; : > $O
for i in $(seq $N); do for v in "" $V; do
  if [ -z "$v" ]; then timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --legs none --legs-file /tmp/x.json "$@" > /dev/null 2>>gpurun_out/headline_ab.err
  else DRIN_LIB_PATH=$PWD/drin_amd/libdrin_hip_$v.so timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --legs none --legs-file /tmp/x.json "$@" > /dev/null 2>>gpurun_out/headline_ab.err; fi
  python -c "
import json; l=json.load(open('/tmp/x.json')); print('${v:-shipped}', round(l['ms_per_step'],3), {k: round(v,3) for k,v in l['kernel_ms_per_step'].items() if v}, (l.get('parity') or {}).get('max_abs_score_err'))" >> $O
done; done; cat $O
