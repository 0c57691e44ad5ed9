#!/bin/bash
# Same-box A/B of one bench leg: the shipped library against a variant build (python -m drin_amd.build --variant NAME with
# DRIN_EXTRA_FLAGS, built in the container), alternating.   tools/leg_ab.sh NAME LEG [runs]
V=${1:?variant}; LEG=${2:?leg}; N=${3:-3}
for i in $(seq $N); do
  for v in "" $V; do
    if [ -z "$v" ]; then python bench.py --legs $LEG --no-cpu-baseline --steps 10 --legs-file /tmp/x.json 2>/dev/null > /dev/null; else
      DRIN_LIB_PATH=$PWD/drin_amd/libdrin_hip_$v.so python bench.py --legs $LEG --no-cpu-baseline --steps 10 --legs-file /tmp/x.json 2>/dev/null > /dev/null; fi
    python - "${v:-shipped}" "$LEG" <<'PY'
import json, sys
l = json.load(open('/tmp/x.json'))
m = l['legs'][sys.argv[2]]
m = m if 'kernel_ms_per_step' in m else next(v for v in m.values() if isinstance(v, dict) and 'kernel_ms_per_step' in v)
print(sys.argv[1], 'headline', round(l['ms_per_step'], 3), {k: round(v, 3) for k, v in l['kernel_ms_per_step'].items() if v},
      '| leg', round(m['ms_per_step'], 3), {k: round(v, 3) for k, v in m['kernel_ms_per_step'].items() if v})
PY
  done
done
