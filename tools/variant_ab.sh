#!/bin/bash
# Same-box A/B of the shipped library against a variant build (python -m drin_amd.build --variant NAME with DRIN_EXTRA_FLAGS,
# built in the container: both .so files travel): alternating runs of the training step at 64 / 512 mentions and of the
# scoring headline in the given precision.   tools/variant_ab.sh NAME [precision]
V=${1:?variant name}
P=${2:-bf16x3}
O=gpurun_out/variant_ab_$V
mkdir -p $O
for i in 1 2 3; do
  for b in 64 512; do
    python bench.py --mode train --batch $b > $O/new_train${b}_$i.json 2>> $O/err.log
    DRIN_LIB_PATH=$PWD/drin_amd/libdrin_hip_$V.so python bench.py --mode train --batch $b > $O/${V}_train${b}_$i.json 2>> $O/err.log
  done
  python bench.py --legs none --no-cpu-baseline --steps 10 --precision $P > $O/new_score_$i.json 2>> $O/err.log
  DRIN_LIB_PATH=$PWD/drin_amd/libdrin_hip_$V.so python bench.py --legs none --no-cpu-baseline --steps 10 --precision $P > $O/${V}_score_$i.json 2>> $O/err.log
done
python - "$O" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        l = json.load(open(f))
        print(f.split("/")[-1], round(l["ms_per_step"], 4), {k: round(v, 3) for k, v in l["kernel_ms_per_step"].items() if v})
    except Exception as e:
        print(f, "unreadable", e)
PY
