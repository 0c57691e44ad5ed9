#!/bin/bash
# Same-box A/B of the packed-fp32 row helpers (DRIN_PK_ROWS, csrc/row_ops.h) against the scalar build
# (DRIN_EXTRA_FLAGS="-DDRIN_PK_ROWS=0" python -m drin_amd.build --variant scalar): the headline in process on one resident batch,
# the per-entity-cache chunk in both row formats alternating between fresh processes.
O=gpurun_out/pk_rows_ab
rm -rf $O; mkdir -p $O
python tools/probes/inprocess_lib_ab.py drin_amd/libdrin_hip_scalar.so 3 > $O/headline_inprocess.txt 2>&1
TAB="--workload table --batch 4096 --entity-cache --steps 5 --warmup 2 --no-cpu-baseline --legs none"
for i in 1 2; do
  for fmt in f32 mixed_f16; do
    python bench.py $TAB --cache-format $fmt --legs-file $O/packed_${fmt}_$i.json > /dev/null 2>> $O/err.log
    DRIN_LIB_PATH=$PWD/drin_amd/libdrin_hip_scalar.so python bench.py $TAB --cache-format $fmt --legs-file $O/scalar_${fmt}_$i.json > /dev/null 2>> $O/err.log
  done
done
python - "$O" <<'PY'
import json, glob, sys
print(open(sys.argv[1] + "/headline_inprocess.txt").read())
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        l = json.loads(open(f).readline())
        print(f.split("/")[-1], round(l["ms_per_step"], 3), {k: round(v, 3) for k, v in l["kernel_ms_per_step"].items() if v}, "err", (l.get("parity") or {}).get("max_abs_score_err"))
    except Exception as e:
        print(f, "unreadable", e)
PY
