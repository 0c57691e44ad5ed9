#!/bin/bash
# The full-row final kernel (gemm_rows.hip) against the two-kernel form (DRIN_ROWS_FINAL=0), same box: parity tests of the timed
# shapes, then the headline and the per-entity-cache chunk by kernel class, alternating; `rowsstub` = the K-loop without its epilogue.
O=gpurun_out/rows_final_ab
rm -rf $O; mkdir -p $O
python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py -m gpu -x -q -k "headline or config5 or bf16_stored or full_width or indexed_batch or trained" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
H="--steps 10 --warmup 3 --no-cpu-baseline --legs none"
TAB="--workload table --batch 4096 --entity-cache --steps 5 --warmup 2 --no-cpu-baseline --legs none"
for i in 1 2; do
  python bench.py $H --legs-file $O/head_rows_$i.json > /dev/null 2>> $O/err.log
  DRIN_ROWS_FINAL=0 python bench.py $H --legs-file $O/head_two_$i.json > /dev/null 2>> $O/err.log
  python bench.py $TAB --legs-file $O/tab_rows_$i.json > /dev/null 2>> $O/err.log
  DRIN_ROWS_FINAL=0 python bench.py $TAB --legs-file $O/tab_two_$i.json > /dev/null 2>> $O/err.log
done
DRIN_LIB_PATH=$PWD/drin_amd/libdrin_hip_rowsstub.so python bench.py $H --legs-file $O/head_stub_1.json > /dev/null 2>> $O/err.log
DRIN_LIB_PATH=$PWD/drin_amd/libdrin_hip_rowsstub.so python bench.py $TAB --legs-file $O/tab_stub_1.json > /dev/null 2>> $O/err.log
python - "$O" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        l = json.loads(open(f).readline())
        print(f.split("/")[-1], round(l["ms_per_step"], 3), {k: round(v, 3) for k, v in l["kernel_ms_per_step"].items() if v}, "err", (l.get("parity") or {}).get("max_abs_score_err"))
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -5 $O/err.log
