#!/bin/bash
# Same-box A/B of two builds of the library (current vs drin_amd/libdrin_hip_prev.so through DRIN_LIB_PATH): parity tests on the current
# build, then the headline (4 alternating runs), WikiDiverse, training at 512.
O=gpurun_out/lib_ab_score
rm -rf $O; mkdir -p $O
PREV=$PWD/drin_amd/libdrin_hip_prev.so
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests (current build): $(tail -1 $O/tests.log)"
for i in 1 2 3 4; do
  python bench.py --legs none --no-cpu-baseline --steps 10 > $O/score_new_$i.json 2>> $O/err.log
  DRIN_LIB_PATH=$PREV python bench.py --legs none --no-cpu-baseline --steps 10 > $O/score_prev_$i.json 2>> $O/err.log
done
python bench.py --workload wikidiverse --legs none --no-cpu-baseline --steps 10 > $O/wd_new.json 2>> $O/err.log
DRIN_LIB_PATH=$PREV python bench.py --workload wikidiverse --legs none --no-cpu-baseline --steps 10 > $O/wd_prev.json 2>> $O/err.log
python bench.py --mode train --batch 512 > $O/t512_new.json 2>> $O/err.log
DRIN_LIB_PATH=$PREV python bench.py --mode train --batch 512 > $O/t512_prev.json 2>> $O/err.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/lib_ab_score/*.json")):
    l = json.load(open(f))
    print(f.split("/")[-1], round(l["ms_per_step"], 3), round(l["value"] / 1e6, 2), {k: round(v, 3) for k, v in l["kernel_ms_per_step"].items() if v})
PY
