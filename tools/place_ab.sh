#!/bin/bash
# Same-box A/B of where the four-phase GEMM kernels issue their DMA / loads (builds libdrin_hip_place1.so, _place2.so against the default)
O=gpurun_out/place_ab
rm -rf $O; mkdir -p $O
for v in 1 2; do
  DRIN_LIB_PATH=$PWD/drin_amd/libdrin_hip_place$v.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or full_size or reference_batch or determinism or bf16 or backward" > $O/tests_$v.log 2>&1
  echo "place $v tests: $(tail -1 $O/tests_$v.log)"
done
for i in 1 2; do
  python tools/gemm_bench.py 103424 2>&1 | grep "^planes" | sed "s/^/place0 /" >> $O/gemm.txt
  for v in 1 2; do DRIN_LIB_PATH=$PWD/drin_amd/libdrin_hip_place$v.so python tools/gemm_bench.py 103424 2>&1 | grep "^planes" | sed "s/^/place$v /" >> $O/gemm.txt; done
done
sort $O/gemm.txt
for i in 1 2 3; do
  python bench.py --legs none --no-cpu-baseline --steps 10 > $O/score_0_$i.json 2>> $O/err.log
  for v in 1 2; do DRIN_LIB_PATH=$PWD/drin_amd/libdrin_hip_place$v.so python bench.py --legs none --no-cpu-baseline --steps 10 > $O/score_${v}_$i.json 2>> $O/err.log; done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/place_ab/*.json")):
    l = json.load(open(f))
    print(f.split("/")[-1], round(l["ms_per_step"], 3), round(l["value"] / 1e6, 2), {k: round(v, 3) for k, v in l["kernel_ms_per_step"].items() if v})
PY
