#!/bin/bash
# headline with the four-phase planes GEMM, image rows through the fp32-staged GEMM (XI=0) or written as planes by the stream kernel (XI=1)
O=gpurun_out/p4_ab2
rm -rf $O; mkdir -p $O
DRIN_P4=1 DRIN_XI_PLANES=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or full_size or reference_batch or determinism or bf16" > $O/tests.log 2>&1
echo "tests (P4=1 XI=1): $(tail -1 $O/tests.log)"
for i in 1 2 3; do for x in 0 1; do
  DRIN_P4=1 DRIN_XI_PLANES=$x python bench.py --steps 10 --warmup 3 --no-cpu-baseline --legs none > $O/head_xi${x}_$i.json 2>> $O/err.log
done; done
DRIN_P4=1 DRIN_XI_PLANES=1 python bench.py --workload wikidiverse --steps 10 --warmup 3 --no-cpu-baseline --legs none > $O/wd_xi1.json 2>> $O/err.log
DRIN_P4=1 DRIN_XI_PLANES=0 python bench.py --workload wikidiverse --steps 10 --warmup 3 --no-cpu-baseline --legs none > $O/wd_xi0.json 2>> $O/err.log
DRIN_P4=0 DRIN_XI_PLANES=0 python bench.py --workload wikidiverse --steps 10 --warmup 3 --no-cpu-baseline --legs none > $O/wd_base.json 2>> $O/err.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/p4_ab2/*.json")):
    l = json.load(open(f))
    print(f.split("/")[-1], round(l["ms_per_step"], 3), round(l["value"] / 1e6, 2), "M pairs/s", {k: round(v, 3) for k, v in l["kernel_ms_per_step"].items() if v})
PY
