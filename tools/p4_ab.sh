#!/bin/bash
# Same-box A/B of the four-phase GEMM pipeline (default) against the previous kernels (DRIN_P4=0): the whole -m gpu suite on
# the default, then the headline, WikiDiverse, a 1024-mention call, training at 64 / 512.
O=gpurun_out/p4_ab
rm -rf $O; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1
echo "tests: $(tail -1 $O/tests.log)"
for i in 1 2 3; do for m in 0 1; do
  DRIN_P4=$m python bench.py --steps 10 --warmup 3 --no-cpu-baseline --legs none > $O/head_${m}_$i.json 2>> $O/err.log
done; done
for m in 0 1; do
  DRIN_P4=$m python bench.py --workload wikidiverse --steps 10 --warmup 3 --no-cpu-baseline --legs none > $O/wd_$m.json 2>> $O/err.log
  DRIN_P4=$m python bench.py --mode train --batch 512 > $O/train512_$m.json 2>> $O/err.log
  DRIN_P4=$m python bench.py --mode train --batch 64 > $O/train64_$m.json 2>> $O/err.log
  DRIN_P4=$m python bench.py --batch 256 --steps 20 --warmup 5 --no-cpu-baseline --legs none > $O/b256_$m.json 2>> $O/err.log
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/p4_ab/*.json")):
    l = json.load(open(f))
    print(f.split("/")[-1], round(l["ms_per_step"], 3), round(l["value"] / 1e6, 2), "M pairs/s", {k: round(v, 3) for k, v in l["kernel_ms_per_step"].items() if v})
PY
