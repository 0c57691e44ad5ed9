#!/bin/bash
# Same-box A/B of tile shapes for the mid-sized split-bf16 NT products (DRIN_SK probe switch): parity tests, then the training step.
# usage: tools/sk_ab.sh mode ...        (off | 64 | 128 | 256: stream-K; t96 t128 t160 t192 t: data-parallel BM x 256)
O=gpurun_out/sk_ab
rm -rf $O; mkdir -p $O
for m in "$@"; do
  DRIN_SK=$m timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -x -q -m gpu -k "backward or reference_batch or reproducible or bit_identical or training_loop or indexed or table" > $O/tests_$m.log 2>&1
  echo "DRIN_SK=$m tests: $(tail -1 $O/tests_$m.log)"
  for i in 1 2; do DRIN_SK=$m python bench.py --mode train --batch 64 > $O/b64_${m}_$i.json 2>> $O/err.log; done
  DRIN_SK=$m python bench.py --mode train --batch 128 > $O/b128_$m.json 2>> $O/err.log
  DRIN_SK=$m python bench.py --mode train --batch 32 > $O/b32_$m.json 2>> $O/err.log
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/sk_ab/*.json")):
    try:
        l = json.load(open(f))
        print(f.split("/")[-1], round(l["ms_per_step"], 4), l["library_launches_per_step"], {k: round(v, 3) for k, v in l["kernel_ms_per_step"].items() if v})
    except Exception as e:
        print(f, "unreadable", e)
PY
