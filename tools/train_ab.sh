#!/bin/bash
# Same-box A/B of the training step (bench.py --mode train, WikiMEL-shaped, split-bf16): in-launch reductions vs separate
# reduce launches (DRIN_UNFOLDED=1), library Adam vs torch.optim.Adam; three runs each, alternating.
set -e
O=gpurun_out/train_ab
mkdir -p $O
for i in 1 2 3; do
  python bench.py --mode train --batch 64 > $O/folded_$i.json 2>> $O/err.log
  DRIN_UNFOLDED=1 python bench.py --mode train --batch 64 > $O/unfolded_$i.json 2>> $O/err.log
done
python bench.py --mode train --batch 64 --torch-adam > $O/folded_torchadam.json 2>> $O/err.log
python bench.py --mode train --batch 64 --graph --torch-adam > $O/folded_graph.json 2>> $O/err.log || true
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/train_ab/*.json")):
    try:
        l = json.load(open(f))
        print(f.split("/")[-1], round(l["ms_per_step"], 4), l["library_launches_per_step"], {k: round(v, 3) for k, v in l["kernel_ms_per_step"].items() if v})
    except Exception as e:
        print(f, "unreadable", e)
PY
