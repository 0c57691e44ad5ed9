#!/bin/bash
# rocprofv3 kernel trace of the training step (B = 64 by default): per-kernel averages of the profiled command -> gpurun_out/train_trace/
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=${1:-64}
O=gpurun_out/train_trace
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 bench.py --mode train --batch $B --steps 20 --warmup 30 > $O/bench_under_rocprof.json 2> $O/err.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/train_trace/t/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per step (70 steps incl. warm-up / instrumented):", round(tot / 70 / 1e3, 1), "us")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    print(f'{float(r["TotalDurationNs"]) / 70 / 1e3:8.1f} us/step  x{int(r["Calls"]) / 70:5.1f}  avg {float(r["AverageNs"]) / 1e3:7.1f} us  {r["Name"][:110]}')
PY
cp $(ls $O/t/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
