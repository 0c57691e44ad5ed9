#!/bin/bash
# SQ counters of k_cached_pairs / k_pair_final on the config-5 workload (one pass per counter group; --pmc alone).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_cached
rm -rf $O; mkdir -p $O
CMD="python3 bench.py --workload table --batch 4096 --entity-cache --steps 2 --warmup 1 --no-cpu-baseline --legs none"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/a -- $CMD > $O/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/b -- $CMD > $O/b.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("a", "b"):
    for f in glob.glob(f"gpurun_out/pmc_cached/{d}/*/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_cached_pairs" in k or "k_pair_final" in k:
                name = "k_cached_pairs" if "cached" in k else "k_pair_final"
                acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
                n[(name, r["Counter_Name"])] += 1
        for name, cs in acc.items():
            print(name, {c: round(v / max(n[(name, c)], 1)) for c, v in cs.items()})
PY
