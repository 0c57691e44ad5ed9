#!/bin/bash
# the stream kernel's two placement modes (fresh processes run it in ~8.2 or ~8.8 ms on the same batch seed).
# N fresh processes, each under its own rocprofv3 --pmc pass (TCC memory-side counters; program right after `--`), then one table:
# per process the kernel's duration and the counters per launch - do the slow processes show more DRAM credit stalls / a deeper read queue?
#   bash tools/placement_pmc.sh [processes per counter set]     -> gpurun_out/placement_pmc/{table.txt, runs.json}
N=${1:-8}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=${OUT:-gpurun_out/placement_pmc}
rm -rf $O; mkdir -p $O
export PYTHONUNBUFFERED=1
HEAD="--steps 4 --warmup 1 --no-cpu-baseline --legs none"
SETA=${SETA:-"TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_sum GRBM_GUI_ACTIVE"}
SETB=${SETB:-"TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE"}
# (second call of round 6: the address-translation side -
#  SETA="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
#  SETB="TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum GRBM_GUI_ACTIVE")
for i in $(seq 1 $N); do
  rocprofv3 --pmc $SETA --output-format csv -d $O/a_$i -- python3 bench.py $HEAD > $O/a_$i.log 2>&1 || echo "[placement_pmc] set A run $i FAILED"
  echo "[placement_pmc] A $i done"
done
for i in $(seq 1 $N); do
  rocprofv3 --pmc $SETB --output-format csv -d $O/b_$i -- python3 bench.py $HEAD > $O/b_$i.log 2>&1 || echo "[placement_pmc] set B run $i FAILED (counter names?)"
  echo "[placement_pmc] B $i done"
done
python3 - "$O" <<'PY'
import collections, csv, glob, json, os, sys
root = sys.argv[1]
runs = []
for d in sorted(glob.glob(root + "/[ab]_*")):
    if not os.path.isdir(d):
        continue
    files = glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv")
    if not files:
        continue
    acc, disp = collections.defaultdict(float), {}
    for r in csv.DictReader(open(files[0])):
        if "k_entity_stream" not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
        disp[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    n = max(len(disp), 1)
    runs.append({"run": os.path.basename(d), "launches": n, "stream_ms": sum(disp.values()) / n, "per_launch": {k: v / n for k, v in acc.items()}})
json.dump(runs, open(root + "/runs.json", "w"), indent=1)
with open(root + "/table.txt", "w") as out:
    for which in "ab":
        rs = sorted([r for r in runs if r["run"].startswith(which)], key=lambda r: r["stream_ms"])
        if not rs:
            continue
        keys = sorted(rs[0]["per_launch"])
        print(f"counter set {which.upper()}: k_entity_stream, per launch, fresh processes sorted by the kernel's duration", file=out)
        print("run      ms      " + "  ".join(f"{k:>36s}" for k in keys), file=out)
        for r in rs:
            print(f"{r['run']:6s} {r['stream_ms']:7.3f}  " + "  ".join(f"{r['per_launch'].get(k, float('nan')):36.4g}" for k in keys), file=out)
        if "TCC_EA0_RDREQ_LEVEL_sum" in keys:
            print("  mean read-queue residency (LEVEL / RDREQ, TCC cycles): " + "  ".join(f"{r['run']} {r['per_launch']['TCC_EA0_RDREQ_LEVEL_sum'] / max(r['per_launch']['TCC_EA0_RDREQ_sum'], 1):.0f}" for r in rs), file=out)
print(open(root + "/table.txt").read())
PY
