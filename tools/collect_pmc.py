"""Turn the rocprofv3 --pmc passes of bench.py into profiles/<tag>_hbm_traffic.json.

Run on the GPU box, FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots, MI355X_MICROARCH.md
'rocprofv3 PMC slots'):
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    python3 tools/collect_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r2_hbm_traffic.json [section] [command]
Corrections (MI355X_MICROARCH.md, HBM): the counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of
the bytes of a wide (16 B/lane) coalesced streaming read, so it is doubled; WRITE_SIZE is exact for 16-byte
streaming stores.
"""
import collections, csv, glob, json, sys

COUNTS = {}


def per_kernel(d):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "drin" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        COUNTS[k] = max(COUNTS.get(k, 0), len(v))
    return {k: sum(v) / len(v) for k, v in acc.items()}

section = sys.argv[4] if len(sys.argv) > 4 else "kernels"          # e.g. kernels_table_cache: merged into an existing file
command = sys.argv[5] if len(sys.argv) > 5 else "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --legs none (defaults: wikimel B=4096 bf16x3 fused)"
fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
out = {}
for k in sorted(set(fetch) | set(write)):
    fb = 2.0 * fetch.get(k, 0.0) * 1024.0
    wb = write.get(k, 0.0) * 1024.0
    out[k] = {"fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
# the whole step: every kernel that runs once or more per scoring call (the anchor - the one pass over the entity bytes - runs
# exactly once per call; weight folds / cache builds run once per process and are left out), launches per call x bytes per launch
anchor = next((k for k in out if "k_entity_stream" in k or "k_cached_pairs" in k), None)
whole = None
if anchor:
    calls = COUNTS[anchor]
    # (a kernel of the step runs a whole number of times per call; weight folds and the per-entity cache build run once per
    #  process - their launch counts are no multiple of the number of calls)
    per_call = {k: COUNTS[k] // calls for k in out if COUNTS.get(k, 0) >= calls and COUNTS[k] % calls == 0}
    whole = {"calls_in_the_pass": calls, "launches_per_call": per_call,
             "hbm_bytes_per_call": sum(out[k]["hbm_bytes_per_launch"] * n for k, n in per_call.items()),
             "fetch_bytes_per_call": sum(out[k]["fetch_bytes_per_launch"] * n for k, n in per_call.items()),
             "write_bytes_per_call": sum(out[k]["write_bytes_per_launch"] * n for k, n in per_call.items())}
import os
doc = json.load(open(sys.argv[3])) if os.path.exists(sys.argv[3]) else {}
doc.setdefault("whole_path", {})[section] = whole
doc["note"] = "FETCH_SIZE x2 (gfx950 wide-read correction) + WRITE_SIZE, KiB -> bytes, mean per launch"
doc.setdefault("commands", {})[section] = command
doc[section] = out
json.dump(doc, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if "k_entity_stream" in k or "k_cached_pairs" in k}, indent=1))
