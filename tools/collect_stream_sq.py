"""Reduce the SQ passes of tools/pmc_stream_r6.sh to one JSON: per row / stream kernel and section, where the waves' cycles went
(parked on s_waitcnt / barrier, issue-stalled, issuing) and how many instructions of each class a launch executes.

    python3 tools/collect_stream_sq.py gpurun_out/pmc_stream OUT.json

Units (MI355X_MICROARCH.md 'rocprofv3 PMC slots', 's_memtime tick vs SQ PMC units'): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_*
count quad-cycles summed over waves; WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES (disjoint); SQ_INSTS_* count
wave-instructions.  A kernel whose waves are parked (WAIT_ANY) most of the time while its instruction count per byte is small is
memory-bound; one with ACTIVE_INST_VALU near WAVE_CYCLES / waves-per-SIMD is issue-bound on the vector pipe.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

KERNELS = ("k_entity_stream", "k_pair_layer1", "k_pair_final", "k_cached_pairs", "k_axis_mean", "k_rows_", "k_gemm_rows")


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def reduce_dir(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(dict)
    for f in glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "drin" not in name or not any(k in name for k in KERNELS):
                continue
            k = short(name)
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    return acc, disp


def main():
    root, out_path = sys.argv[1], sys.argv[2]
    doc = {"note": "SQ counters summed over the launches of a pass, divided by the launch count; fractions are of SQ_WAVE_CYCLES"}
    for d in sorted(glob.glob(root + "/*_sq")):
        sec = os.path.basename(d)[:-3]
        acc, disp = reduce_dir(d)
        acc2, _ = reduce_dir(d + "2") if os.path.isdir(d + "2") else ({}, {})
        for k, c in acc.items():
            n = max(len(disp[k]), 1)
            wave = c.get("SQ_WAVE_CYCLES", 0.0)
            e = {"launches": n, "avg_launch_ms_profiled": sum(disp[k].values()) / n,
                 "per_launch": {x: c[x] / n for x in sorted(c)}}
            if wave > 0:
                e["frac_of_wave_cycles"] = {x[3:].lower(): c[x] / wave for x in
                                            ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS") if x in c}
            if k in acc2:
                e["per_launch"].update({x: acc2[k][x] / n for x in sorted(acc2[k])})
                w = acc2[k]
                waves = c.get("SQ_WAVES", 0.0)
                if waves > 0:
                    e["instructions_per_wave"] = {x[9:].lower(): w[x] / waves for x in w if x.startswith("SQ_INSTS_")}
            doc.setdefault(sec, {})[k] = e
            f = e.get("frac_of_wave_cycles", {})
            print(f"{sec:18s} {k[:70]:70s} {e['avg_launch_ms_profiled']:8.3f} ms  parked {f.get('wait_any', float('nan')):.3f}  "
                  f"issue-stalled {f.get('wait_inst_any', float('nan')):.3f}  issuing {f.get('active_inst_any', float('nan')):.3f}  "
                  f"valu {f.get('active_inst_valu', float('nan')):.3f}  " +
                  " ".join(f"{a}={b:.0f}" for a, b in sorted(e.get("instructions_per_wave", {}).items())))
    json.dump(doc, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
