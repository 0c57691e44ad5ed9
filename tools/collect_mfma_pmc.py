"""Reduce rocprofv3 --pmc SQ passes of bench.py to profiles/<tag>_mfma_pmc.json: per GEMM kernel, how busy the matrix pipes
were and where the waves' cycles went.

Run on the GPU box (program directly after `--`, counters alone in the pass - tools/pmc_mfma.sh):
    rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
              SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d DIR -- python3 bench.py ...
    python3 tools/collect_mfma_pmc.py DIR OUT.json SECTION "COMMAND"

Units (MI355X_MICROARCH.md, 'Per-instruction cycle constants' / 's_memtime tick vs SQ PMC units'):
  SQ_VALU_MFMA_BUSY_CYCLES  cycles, summed over the chip's 1024 SIMDs (= 16 x the v_mfma_f32_16x16x32_bf16 wave-instructions issued,
                            64 x the v_mfma_f32_32x32x2_f32 ones)
  SQ_BUSY_CYCLES            cycles, summed over the 32 shader engines (8 XCDs x 4) that rocprofv3 adds up
  SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_*   quad-cycles, summed over waves
so   mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x SQ_BUSY_CYCLES / 32) = MFMA_BUSY / (32 x SQ_BUSY_CYCLES)
is the fraction of the kernel's SIMD-cycles in which a matrix instruction was executing, and WAIT_ANY / WAVE_CYCLES the fraction of
wave time parked on s_waitcnt / barriers (WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES, disjoint).
"""
import collections
import csv
import glob
import json
import os
import re
import sys

FAMILIES = ("k_gemm_bf16x3", "k_gemm_x3_planes", "k_gemm_tn_bf16x3", "k_gemm_f32", "k_gemm_f32_group")


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", name)
    if not m:
        m = re.search(r"(k_[a-z0-9_]+)", name)          # mangled: _ZN4drin3x3p16k_gemm_x3_planesILb1ELb1EE...
    if m and m.group(2):
        return m.group(1) + m.group(2)
    if m:
        tail = re.search(re.escape(m.group(1)) + r"I([A-Za-z0-9_]*?)E+v", name)
        return m.group(1) + ("<" + tail.group(1) + ">" if tail else "")
    return name[:60]


def reduce_dir(d):
    files = glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv")
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(dict)
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "drin" not in name:
                continue
            k = short(name)
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    out = {}
    for k, c in acc.items():
        if not any(fam in k for fam in FAMILIES):
            continue
        n = len(disp[k])
        busy, sqb, wave = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("SQ_BUSY_CYCLES", 0.0), c.get("SQ_WAVE_CYCLES", 0.0)
        e = {"launches": n, "avg_launch_ms_profiled": sum(disp[k].values()) / max(n, 1),
             "sum": {x: c[x] for x in sorted(c)},
             "mfma_busy": busy / (32.0 * sqb) if sqb else None,
             "wave_time": {"parked_on_waitcnt_or_barrier": c.get("SQ_WAIT_ANY", 0.0) / wave if wave else None,
                           "issue_stalled": c.get("SQ_WAIT_INST_ANY", 0.0) / wave if wave else None,
                           "issuing": c.get("SQ_ACTIVE_INST_ANY", 0.0) / wave if wave else None,
                           "issuing_lds": c.get("SQ_ACTIVE_INST_LDS", 0.0) / wave if wave else None}}
        if c.get("GRBM_GUI_ACTIVE") and n:
            # clock the chip held under this kernel: GUI-active cycles (summed over the 8 XCDs) / 8 / kernel time
            e["effective_clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / (sum(disp[k].values()) * 1e-3) / 1e9
        out[k] = e
    fam = {}
    for name in ("k_gemm_bf16x3", "k_gemm_x3_planes", "k_gemm_tn_bf16x3"):
        ks = [k for k in out if k.startswith(name)]   # every form of the family: <tile shapes>, _p4 (four-phase pipeline), _sk
        b = sum(out[k]["sum"].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for k in ks)
        s = sum(out[k]["sum"].get("SQ_BUSY_CYCLES", 0.0) for k in ks)
        if s:
            fam[name] = {"mfma_busy": b / (32.0 * s), "kernels": sorted(ks)}
    ks = [k for k in out if any(k.startswith(n) for n in ("k_gemm_bf16x3", "k_gemm_tn_bf16x3", "k_gemm_x3_planes"))]
    b = sum(out[k]["sum"].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for k in ks)
    s = sum(out[k]["sum"].get("SQ_BUSY_CYCLES", 0.0) for k in ks)
    if s:
        fam["split_bf16_family"] = {"mfma_busy": b / (32.0 * s), "kernels": sorted(ks)}
    return out, fam


def main():
    d, dst, section, command = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
    kernels, fam = reduce_dir(d)
    doc = json.load(open(dst)) if os.path.exists(dst) else {}
    doc["note"] = ("mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES): fraction of the kernel's SIMD-cycles with a matrix instruction "
                   "executing (MFMA_BUSY: cycles over 1024 SIMDs; SQ_BUSY_CYCLES: cycles over 32 shader engines); wave_time fractions are "
                   "quad-cycle counters over SQ_WAVE_CYCLES; sums are over every launch of the profiled command (tools/collect_mfma_pmc.py)")
    doc.setdefault("sections", {})[section] = {"command": command, "families": fam, "kernels": kernels}
    json.dump(doc, open(dst, "w"), indent=1)
    for k, v in sorted(kernels.items()):
        w = v["wave_time"]
        print(f'{section:22s} {k[:58]:58s} x{v["launches"]:<4d} {v["avg_launch_ms_profiled"]:.4f} ms  mfma_busy {v["mfma_busy"] if v["mfma_busy"] is None else round(v["mfma_busy"], 3)}'
              f'  parked {w["parked_on_waitcnt_or_barrier"] and round(w["parked_on_waitcnt_or_barrier"], 3)} stalled {w["issue_stalled"] and round(w["issue_stalled"], 3)}'
              f' issuing {w["issuing"] and round(w["issuing"], 3)} clock {round(v.get("effective_clock_ghz", 0), 2)}')
    print(section, "families:", {k: round(v["mfma_busy"], 3) for k, v in fam.items()})


if __name__ == "__main__":
    main()
