"""Register / scratch / LDS usage of every kernel in the built gfx950 code objects, read from the AMDGPU metadata notes
(`python -m drin_amd.resources [--all]`).  No GPU needed.  `tests/test_host.py::test_no_kernel_uses_scratch` fails the build
check when any kernel of the library spills to scratch memory."""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile
from typing import Dict, List

from . import build as _build

LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = f"hipv4-amdgcn-amd-amdhsa--{_build.ARCH}"
_FIELDS = ("private_segment_fixed_size", "vgpr_count", "vgpr_spill_count", "sgpr_count", "sgpr_spill_count",
           "group_segment_fixed_size", "agpr_count")


def kernel_resources(obj_dir: str | None = None) -> List[Dict]:
    """One dict per kernel of every object under `drin_amd/csrc/build/`: demangled-ish name, source object, and the note's
    `.private_segment_fixed_size` (scratch bytes per lane), `.vgpr_count`, `.vgpr_spill_count`, `.sgpr_spill_count`, ..."""
    obj_dir = obj_dir or os.path.join(_build.CSRC, "build")
    out: List[Dict] = []
    with tempfile.TemporaryDirectory() as tmp:
        for name in sorted(os.listdir(obj_dir)):
            if not name.endswith(".o"):
                continue
            fat, co = os.path.join(tmp, name + ".fat"), os.path.join(tmp, name + ".co")
            r = subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", os.path.join(obj_dir, name)],
                               capture_output=True)
            if r.returncode != 0:                               # an object without device code (host-only translation unit)
                continue
            subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", f"--targets={TARGET}", f"--input={fat}", f"--output={co}",
                            "--unbundle"], check=True, capture_output=True)
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
            cur: Dict = {}
            for ln in notes.splitlines():
                m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", ln)
                if not m:
                    continue
                key, val = m.group(1), m.group(2).strip()
                if key == "name":
                    cur["name"] = val
                elif key in _FIELDS:
                    cur[key] = int(val)
                elif key == "wavefront_size" and "name" in cur:   # the record's keys are sorted: this is the last one
                    cur["object"] = name
                    out.append(cur)
                    cur = {}
    for k in out:
        try:
            k["demangled"] = subprocess.run([f"{LLVM}/llvm-cxxfilt", k["name"]], capture_output=True, text=True).stdout.strip()
        except OSError:
            k["demangled"] = k["name"]
    return out


def _code_object(obj: str, tmp: str) -> str:
    fat, co = os.path.join(tmp, "o.fat"), os.path.join(tmp, "o.co")
    subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj], check=True, capture_output=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", f"--targets={TARGET}", f"--input={fat}", f"--output={co}", "--unbundle"],
                   check=True, capture_output=True)
    return co


def kernel_isa(obj_name: str, kernel_substring: str, obj_dir: str | None = None) -> List[tuple]:
    """`(address, "mnemonic operands", branch target address or None)` for every instruction, in layout order, of the first
    kernel of `csrc/build/<obj_name>` whose mangled name contains `kernel_substring` (`llvm-objdump -d` of the gfx950 code object)."""
    obj_dir = obj_dir or os.path.join(_build.CSRC, "build")
    with tempfile.TemporaryDirectory() as tmp:
        co = _code_object(os.path.join(obj_dir, obj_name), tmp)
        text = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
    out: List[tuple] = []
    inside, base = False, 0
    for ln in text.splitlines():
        m = re.match(r"^([0-9a-f]+) <(\S+)>:", ln)
        if m:
            if inside:
                break
            inside, base = kernel_substring in m.group(2), int(m.group(1), 16)
            continue
        if inside and ln.startswith("\t"):
            body, _, comment = ln.partition("//")
            a = re.match(r"\s*([0-9A-Fa-f]+):", comment)
            t = re.search(r"<[^>]*\+0x([0-9a-f]+)>\s*$", comment)
            out.append((int(a.group(1), 16) if a else -1, body.strip(), base + int(t.group(1), 16) if t else None))
    return out


def _vgprs(operands: str) -> set:
    regs = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", operands):
        regs.update(range(int(a), int(b) + 1))
    regs.update(int(a) for a in re.findall(r"\bv(\d+)\b", operands))
    return regs


def untracked_load_hazards(isa: List[tuple]) -> List[str]:
    """Instructions that touch the destination registers of a vector-memory load before an `s_waitcnt vmcnt(n)` has retired
    it - ADVICE r3 (medium): the fp32-A four-phase GEMM issues its A loads by inline asm, so the compiler believes their
    registers defined at the asm statement while the data lands behind the hand-counted wait; a copy / re-materialisation the
    register allocator placed in between would read stale data and nothing else would notice.  The check walks the layout
    order once (forward branches fall through: both sides are seen) and the body of every loop (backward branch) a second
    time with the in-flight state of the first trip carried over; loads retire in order and `vmcnt(n)` leaves the n newest
    vector-memory operations (LDS-DMA and stores included) outstanding."""
    hazards: List[str] = []
    pending: list = []     # (text, destination VGPRs) of the vector-memory operations not yet retired, oldest first

    def step(text: str) -> None:
        mnem, _, ops = text.partition(" ")
        if mnem == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", ops)
            if m:
                del pending[:max(0, len(pending) - int(m.group(1)))]
            return
        in_flight = set().union(*[r for _t, r in pending]) if pending else set()
        vmem = mnem.startswith(("global_load", "global_store", "buffer_load", "buffer_store", "flat_load", "flat_store", "global_atomic",
                                "scratch_load", "scratch_store"))
        used = _vgprs(ops)
        if vmem and "load" in mnem and "_lds_" not in mnem and not re.search(r"\blds\b", ops):
            dst, _, rest = ops.partition(",")
            if _vgprs(rest) & in_flight:
                hazards.append(f"{text}   <- reads {sorted(_vgprs(rest) & in_flight)} still in flight")
            pending.append((text, _vgprs(dst)))
        else:
            if used & in_flight:
                hazards.append(f"{text}   <- touches {sorted(used & in_flight)} of {[t for t, r in pending if r & used]}")
            if vmem:
                pending.append((text, set()))

    index = {a: k for k, (a, _t, _b) in enumerate(isa)}
    # if / else as the compiler lays it out - `s_cbranch ELSE; <then>; s_branch END; ELSE: <else>; END:` - is two paths, not one
    # sequence: nothing falls through an unconditional branch, so the instruction behind it starts from the state of the
    # conditional branch that leads there (the first one recorded).  At a join the fall-through state goes on (the straight-line
    # K-loops this check exists for have no joins; a then-side load that is still in flight behind its join is not followed).
    entry: dict = {}
    dead = False
    for k, (addr, text, target) in enumerate(isa):
        if dead and addr in entry:
            pending[:] = entry[addr]
        entry.pop(addr, None)
        dead = False
        step(text)
        mnem = text.split(" ", 1)[0]
        if target is not None and target > addr and mnem.startswith("s_cbranch"):
            entry.setdefault(target, list(pending))
        dead = mnem == "s_branch" and target is not None and target > addr
        if target is not None and target <= addr and target in index:     # a loop closes here: its body once more
            for _a, t2, _b in isa[index[target]:k + 1]:
                step(t2)
    return hazards


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else argv
    rows = kernel_resources()
    # (SGPR spills go to VGPR lanes, not to memory: listed, not counted)
    bad = [k for k in rows if k.get("private_segment_fixed_size", 0) or k.get("vgpr_spill_count", 0)]
    show = rows if "--all" in argv else bad
    for k in sorted(show, key=lambda k: (-k.get("vgpr_count", 0), k["name"])):
        print(f'{k["object"]:24s} vgpr {k.get("vgpr_count", 0):3d} agpr {k.get("agpr_count", 0):3d} scratch {k.get("private_segment_fixed_size", 0):4d} B '
              f'vgpr spills {k.get("vgpr_spill_count", 0):3d} sgpr spills {k.get("sgpr_spill_count", 0):3d} lds {k.get("group_segment_fixed_size", 0):6d}  {k["demangled"][:150]}')
    print(f"{len(rows)} kernels, {len(bad)} with scratch / spills")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
