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
