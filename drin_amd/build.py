"""Builds libdrin_hip.so (gfx950) in-tree with hipcc.  No GPU is needed: hipcc cross-compiles.

    python -m drin_amd.build [--force] [--debug]
    python -m drin_amd.build --asan-host      # libdrin_hip_asan.so: the HOST side (argument validation, workspace layout, launch
                                              # sequencing) under AddressSanitizer + UBSan, device code uninstrumented
                                              # (-fno-gpu-sanitize; GPU sanitizer runs are not available on this pool).
                                              # Load it with DRIN_LIB_PATH=... and LD_PRELOAD=<asan runtime> (tests/test_cabi.py).

The shared library is written next to this file so that it travels with the source tree
(it is git-ignored, not gpurun-ignored).  Objects are rebuilt only when a source or header
is newer than them.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libdrin_hip.so")
ARCH = "gfx950"
SOURCES = ["api.hip", "stream_kernels.hip", "gemm_f32.hip", "gemm_bf16x3.hip", "gemm_x3_planes.hip", "gemm_tn_bf16x3.hip", "gcn_kernels.hip", "backward_kernels.hip", "fused_kernels.hip", "fused_forward.hip", "vector_kernels.hip", "loss_kernels.hip", "entity_cache.hip", "optim_kernels.hip"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newest_header() -> float:
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return max(os.path.getmtime(h) for h in hs)


ASAN_LIB = os.path.join(HERE, "libdrin_hip_asan.so")
SAN_FLAGS = ["-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined", "-g"]


def asan_runtime() -> str:
    """Path of clang's shared ASan runtime (what LD_PRELOAD needs when python loads the instrumented library)."""
    out = subprocess.run([_hipcc(), "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(out) or not os.path.exists(out):
        import glob
        hits = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
        if not hits:
            raise RuntimeError("clang's libclang_rt.asan-x86_64.so not found")
        out = hits[-1]
    return out


def build(force: bool = False, debug: bool = False, verbose: bool = True, asan_host: bool = False, variant: str = "") -> str:
    """`variant`: a second build of the library next to the shipped one - objects under csrc/build_<variant>, the library as
    libdrin_hip_<variant>.so - for same-box A/Bs of a compile-time switch (DRIN_EXTRA_FLAGS="-D..." python -m drin_amd.build
    --variant NAME; select it at run time with DRIN_LIB_PATH)."""
    obj_dir = os.path.join(CSRC, "build_asan" if asan_host else ("build_" + variant if variant else "build"))
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    flags = [f"--offload-arch={ARCH}", "-O1" if asan_host else "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
             "-fvisibility=hidden", f"-I{INCLUDE}", f"-I{CSRC}"]
    flags += os.environ.get("DRIN_EXTRA_FLAGS", "").split()
    if asan_host:
        flags += SAN_FLAGS
    if debug:
        flags += ["-g", "-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
    hdr_t = _newest_header()
    # objects compiled with other flags (DRIN_EXTRA_FLAGS probes, --debug) are stale whatever their timestamps say
    stamp = os.path.join(obj_dir, "flags.txt")
    want = " ".join([hipcc] + flags)
    try:
        same_flags = open(stamp).read() == want
    except OSError:
        same_flags = False
    if not same_flags:
        force = True
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(obj_dir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_t):
            jobs.append([hipcc, "-c", s, "-o", o] + flags)

    def run(cmd):
        if verbose:
            print("[drin_amd.build]", " ".join(cmd[:5]), "...", flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip() and verbose:
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    with open(stamp, "w") as f:
        f.write(want)
    lib = ASAN_LIB if asan_host else (os.path.join(HERE, f"libdrin_hip_{variant}.so") if variant else LIB)
    if jobs or not os.path.exists(lib):
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib] + (SAN_FLAGS[:2] + ["-shared-libsan"] if asan_host else []) + objs)
    return lib


if __name__ == "__main__":
    variant = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else ""
    print(build(force="--force" in sys.argv, debug="--debug" in sys.argv, asan_host="--asan-host" in sys.argv, variant=variant))
