"""CPU-side checks of the C-ABI library: it loads without a GPU, exports exactly what
include/drin_hip.h declares, and its host-only entry points behave."""
import ctypes as C
import os
import re
import subprocess

import pytest

from drin_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "drin_hip.h")


def _declared():
    src = open(HEADER).read()
    return sorted(set(re.findall(r"DRIN_API\s+[\w\s\*]+?\b(drin_\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared()
    assert len(declared) >= 10
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/drin_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == declared, "python binding and header disagree"
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l and "__device_stub__" not in l)
    assert exported == declared, "library exports symbols the header does not declare"


def test_version_and_defaults():
    lib = _lib.load()
    assert lib.drin_version() == _lib.ABI_VERSION
    assert b"gfx950" in lib.drin_build_info()
    c = _lib.DrinConfigC()
    assert lib.drin_default_config(C.byref(c)) == _lib.OK
    assert (c.num_candidates, c.embed_dim, c.image_dim, c.num_layers, c.dynamic_edges) == (11, 768, 2048, 2, 1)
    assert list(c.edge_enabled) == [1.0] * 4
    assert abs(c.layer_norm_eps - 1e-5) < 1e-12 and abs(c.cosine_eps - 1e-8) < 1e-15 and c.clip_scale == 100.0
    assert lib.drin_default_config(None) == _lib.E_NULL


def test_workspace_query_and_validation():
    lib = _lib.load()
    c = _lib.DrinConfigC()
    lib.drin_default_config(C.byref(c))
    c.batch = 64
    inf, trn = lib.drin_workspace_bytes(C.byref(c), 0), lib.drin_workspace_bytes(C.byref(c), 1)
    assert 0 < inf < trn
    c.batch = 128
    assert lib.drin_workspace_bytes(C.byref(c), 0) > inf
    c.embed_dim = 770
    assert lib.drin_workspace_bytes(C.byref(c), 0) == 0
    assert b"multiples of 4" in lib.drin_last_error()
    c.embed_dim = 768
    for removed_or_unknown in (2, 4, 7):                                # 2, 4: the out-of-tolerance one-pass modes removed with ABI 6
        c.precision = removed_or_unknown
        assert lib.drin_workspace_bytes(C.byref(c), 0) == 0
    c.precision = 0
    c.num_layers = 9
    assert lib.drin_workspace_bytes(C.byref(c), 0) == 0


def test_fused_workspace_size_is_a_pure_function_of_the_configuration():
    """ABI 6 removed the process-wide setters (`drin_set_pipeline`, `drin_set_weight_gradient_passes`): a size query depends on its
    `drin_config` alone - the same answer before and after other calls, from any thread."""
    lib = _lib.load()
    for name in ("drin_set_pipeline", "drin_set_weight_gradient_passes"):
        assert not hasattr(lib, name), name
    c = _lib.DrinConfigC()
    _lib.check(lib.drin_default_config(C.byref(c)))
    sizes = []
    for _ in range(2):
        row = []
        for batch, prec in ((1200, _lib.PREC_BF16X3), (1200, _lib.PREC_F32), (64, _lib.PREC_BF16X3), (1200, _lib.PREC_BF16X3_IF16)):
            c.batch, c.precision = batch, prec
            row.append(lib.drin_fused_workspace_bytes(C.byref(c)))
        sizes.append(row)
    assert sizes[0] == sizes[1] and all(v > 0 for v in sizes[0])
    assert sizes[0][3] > sizes[0][0]                                    # the fp16 mode's row scales


def test_null_arguments_are_reported_not_crashed():
    lib = _lib.load()
    c = _lib.DrinConfigC()
    lib.drin_default_config(C.byref(c))
    c.batch = 1
    b = _lib.DrinBatchC()
    assert lib.drin_edges_fwd(C.byref(c), C.byref(b), None, None, None) == _lib.E_NULL
    assert b"NULL" in lib.drin_last_error()
    assert lib.drin_forward(C.byref(c), C.byref(b), None, None, 0, None, 0, None, None) == _lib.E_NULL
    assert lib.drin_profile_end(None, None) == _lib.E_SHAPE


def test_loss_entry_point_validates_on_host():
    lib = _lib.load()
    assert lib.drin_loss_workspace_bytes(0) == 0 and lib.drin_loss_workspace_bytes(64) == 64 * 40
    ks = (C.c_int32 * 2)(1, 5)
    one = C.c_void_p(16)   # never dereferenced: every check below fails before a launch
    assert lib.drin_triplet_topk(one, one, 0, 11, 0.25, ks, 2, one, None, one, one, 4096, None) == _lib.E_SHAPE
    assert lib.drin_triplet_topk(one, one, 4, 11, 0.25, ks, 9, one, None, one, one, 4096, None) == _lib.E_SHAPE
    ks[1] = 11                                                             # torch.topk(k > row length) raises too
    assert lib.drin_triplet_topk(one, one, 4, 11, 0.25, ks, 2, one, None, one, one, 4096, None) == _lib.E_SHAPE
    assert b"top-k 11" in lib.drin_last_error()
    ks[1] = 5
    assert lib.drin_triplet_topk(None, one, 4, 11, 0.25, ks, 2, one, None, one, one, 4096, None) == _lib.E_NULL
    assert lib.drin_triplet_topk(one, one, 4, 11, 0.25, ks, 2, one, None, one, one, 8, None) == _lib.E_WORKSPACE


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No CPU / eager fallback: without the built .so the binding raises and names the build command."""
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libdrin_hip.so"))
    with pytest.raises(ImportError, match="no fallback"):
        _lib.load()


def test_plain_c_abi_example_builds(tmp_path):
    """examples/score_c_abi.cpp (no torch, no Python) compiles and links against the header and the library as shipped."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "score_c_abi")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", f"-I{REPO}/include", f"{REPO}/examples/score_c_abi.cpp",
                        f"-L{REPO}/drin_amd", "-ldrin_hip", f"-Wl,-rpath,{REPO}/drin_amd", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.exists(exe)


def test_ctypes_mirrors_match_the_header_layout(tmp_path):
    """Every struct of include/drin_hip.h as a plain C compiler lays it out == its ctypes mirror in drin_amd/_lib.py
    (size and the offset of every field): the binding a maintainer writes from the header alone stays in step."""
    pairs = [("drin_config", _lib.DrinConfigC), ("drin_batch", _lib.DrinBatchC), ("drin_layer_params", _lib.DrinLayerParamsC),
             ("drin_params", _lib.DrinParamsC), ("drin_param_grads", _lib.DrinParamGradsC), ("drin_trace", _lib.DrinTraceC)]
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', 'int main(void) {']
    for cname, cls in pairs:
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for field, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{field} %zu\\n", offsetof({cname}, {field}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-o", str(exe), str(src)], check=True)   # the header is plain C
    got = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for cname, cls in pairs:
        assert int(got[cname]) == C.sizeof(cls), cname
        for field, _ in cls._fields_:
            assert int(got[f"{cname}.{field}"]) == getattr(cls, field).offset, f"{cname}.{field}"
    # and the header declares no field the mirror lacks (sizes equal + every mirrored field at its offset + no padding holes)
    assert C.sizeof(_lib.DrinBatchC) == 8 * len(_lib.DrinBatchC._fields_)


def test_activation_ids_are_validated_on_host():
    lib = _lib.load()
    c = _lib.DrinConfigC()
    lib.drin_default_config(C.byref(c))
    c.batch = 4
    assert (c.vertex_activation, c.edge_activation) == (0, 0)             # DRIN_ACT_DEFAULT: gelu / sigmoid
    assert lib.drin_fused_supported(C.byref(c)) == _lib.OK
    c.vertex_activation = _lib.ACTIVATIONS["relu"]
    assert lib.drin_workspace_bytes(C.byref(c), 1) > 0
    assert lib.drin_fused_supported(C.byref(c)) == _lib.OK                # the folded paths take the built activations too
    plain = lib.drin_workspace_bytes(C.byref(c), 1)
    c.edge_activation = _lib.ACTIVATIONS["silu"]                          # no derivative-from-output: the forward keeps the pre-activation
    kept = lib.drin_workspace_bytes(C.byref(c), 1)
    assert kept >= plain + 4 * 4 * c.batch * c.num_candidates             # one [4][B N] fp32 per layer that updates its edges
    assert lib.drin_workspace_bytes(C.byref(c), 0) == lib.drin_workspace_bytes(C.byref(c), 0)
    c.edge_activation = 17
    assert lib.drin_workspace_bytes(C.byref(c), 1) == 0 and b"edge_activation" in lib.drin_last_error()
    c.edge_activation, c.vertex_activation = 0, 17
    assert lib.drin_workspace_bytes(C.byref(c), 1) == 0


def test_host_selftest_layout_scratch_and_empty_reductions():
    """`drin_host_selftest` (runs under the ASan / UBSan build through the test below): the slice scratch of the mention-sized
    weight-gradient group covers its worst case at every batch size (exact fp32 at 1024 < B N <= 2048 got none),
    and a grouped GEMM launch handed an item with an empty reduction - what round 3's staged backward once did, SIGFPE on the
    host - answers DRIN_E_SHAPE instead of dividing by the slice length it would derive from it."""
    lib = _lib.load()
    assert lib.drin_host_selftest() == _lib.OK, lib.drin_last_error()
    # the window ADVICE r3 names, through the public size query: WikiMEL widths, exact fp32, B = 16 (M = 1 616 <= 2 048 < 2 M):
    # the entity side's single-type products (dW_h of the top layer, the two entity encoders) store four slices each
    c = _lib.DrinConfigC()
    lib.drin_default_config(C.byref(c))
    c.batch, c.num_candidates, c.entity_tokens, c.precision = 16, 101, 8, _lib.PREC_F32
    D, R = c.embed_dim, c.image_dim
    assert lib.drin_workspace_bytes(C.byref(c), 1) >= 4 * 4 * (5 * D * D + D * R)


def test_adam_entry_point_validates_on_host():
    lib = _lib.load()
    one = C.c_void_p(16)
    args = (0.1, 0.999, 0.001, 1.0, 1e-8, -1e-3, None)
    assert lib.drin_adam_step(None, one, one, one, 8, *args) == _lib.E_NULL
    assert lib.drin_adam_step(one, one, one, one, -1, *args) == _lib.E_SHAPE
    assert lib.drin_adam_step(one, one, one, one, 0, *args) == _lib.OK    # nothing to do, nothing launched
    assert lib.drin_adam_step(C.c_void_p(20), one, one, one, 8, *args) == _lib.E_ALIGN
    assert lib.drin_adam_step(one, one, one, one, 8, 0.75, 0.999, 0.001, 1.0, 1e-8, -1e-3, None) == _lib.E_UNSUPPORTED


@pytest.mark.skipif(os.environ.get("DRIN_LIB_PATH") is not None, reason="already running against another build of the library")
def test_host_side_under_address_and_ub_sanitizers(tmp_path):
    """an ASan / UBSan build of the HOST side of the C-ABI library (argument validation, workspace
    layouts, error strings; device code uninstrumented, no GPU needed): every host-only test of this module passes against
    it with the sanitizers armed (`python -m drin_amd.build --asan-host`)."""
    import shutil
    import sys
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    from drin_amd import build as B
    lib = B.build(asan_host=True, verbose=False)
    env = dict(os.environ, DRIN_LIB_PATH=lib, LD_PRELOAD=B.asan_runtime(),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-p", "no:cacheprovider",
                        "-k", "not sanitizers and not plain_c_abi_example"], capture_output=True, text=True, env=env, cwd=REPO)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "passed" in r.stdout and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
