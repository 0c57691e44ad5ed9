"""ctypes binding of libdrin_hip.so (include/drin_hip.h).

This is the same stub a maintainer of the reference would add to call the library from
`drin/model.py`.  The library is loaded from the package directory
(in-tree build, `python -m drin_amd.build`); if it is missing the import of the HIP path
fails loudly - there is no CPU or PyTorch fallback in the product.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# DRIN_LIB_PATH: another build of the same library (the sanitizer build of `python -m drin_amd.build --asan-host`)
LIB_PATH = os.environ.get("DRIN_LIB_PATH") or os.path.join(_HERE, "libdrin_hip.so")
MAX_LAYERS = 8
ABI_VERSION = 7

OK, E_SHAPE, E_NULL, E_ALIGN, E_WORKSPACE, E_HIP, E_UNSUPPORTED, E_INDEX = 0, -1, -2, -3, -4, -5, -6, -7
PREC_F32, PREC_BF16X3, PREC_BF16X3_ALL, PREC_BF16X3_IF16 = 0, 1, 3, 5     # (2 and 4: removed with ABI 6 - outside the 1e-4 bar)
FEAT_F32, FEAT_BF16 = 0, 1
CACHE_F32, CACHE_MIXED_F16 = 0, 1                                      # drin_cache_format
CACHE_FORMATS = {"f32": CACHE_F32, "mixed_f16": CACHE_MIXED_F16}
ACTIVATIONS = {"gelu": 1, "sigmoid": 2, "relu": 3, "tanh": 4, "silu": 5}   # drin_activation

fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int64)


class DrinConfigC(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("num_candidates", C.c_int32), ("embed_dim", C.c_int32), ("image_dim", C.c_int32),
        ("mention_tokens", C.c_int32), ("image_regions", C.c_int32), ("mention_objects", C.c_int32),
        ("entity_objects", C.c_int32), ("entity_tokens", C.c_int32), ("mention_object_inner", C.c_int32),
        ("entity_image_inner", C.c_int32), ("entity_object_inner", C.c_int32), ("num_layers", C.c_int32),
        ("dynamic_edges", C.c_int32), ("edge_enabled", C.c_float * 4), ("layer_norm_eps", C.c_float),
        ("cosine_eps", C.c_float), ("miei_eps", C.c_float), ("clip_scale", C.c_float), ("precision", C.c_int32),
        ("num_entities", C.c_int32), ("vector_edges", C.c_int32), ("feature_dtype", C.c_int32),
        ("vertex_activation", C.c_int32), ("edge_activation", C.c_int32), ("cache_format", C.c_int32),
    ]


class DrinBatchC(C.Structure):
    _fields_ = [
        ("mention_text", C.c_void_p), ("mention_start", C.c_void_p), ("mention_end", C.c_void_p),
        ("mention_image", C.c_void_p), ("mention_object", C.c_void_p), ("mention_object_score", C.c_void_p),
        ("entity_text", C.c_void_p), ("entity_text_mask", C.c_void_p), ("entity_image", C.c_void_p),
        ("entity_object", C.c_void_p), ("entity_object_score", C.c_void_p), ("miet_similarity", C.c_void_p),
        ("mtei_similarity", C.c_void_p), ("entity_index", C.c_void_p), ("entity_text_cls", C.c_void_p),
        ("index_status", C.c_void_p),
    ]


class DrinLayerParamsC(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w_h", "b_h", "w_u", "b_u", "w_v", "b_v", "ln_weight", "ln_bias", "w_m", "b_m")]


class DrinParamsC(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "w_mention_text", "b_mention_text", "w_entity_text", "b_entity_text",
        "w_mention_image", "b_mention_image", "w_entity_image", "b_entity_image")] + [("layer", DrinLayerParamsC * MAX_LAYERS)]


class DrinParamGradsC(C.Structure):  # same shape as DrinParamsC, mutable pointers
    _fields_ = DrinParamsC._fields_


class DrinTraceC(C.Structure):
    _fields_ = [(n, C.c_void_p * (MAX_LAYERS + 1)) for n in (
        "mention_text_vertex", "mention_image_vertex", "entity_text_vertex", "entity_image_vertex", "edges")]


EXPORTS = {
    # name: (restype, argtypes)
    "drin_version": (C.c_int, []),
    "drin_last_error": (C.c_char_p, []),
    "drin_build_info": (C.c_char_p, []),
    "drin_default_config": (C.c_int, [C.POINTER(DrinConfigC)]),
    "drin_host_selftest": (C.c_int, []),
    "drin_workspace_bytes": (C.c_size_t, [C.POINTER(DrinConfigC), C.c_int]),
    "drin_edges_fwd": (C.c_int, [C.POINTER(DrinConfigC), C.POINTER(DrinBatchC), C.c_void_p, C.c_void_p, C.c_void_p]),
    "drin_pool_fwd": (C.c_int, [C.POINTER(DrinConfigC), C.POINTER(DrinBatchC), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "drin_linear_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "drin_linear_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                  C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "drin_forward": (C.c_int, [C.POINTER(DrinConfigC), C.POINTER(DrinBatchC), C.POINTER(DrinParamsC), C.c_void_p,
                               C.c_size_t, C.c_void_p, C.c_int, C.POINTER(DrinTraceC), C.c_void_p]),
    "drin_forward_staged": (C.c_int, [C.POINTER(DrinConfigC), C.POINTER(DrinBatchC), C.POINTER(DrinParamsC), C.c_void_p,
                                      C.c_size_t, C.c_void_p, C.c_int, C.POINTER(DrinTraceC), C.c_void_p, C.c_void_p]),
    "drin_backward": (C.c_int, [C.POINTER(DrinConfigC), C.POINTER(DrinBatchC), C.POINTER(DrinParamsC), C.c_void_p,
                                C.c_size_t, C.c_void_p, C.POINTER(DrinParamGradsC), C.c_void_p]),
    "drin_backward_staged": (C.c_int, [C.POINTER(DrinConfigC), C.POINTER(DrinBatchC), C.POINTER(DrinParamsC), C.c_void_p,
                                       C.c_size_t, C.c_void_p, C.POINTER(DrinParamGradsC), C.c_void_p, C.c_void_p]),
    "drin_fused_supported": (C.c_int, [C.POINTER(DrinConfigC)]),
    "drin_prepared_bytes": (C.c_size_t, [C.POINTER(DrinConfigC)]),
    "drin_fused_workspace_bytes": (C.c_size_t, [C.POINTER(DrinConfigC)]),
    "drin_workgroups_per_mention": (C.c_int32, [C.POINTER(DrinConfigC), C.c_int32]),
    "drin_index_status": (C.c_int, [C.c_void_p, C.c_void_p]),
    "drin_image_contraction_passes": (C.c_int32, [C.POINTER(DrinConfigC), C.c_int32]),
    "drin_prepare": (C.c_int, [C.POINTER(DrinConfigC), C.POINTER(DrinParamsC), C.c_void_p, C.c_size_t, C.c_void_p]),
    "drin_forward_prepared": (C.c_int, [C.POINTER(DrinConfigC), C.POINTER(DrinBatchC), C.POINTER(DrinParamsC), C.c_void_p,
                                        C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "drin_entity_cache_bytes": (C.c_size_t, [C.POINTER(DrinConfigC)]),
    "drin_entity_cache_build_workspace_bytes": (C.c_size_t, [C.POINTER(DrinConfigC)]),
    "drin_cached_workspace_bytes": (C.c_size_t, [C.POINTER(DrinConfigC)]),
    "drin_build_entity_cache": (C.c_int, [C.POINTER(DrinConfigC), C.POINTER(DrinBatchC), C.POINTER(DrinParamsC), C.c_void_p,
                                          C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "drin_forward_cached": (C.c_int, [C.POINTER(DrinConfigC), C.POINTER(DrinBatchC), C.POINTER(DrinParamsC), C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "drin_split_planes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "drin_linear_planes_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "drin_loss_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "drin_triplet_topk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.POINTER(C.c_int32), C.c_int32,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "drin_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float,
                                 C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "drin_profile_begin": (C.c_int, [C.c_int]),
    "drin_profile_end": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "drin_kernel_class_name": (C.c_char_p, [C.c_int]),
}
KERNEL_CLASSES = 8


class DrinError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libdrin_hip status {status}: {message}")
        self.status = status


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """dlopen the in-tree library and bind every entry point of include/drin_hip.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the DRIN HIP path has no fallback. Build it with `python -m drin_amd.build`."
        )
    # The library works on PyTorch's device pointers and streams, so it must run on the HIP runtime PyTorch runs on: torch
    # ships its own libamdhip64 and the library needs "libamdhip64.so.7" - whichever is in the process first serves both.
    # Loaded before torch, the library pulls /opt/rocm's copy in, torch then brings its own, and the library's calls fail
    # with "no ROCm-capable device is detected" (seen: __graft_entry__.build() followed by smoke() in one process).
    try:
        import torch  # noqa: F401
    except ImportError:  # a torch-free caller (examples/score_c_abi.cpp has no Python at all): the system runtime alone
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.drin_version() != ABI_VERSION:
        raise ImportError(f"libdrin_hip ABI {lib.drin_version()} != binding {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def profile_begin(max_launches: int = 65536) -> None:
    check(load().drin_profile_begin(max_launches))


def profile_end() -> dict:
    """{class name: (gpu milliseconds, launches)} since profile_begin (launches from every thread)."""
    ms = (C.c_double * KERNEL_CLASSES)()
    n = (C.c_int64 * KERNEL_CLASSES)()
    check(load().drin_profile_end(ms, n))
    lib = load()
    return {lib.drin_kernel_class_name(k).decode(): (ms[k], n[k]) for k in range(KERNEL_CLASSES)}


def check(status: int) -> None:
    if status != OK:
        raise DrinError(status, load().drin_last_error().decode())
