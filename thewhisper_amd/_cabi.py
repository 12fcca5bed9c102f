"""ctypes binding of include/thewhisper.h (the C ABI of libthewhisper_gfx950.so).

This is the stub a maintainer of the reference would add under ``thestage_speechkit/amd/`` (see
INTEGRATION.md).  No CPU fallback exists: if the library is missing or a call fails, a
``RuntimeError`` is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

from . import build as _build

TW_F32, TW_BF16, TW_F16 = 0, 1, 2
TW_BF16_MXFP8 = 3  # context dtype only: bf16 activations, MXFP8 decoder projection weights, activations quantised in registers (W8A8)
TW_BF16_W8A16 = 4  # context dtype only: MXFP8 decoder projection weights widened to bf16 in registers, bf16 activations
TW_MAX_ALIGN_HEADS = 32


class tw_config(C.Structure):
    _fields_ = [
        ("d_model", C.c_int32), ("enc_layers", C.c_int32), ("dec_layers", C.c_int32), ("heads", C.c_int32),
        ("ffn", C.c_int32), ("vocab", C.c_int32), ("n_mels", C.c_int32), ("source_positions", C.c_int32),
        ("target_positions", C.c_int32), ("max_batch", C.c_int32), ("dtype", C.c_int32),
        ("n_align_heads", C.c_int32), ("align_heads", C.c_int32 * (2 * TW_MAX_ALIGN_HEADS)),
        ("device", C.c_int32), ("use_graph", C.c_int32),
    ]


class tw_greedy_opts(C.Structure):
    _fields_ = [
        ("eos_id", C.c_int32), ("pad_id", C.c_int32), ("max_new_tokens", C.c_int32), ("min_new_tokens", C.c_int32),
        ("max_length", C.c_int32), ("timestamps", C.c_int32), ("no_timestamps_id", C.c_int32),
        ("max_initial_timestamp_index", C.c_int32), ("n_begin_suppress", C.c_int32),
        ("begin_suppress", C.POINTER(C.c_int32)), ("n_suppress", C.c_int32), ("suppress", C.POINTER(C.c_int32)),
        ("want_alignment", C.c_int32), ("n_forced", C.c_int32), ("n_draft", C.c_int32),
    ]


# every symbol declared in include/thewhisper.h: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("tw_version", C.c_char_p, []),
    ("tw_last_error", C.c_char_p, [_P]),
    ("tw_create", C.c_int, [C.POINTER(tw_config), C.POINTER(_P)]),
    ("tw_destroy", C.c_int, [_P]),
    ("tw_create_sibling", C.c_int, [_P, C.c_int32, C.POINTER(_P)]),
    ("tw_load_weight", C.c_int, [_P, C.c_char_p, _P, C.c_int32, C.c_int32, C.POINTER(C.c_int64), _P]),
    ("tw_finalize_weights", C.c_int, [_P, _P]),
    ("tw_logmel", C.c_int, [_P, _P, C.c_int64, C.POINTER(C.c_int32), C.c_int32, C.c_int32, _P, C.c_int32, _P]),
    ("tw_encode", C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, C.c_int32, _P]),
    ("tw_cross_kv", C.c_int, [_P, C.c_int32, _P]),
    ("tw_encode_at", C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    ("tw_cross_kv_at", C.c_int, [_P, C.c_int32, C.c_int32, _P]),
    ("tw_adopt_cross_kv", C.c_int, [_P, C.c_int32, _P, C.c_int32, C.c_int32, _P, _P]),
    ("tw_decoder_reset", C.c_int, [_P, C.c_int32, _P]),
    ("tw_decode_step", C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32), _P, _P]),
    ("tw_generate_greedy", C.c_int, [_P, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.POINTER(tw_greedy_opts),
                                     C.POINTER(C.c_int32), C.POINTER(C.c_int32), _P]),
    ("tw_last_draft", C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("tw_token_timestamps", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_double,
                                      C.POINTER(C.c_float), _P]),
    ("tw_get_alignment", C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(C.c_float), _P]),
    ("tw_last_timings", C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    ("tw_vad_energy", C.c_int, [C.c_int32, _P, C.c_int64, C.c_int32, C.c_int32, _P, _P, _P]),
    ("tw_stream_create_masked", C.c_int, [C.c_int32, C.POINTER(C.c_uint32), C.c_int32, C.POINTER(_P)]),
    ("tw_stream_destroy", C.c_int, [_P]),
    ("tw_stream_synchronize", C.c_int, [_P]),
    ("tw_event_create", C.c_int, [C.c_int32, C.POINTER(_P)]),
    ("tw_event_destroy", C.c_int, [_P]),
    ("tw_event_record", C.c_int, [_P, _P]),
    ("tw_stream_wait_event", C.c_int, [_P, _P]),
]

_lib: Optional[C.CDLL] = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen the in-tree library and bind every symbol; raises if anything is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # torch bundles its own HIP runtime: import it FIRST so that libthewhisper's libamdhip64 dependency resolves to the
    # runtime torch already mapped (one runtime per process - device pointers and streams are shared with torch).
    import torch  # noqa: F401

    p = path or os.environ.get("THEWHISPER_LIB") or _build.library_path()
    if not os.path.exists(p):
        raise RuntimeError(
            f"{p} not found: run `python __graft_entry__.py` (hipcc --offload-arch=gfx950) first. "
            "thewhisper_amd has no CPU fallback."
        )
    lib = C.CDLL(p)
    for name, restype, argtypes in SYMBOLS:
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise RuntimeError(f"{p} does not export {name}") from e
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(lib: C.CDLL, ctx, rc: int, what: str) -> None:
    if rc != 0:
        msg = lib.tw_last_error(ctx)
        raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")
