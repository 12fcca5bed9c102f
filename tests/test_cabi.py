"""The C-ABI library loads and exports every symbol include/thewhisper.h declares; without a GPU the product path
fails loudly (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "thewhisper.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tw_[a-z_]+)\s*\(", src)))


def test_header_symbols_are_exported(built_library):
    lib = ctypes.CDLL(built_library)
    names = header_functions()
    assert {"tw_create", "tw_logmel", "tw_encode", "tw_cross_kv", "tw_decode_step", "tw_generate_greedy",
            "tw_token_timestamps"} <= set(names)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/thewhisper.h but not exported"


def test_binding_table_matches_header(built_library):
    from thewhisper_amd import _cabi

    assert sorted(n for n, _, _ in _cabi.SYMBOLS) == header_functions()
    lib = _cabi.load_library()
    assert lib.tw_version().decode().startswith("thewhisper-gfx950")


def test_struct_layouts_match_header():
    from thewhisper_amd import _cabi

    assert ctypes.sizeof(_cabi.tw_config) == 4 * (12 + 64 + 2)
    # tw_greedy_opts: 9 int32, pad to 8, ptr, int32 (+pad), ptr, 3 int32 (+pad)
    assert ctypes.sizeof(_cabi.tw_greedy_opts) == 40 + 8 + 8 + 8 + 16


def test_no_cpu_fallback(built_library):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from thewhisper_amd import _cabi
    from thewhisper_amd.engine import WhisperEngine

    lib = _cabi.load_library()
    cfg = _cabi.tw_config()
    cfg.d_model, cfg.heads, cfg.enc_layers, cfg.dec_layers, cfg.ffn, cfg.vocab, cfg.n_mels = 128, 2, 1, 1, 256, 1000, 80
    cfg.source_positions, cfg.target_positions, cfg.max_batch, cfg.dtype = 100, 448, 1, 1
    ctx = ctypes.c_void_p()
    assert lib.tw_create(ctypes.byref(cfg), ctypes.byref(ctx)) < 0
    assert b"no HIP device" in lib.tw_last_error(None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        WhisperEngine(dict(d_model=128, heads=2, enc_layers=1, dec_layers=1, ffn=256, vocab=1000, n_mels=80), 100)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "thewhisper_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
