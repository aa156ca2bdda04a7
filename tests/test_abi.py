# -*- coding: utf-8 -*-
"""CPU: liblookahead_hip.so loads and exports every symbol include/lookahead_hip.h declares (no compute calls)."""
import ctypes
import os
import re

from painlessinferenceacceleration_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, 'include', 'lookahead_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(la_[a-z0-9_]+)\s*\(', src)))


def test_every_declared_symbol_is_exported_and_bound():
    syms = header_symbols()
    assert len(syms) >= 35
    dll = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(dll, s), f'{s} declared in the header but not exported'
        assert s in _lib.PROTOTYPES, f'{s} has no ctypes prototype in _lib.py'
    assert sorted(_lib.PROTOTYPES) == syms


def test_abi_version_and_error_channel():
    assert _lib.lib.la_abi_version() == 1
    c = _lib.lib.la_cache_create(10, 10)
    assert c
    rc = _lib.lib.la_cache_put(c, None, 3, 8, 0, 7, 0)      # bad mode / null tokens -> LA_E_ARG, never a crash
    assert rc == -1
    _lib.lib.la_cache_destroy(c)


def test_device_entry_points_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        return
    cfg = _lib.LlamaConfigC()
    cfg.n_layers, cfg.hidden, cfg.n_heads, cfg.n_kv_heads, cfg.head_dim = 1, 256, 2, 2, 128
    cfg.ffn, cfg.vocab, cfg.max_keys, cfg.max_pos, cfg.rms_eps = 512, 512, 128, 256, 1e-5
    assert _lib.lib.la_llama_workspace_bytes(ctypes.byref(cfg)) > 0
    w = _lib.LlamaWeightsC()
    layers = (_lib.LlamaLayerWeightsC * 1)()
    w.layers = ctypes.cast(layers, ctypes.POINTER(_lib.LlamaLayerWeightsC))
    buf = ctypes.create_string_buffer(16)
    h = _lib.lib.la_llama_create(ctypes.byref(cfg), ctypes.byref(w), buf, 1 << 40)
    assert not h and 'no HIP device' in _lib.last_error()
    from painlessinferenceacceleration_amd.llama_engine import LlamaVerifyEngine
    import pytest
    with pytest.raises(RuntimeError):
        LlamaVerifyEngine(None, {}, device='cuda:0')
