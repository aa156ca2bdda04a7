# -*- coding: utf-8 -*-
"""CPU: liblookahead_hip.so loads and exports every symbol include/lookahead_hip.h declares (no compute calls)."""
import ctypes
import os
import re

from painlessinferenceacceleration_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols(name='lookahead_hip.h'):
    src = open(os.path.join(ROOT, 'include', name)).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(la_[a-z0-9_]+)\s*\(', src)))


def exported_symbols(path=None):
    import subprocess
    out = subprocess.run(['nm', '-D', '--defined-only', path or _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    return sorted(ln.split()[-1] for ln in out.splitlines() if ' T la_' in ln)


def test_every_declared_symbol_is_exported_and_bound():
    """The product header == the bindings == the PRODUCT library's exports, symbol for symbol: no la_lab_* entry point, no knob storage.
    The kernel lab (la_lab_*: measurement knobs and A/B switches) exists in the LAB builds of the same sources only
    (liblookahead_hip_lab.so / _lab_f16.so, -DLA_LAB=1), declared by its own header."""
    syms = header_symbols()
    assert len(syms) >= 35
    dll = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(dll, s), f'{s} declared in the header but not exported'
        assert s in _lib.PROTOTYPES, f'{s} has no ctypes prototype in _lib.py'
    assert sorted(_lib.PROTOTYPES) == syms
    lab = header_symbols('lookahead_hip_lab.h')
    assert lab == sorted(_lib.LAB_PROTOTYPES) == ['la_lab_get', 'la_lab_set', 'la_lab_set_ptr']
    assert not any(s.startswith('la_lab_') for s in syms), 'the product header does not declare the lab'
    assert exported_symbols() == sorted(syms)                # nothing exported that the product header does not declare: no lab
    # the float16 build (the same sources with -DLA_DTYPE=1) exports the same ABI and reports its dtype
    assert exported_symbols(_lib.LIB_PATH_F16) == sorted(syms)
    # the lab builds: the product ABI + the three lab entry points
    assert exported_symbols(_lib.LAB_PATH) == sorted(syms + lab) == exported_symbols(_lib.LAB_PATH_F16)
    assert not _lib.LAB_BUILD, 'the test suite runs on the product libraries (LA_LAB_BUILD is for the A/B scripts)'
    import subprocess
    knob_syms = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert 'g_la_mb_pair' not in knob_syms and 'g_la_pf_kib' not in knob_syms, 'product knobs are constexpr defaults (csrc/la_knobs.h)'
    import torch
    lib16 = _lib.lib_for(torch.float16)
    assert (lib16.la_abi_dtype(), _lib.lib.la_abi_dtype()) == (_lib.LA_DTYPE_F16, _lib.LA_DTYPE_BF16) and lib16.la_abi_version() == _lib.ABI_VERSION


def test_product_debug_key_is_the_depth_probe_only():
    lib = _lib.lib
    assert not hasattr(lib, 'la_lab_get') and not hasattr(lib, 'la_lab_set')
    assert lib.la_debug_get(13) == 0 and lib.la_debug_set(13, 5) == 0 and lib.la_debug_get(13) == 5
    assert lib.la_debug_set(13, 0) == 0
    for key in (0, 6, 7, 10, 17, 19, 99):
        assert lib.la_debug_set(key, 0) == -1 and lib.la_debug_get(key) == -1      # LA_E_ARG: lab knobs are not reachable here


def test_abi_version_and_error_channel():
    assert _lib.lib.la_abi_version() == _lib.ABI_VERSION
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'lookahead_hip.h')).read()
    assert int(re.search(r'#define LA_ABI_VERSION\s+(\d+)', hdr).group(1)) == _lib.ABI_VERSION
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


def test_rowplan_covers_every_row_exactly_once():
    """la_rowplan (host function): balanced packing plans for the Llama-2-7B / 13B shapes and small odd ones."""
    import numpy as np
    lib = _lib.lib
    for kind, n_rows, nwg in [(0, 32000, 256), (1, 11008, 256), (2, 12288, 256), (1, 13824, 256), (2, 15360, 256),
                              (0, 1000, 8), (1, 688, 16), (2, 5 * 128, 16),
                              (2, 6144, 96), (2, 6144, 128)]:      # the multi-block step's second QKV image (Mistral: 96 x 32 pairs)
        n = lib.la_rowplan(kind, n_rows, nwg, None)
        assert n > 0 and n % 32 == 0
        plan = np.zeros(n, dtype=np.int32)
        assert lib.la_rowplan(kind, n_rows, nwg, plan.ctypes.data_as(_lib.pi32)) == n
        valid = plan[plan >= 0]
        total = n_rows * (2 if kind == 1 else 1)
        assert sorted(valid.tolist()) == list(range(total))
        blocks = plan.reshape(-1, 32)
        # inside a block the valid rows come first (lanes of invalid rows re-read the last valid one)
        for b in blocks:
            nv = int((b >= 0).sum())
            assert nv >= 1 and (b[:nv] >= 0).all() and (b[nv:] < 0).all()
    # qkv plan: block pairs hold RoPE partners (d, d+64) of the same head slot
    n = lib.la_rowplan(2, 12288, 256, None)
    plan = np.zeros(n, dtype=np.int32); lib.la_rowplan(2, 12288, 256, plan.ctypes.data_as(_lib.pi32))
    b = plan.reshape(256, 2, 32)
    ok = b[:, 0, :] >= 0
    assert ((b[:, 1, :] - b[:, 0, :])[ok] == 64).all() and ((b[:, 0, :][ok] % 128) < 64).all()
    # shapes that cannot be dealt out evenly are refused (the engine then uses the classic grid)
    assert lib.la_rowplan(1, 11000, 256, None) < 0 and lib.la_rowplan(0, 32000, 7, None) < 0


def test_qkv_row_perm_is_a_permutation_of_rope_pairs():
    import numpy as np
    perm = np.zeros((4 + 2 * 2) * 128, dtype=np.int32)
    assert _lib.lib.la_qkv_row_perm(4, 2, perm.ctypes.data_as(_lib.pi32)) == 0
    assert sorted(perm.tolist()) == list(range(len(perm)))
    b = perm.reshape(-1, 2, 32)
    assert ((b[:, 1, :] - b[:, 0, :]) == 64).all()


def test_head_lane_map_keeps_rotary_partners_64_lanes_apart():
    """la_head_lane_map: a head of head_dim < 128 features inside its 128-feature lane — feature d < hd/2 in lane d, its rotary partner
    d + hd/2 (rotate_half, modeling_llama.py:146-151) in lane 64 + d, -1 (zero padding) elsewhere; the identity at 128."""
    import numpy as np
    for hd in (8, 32, 64, 96, 128):
        m = np.zeros(128, dtype=np.int32)
        assert _lib.lib.la_head_lane_map(hd, m.ctypes.data_as(_lib.pi32)) == 0
        used = m[m >= 0]
        assert sorted(used.tolist()) == list(range(hd))
        for d in range(hd // 2):
            assert m[d] == d and m[64 + d] == hd // 2 + d
        assert (m[hd // 2:64] == -1).all() and (m[64 + hd // 2:] == -1).all()
    assert _lib.lib.la_head_lane_map(130, m.ctypes.data_as(_lib.pi32)) != 0 and _lib.lib.la_head_lane_map(63, m.ctypes.data_as(_lib.pi32)) != 0


def test_debug_knobs_roundtrip_and_defaults():
    """la_lab_set / la_lab_get: every knob reads back, out-of-range values are refused, and the library defaults are the
    documented ones (everything 0 except key 6 = 12657: the paired wide launches + (round 5) the fat-wave forms of gate/up and of the paired slab / QKV launches, and key 11 = 1 step per graph; round 3: 13 = depth probe,
    14 = split head / tail kernels, 15 = 4-wave GEMM variants; round 4: 17 = single-launch tree attention, default ON).  Round 6: the
    knobs live in the LAB build only; its defaults are the product's constexprs (one table, csrc/la_knobs.h)."""
    import torch
    lib = _lib.lab_lib_for(torch.bfloat16)
    import re
    table = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'painlessinferenceacceleration_amd', 'csrc', 'la_knobs.h')).read()
    rows = re.findall(r'X\((g_la_\w+),\s*(-?\d+),\s*(\d+),', table)
    assert len(rows) >= 29
    for _, dflt, key in rows:
        assert lib.la_lab_get(int(key)) == int(dflt), key
    defaults = {0: 0, 1: 0, 2: 0, 3: 0, 4: 0, 5: 0, 6: 12657, 7: 0, 8: 0, 9: 0, 10: 0, 11: 1, 12: 0, 13: 0, 14: 0, 15: 0, 16: 0, 17: 1, 18: 0, 19: 0, 20: 0, 21: 0, 22: 0, 23: 0, 24: 1, 25: 13}
    for key, d in defaults.items():
        assert lib.la_lab_get(key) == d, key
    try:
        for key, ok, bad in ((6, 32767, 32768), (16, 7, 8), (14, 1, 2), (15, 7, 8), (7, 128, 129), (8, 16, 17), (9, 64, 65), (10, 1, 2), (11, 8, 9), (17, 0, 2)):
            assert lib.la_lab_set(key, ok) == 0 and lib.la_lab_get(key) == ok
            assert lib.la_lab_set(key, bad) == -1 and lib.la_lab_get(key) == ok          # LA_E_ARG, value kept
        assert lib.la_lab_get(99) == -1 and lib.la_lab_set(99, 0) == -1
    finally:
        for key, d in defaults.items():
            lib.la_lab_set(key, d)
