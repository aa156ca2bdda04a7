# -*- coding: utf-8 -*-
"""Helpers for the -m gpu tests: HBM layout index maps (mirrors of csrc/la_common.h) and ctypes plumbing."""
import contextlib
import ctypes as C

import numpy as np
import torch

from oracle.tiny import random_tree  # noqa: F401  (build-free recipe shared with the golden generators)
from painlessinferenceacceleration_amd import _lib
from painlessinferenceacceleration_amd._lib import lib, check

DEV = 'cuda:0'


def sp():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def xp_index(K):
    t = np.arange(64)[:, None]
    k = np.arange(K)[None, :]
    return ((k >> 4) * 1024 + (t >> 5) * 512 + ((t & 31) + 32 * ((k >> 3) & 1)) * 8 + (k & 7)).astype(np.int64)


def rf_index(rows, d=128):
    r = np.arange(rows)[:, None]
    c = np.arange(d)[None, :]
    return (((r >> 5) * 8 + (c >> 4)) * 512 + ((r & 31) + 32 * ((c >> 3) & 1)) * 8 + (c & 7)).astype(np.int64)


def vf_index(keys, d=128):
    key = np.arange(keys)[:, None]
    c = np.arange(d)[None, :]
    kk = key & 31
    s2, rr = kk >> 4, kk & 15
    hh, e = (rr >> 2) & 1, (rr & 3) + 4 * (rr >> 3)
    return (((key >> 5) * 8 + (c >> 5) * 2 + s2) * 512 + ((c & 31) + 32 * hh) * 8 + e).astype(np.int64)


def to_packed(dense, index, size=None):
    """dense tensor [..., R, C] -> flat packed tensor using an index map [R, C]."""
    idx = torch.from_numpy(index).to(dense.device)
    size = int(index.max()) + 1 if size is None else size
    lead = dense.shape[:-2]
    out = torch.zeros(lead + (size,), dtype=dense.dtype, device=dense.device)
    out[..., idx.reshape(-1)] = dense.reshape(lead + (-1,))
    return out


def from_packed(flat, index):
    idx = torch.from_numpy(index).to(flat.device)
    return flat[..., idx.reshape(-1)].reshape(flat.shape[:-1] + tuple(index.shape))


def pack_weight(w, w2=None):
    n, k = w.shape
    out = torch.empty((2 if w2 is not None else 1) * n * k, dtype=torch.bfloat16, device=w.device)
    check(lib.la_pack_weight(sp(), ptr(w), ptr(w2), n, k, 1 if w2 is not None else 0, ptr(out)), 'pack_weight')
    return out


def pack_x(x):
    out = torch.empty_like(x).reshape(-1)
    check(lib.la_pack_x(sp(), ptr(x), x.shape[1], ptr(out)), 'pack_x')
    return out


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def pack_planned(kind, mats, n_wg):
    """Balanced packing: gather rows by la_rowplan (-1 -> zero row), then la_pack_weight."""
    n_rows = mats[0].shape[0]
    n = lib.la_rowplan(kind, n_rows, n_wg, None)
    assert n > 0, 'shape cannot be balanced'
    plan = np.zeros(n, dtype=np.int32)
    assert lib.la_rowplan(kind, n_rows, n_wg, plan.ctypes.data_as(_lib.pi32)) == n
    d_plan = torch.from_numpy(plan).to(mats[0].device)
    mats = [m.contiguous() for m in mats]
    K = mats[0].shape[1]
    out = torch.empty(lib.la_planned_elems(kind, n_rows, K, n_wg), dtype=torch.bfloat16, device=mats[0].device)
    check(lib.la_pack_planned(sp(), ptr(mats[0]), ptr(mats[1]) if len(mats) > 1 else None, ptr(d_plan), kind, n_rows, K,
                              n_wg, ptr(out)), 'pack_planned')
    torch.cuda.synchronize()
    return out


@contextlib.contextmanager
def debug_knob(key, value):
    """la_lab_set(key, value) for the duration of a block (capture-time knobs re-capture the step graphs on both edges)."""
    old = _lib.lab_get(key)
    check(_lib.lab_set(key, value), 'lab_set')          # every loaded LAB build: the engines of a variant test are created under the lab_build fixture
    try:
        yield
    finally:
        _lib.lab_set(key, old)


@contextlib.contextmanager
def split_attention():
    """The single-sequence step on the kernels the cursor-batch step runs — key-split attention + combine launches (la_debug_set
    key 17 = 0) and one workgroup per norm row (key 19 = 0) — for tests that compare two paths BITWISE; the defaults (single-launch
    attention, four workgroups per norm row) sum keys / squares in another order."""
    with debug_knob(17, 0), debug_knob(19, 0):
        yield
