# -*- coding: utf-8 -*-
"""GPU micro-benchmarks of the individual kernels (HIP-event timing, weights rotated over > 256 MB so the
Infinity Cache cannot hold them).  Prints GB/s per GEMM configuration and microseconds per small kernel."""
import sys, os, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from painlessinferenceacceleration_amd import _lib
from painlessinferenceacceleration_amd._lib import lib, check
from painlessinferenceacceleration_amd.llama_engine import rope_tables
from tests import gpu_utils as gu
from tests.gpu_utils import DEV, ptr, sp

torch.cuda.set_device(0)
NBUF = 6


def timeit(fn, iters=30, warm=5):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3     # us


def gemm_sweep(name, N, K, kind):
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(64, K, generator=g, device=DEV).to(torch.bfloat16)
    xp = gu.pack_x(x)
    nrows = 2 * N if kind == 'swiglu' else N
    wps = [torch.randn(nrows * K, generator=g, device=DEV, dtype=torch.float32).to(torch.bfloat16) for _ in range(NBUF)]
    wbytes = nrows * K * 2
    slabs = torch.zeros(8 * 64 * N, dtype=torch.float32, device=DEV)
    act = torch.zeros(64 * N, dtype=torch.bfloat16, device=DEV)
    logits = torch.zeros(64 * N, dtype=torch.bfloat16, device=DEV)
    cv = torch.zeros((N // 32) * 4 * 64, dtype=torch.float32, device=DEV)
    ci = torch.zeros((N // 32) * 4 * 64, dtype=torch.int32, device=DEV)
    res = []
    base = [(2, 1)] if kind == 'swiglu' else [(1, 1), (2, 1)] if kind == 'logits' else [(1, 1), (1, 2), (2, 1), (2, 2), (2, 4), (2, 8)]
    for rb, ks in base:
        for var in (0, 1, 2):
            rbv = rb | (var << 8)
            if kind == 'slab':
                fn = lambda i: lib.la_gemm64_slab(sp(), ptr(wps[i % NBUF]), ptr(xp), N, K, rbv, ks, ptr(slabs))
            elif kind == 'swiglu':
                fn = lambda i: lib.la_gemm64_swiglu(sp(), ptr(wps[i % NBUF]), ptr(xp), N, K, ptr(act), var)
            else:
                fn = lambda i: lib.la_gemm64_logits(sp(), ptr(wps[i % NBUF]), ptr(xp), N, K, rbv, ptr(logits), ptr(cv), ptr(ci))
            us = timeit(fn)
            res.append((rb, ks, var, us, wbytes / us / 1e3))
            print(f'{name:8s} N={N:6d} K={K:6d} rb={rb} ks={ks} var={var}: {us:8.2f} us  {wbytes / us / 1e3:8.1f} GB/s', flush=True)
    return res


def small_kernels():
    hidden, nh, nkv = 4096, 32, 32
    g = torch.Generator(device=DEV).manual_seed(2)
    h = torch.randn(64, hidden, generator=g, device=DEV).to(torch.bfloat16)
    nw = torch.ones(hidden, device=DEV, dtype=torch.bfloat16)
    xp = torch.zeros(64 * hidden, dtype=torch.bfloat16, device=DEV)
    for ns in (1, 2, 4):
        slabs = torch.randn(ns, 64, hidden, generator=g, device=DEV)
        us = timeit(lambda i: lib.la_resid_norm(sp(), ptr(h), ptr(slabs), ns, ptr(nw), hidden, 1e-5, ptr(xp)), 50)
        print(f'resid_norm n_slabs={ns}: {us:.2f} us', flush=True)
    N = (nh + 2 * nkv) * 128
    pos = torch.arange(64, device=DEV, dtype=torch.int32) + 600
    rc, rs_ = rope_tables(128, 2048, 10000.0, DEV)
    qf = torch.zeros(nh * 8192, dtype=torch.bfloat16, device=DEV)
    kf = torch.zeros(nkv * 8192, dtype=torch.bfloat16, device=DEV)
    vf = torch.zeros(nkv * 8192, dtype=torch.bfloat16, device=DEV)
    for ns in (1, 2):
        slabs = torch.randn(ns, 64, N, generator=g, device=DEV)
        us = timeit(lambda i: lib.la_qkv_post(sp(), ptr(slabs), ns, nh, nkv, ptr(pos), ptr(rc), ptr(rs_), ptr(qf), ptr(kf), ptr(vf)), 50)
        print(f'qkv_post n_slabs={ns}: {us:.2f} us', flush=True)
    max_keys = 2048
    NL = 8          # rotate over layers' caches so K/V come from HBM like in the real step
    km = torch.randn(NL, nkv * max_keys * 128, generator=g, device=DEV).to(torch.bfloat16)
    vm = torch.randn(NL, nkv * max_keys * 128, generator=g, device=DEV).to(torch.bfloat16)
    rm = torch.from_numpy(np.array([(2 << t) - 1 for t in range(63)] + [-1], dtype=np.int64)).to(DEV)
    out = torch.zeros(64 * nh * 128, dtype=torch.bfloat16, device=DEV)
    for nkeys in (512, 1024, 1984):
        state = torch.zeros(_lib.LA_ST_WORDS, dtype=torch.int32, device=DEV); state[0] = nkeys
        for nsplit in (2, 4, 8, 16):
            opart = torch.zeros(nh * nsplit * 64 * 128, dtype=torch.float32, device=DEV)
            mpart = torch.zeros(nh * nsplit * 64, dtype=torch.float32, device=DEV); lpart = torch.zeros_like(mpart)
            us = timeit(lambda i: lib.la_tree_attn(sp(), ptr(qf), ptr(km[i % NL]), ptr(vm[i % NL]), ptr(kf), ptr(vf), ptr(rm), ptr(state),
                                                   nh, nkv, max_keys, nsplit, ptr(opart), ptr(mpart), ptr(lpart), ptr(out)), 40)
            kvb = 2 * nkv * 128 * 2 * (nkeys + 64)
            print(f'tree_attn(+combine) nkeys={nkeys} nsplit={nsplit}: {us:.2f} us  ({kvb / us / 1e3:.0f} GB/s KV)', flush=True)
    # launch floor: an almost empty kernel on the same stream
    st = torch.zeros(_lib.LA_ST_WORDS, dtype=torch.int32, device=DEV)
    cv = torch.zeros(64 * 8, dtype=torch.float32, device=DEV); ci = torch.zeros(64 * 8, dtype=torch.int32, device=DEV)
    us = timeit(lambda i: lib.la_argmax_finalize(sp(), ptr(cv), ptr(ci), 8, ptr(st)), 200)
    print(f'launch floor (argmax_finalize, 64 blocks): {us:.2f} us', flush=True)


if __name__ == '__main__':
    which = sys.argv[1:] or ['gemm', 'small']
    if 'small' in which:
        small_kernels()
    if 'gemm' in which:
        gemm_sweep('qkv', 12288, 4096, 'slab')
        gemm_sweep('o', 4096, 4096, 'slab')
        gemm_sweep('down', 4096, 11008, 'slab')
        gemm_sweep('gateup', 11008, 4096, 'swiglu')
        gemm_sweep('lm_head', 32000, 4096, 'logits')
