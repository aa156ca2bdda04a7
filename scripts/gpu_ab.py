import os; os.environ.setdefault('LA_LAB_BUILD', '1')      # A/B script: the lab build (kernel-lab knobs, phase stamps) is the process library
# -*- coding: utf-8 -*-
"""Kernel-level A/B on one MI355X: HIP-event time of every kernel class of the verify step at the Llama-2-7B layer
shape (weights rotated over > 256 MB so the Infinity Cache cannot hold them), each GEMM also with the epilogue switched
off (la_lab_set(0, 1)) so that the streaming loop and the reduction/epilogue tail are separable.

    python scripts/gpu_ab.py [gemm] [small]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from painlessinferenceacceleration_amd import _lib                      # noqa: E402
from painlessinferenceacceleration_amd._lib import lib, check           # noqa: E402
from painlessinferenceacceleration_amd.llama_engine import rope_tables  # noqa: E402
from tests import gpu_utils as gu                                       # noqa: E402
from tests.gpu_utils import DEV, ptr, sp                                # noqa: E402

torch.cuda.set_device(0)
NBUF = 4
NWG = torch.cuda.get_device_properties(0).multi_processor_count


def timeit(fn, iters=40, warm=6):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3     # us


def both(name, fn, wbytes):
    """time fn normally and with the epilogue disabled"""
    out = []
    for noepi in (0, 1):
        check(lib.la_lab_set(0, noepi), 'debug_set')
        out.append(timeit(fn))
    check(lib.la_lab_set(0, 0), 'debug_set')
    full, loop = out
    print(f'{name:34s} full {full:7.2f} us ({wbytes / full / 1e3:7.1f} GB/s)   loop only {loop:7.2f} us '
          f'({wbytes / loop / 1e3:7.1f} GB/s)   tail {full - loop:5.2f} us', flush=True)
    return full, loop


def timeline(name, fn, n_wg, n_waves, reps=6):
    """Per-wave wall-clock stamps of ONE launch (the kernels write wall_clock64() at entry / end of the streaming loop /
    exit): dispatch skew, spread of the loop ends (CU imbalance) and the length of the epilogue."""
    buf = torch.zeros(n_wg * n_waves * 8, dtype=torch.int64, device=DEV)
    check(lib.la_lab_set_ptr(0, ptr(buf)), 'debug_set_ptr')
    for i in range(reps):
        fn(i)
    torch.cuda.synchronize()
    check(lib.la_lab_set_ptr(0, None), 'debug_set_ptr')
    raw = buf.cpu().numpy().reshape(n_wg, n_waves, 8)
    t = raw.astype(np.float64)
    rate = 100.0          # wall_clock64 ticks per us (100 MHz constant clock)
    t0, t1, t2 = t[:, :, 0], t[:, :, 1], t[:, :, 2]
    base = t0.min()

    def pct(a):
        a = (a.reshape(-1) - base) / rate
        return ' '.join(f'{np.percentile(a, q):6.2f}' for q in (0, 10, 50, 90, 100))
    print(f'-- timeline {name}: us since the first wave started, percentiles 0/10/50/90/100 over {n_wg} WGs x {n_waves} waves')
    print(f'   entry      {pct(t0)}')
    print(f'   loop end   {pct(t1)}')
    print(f'   exit       {pct(t2)}')
    dur = (t1 - t0) / rate
    per_xcd = [dur[x::8].mean() for x in range(8)]
    print('   loop duration per wave: mean %.2f  min %.2f  max %.2f ; by XCD (wg %% 8): %s' %
          (dur.mean(), dur.min(), dur.max(), ' '.join(f'{v:.2f}' for v in per_xcd)))
    mid = (t[:, :, 3] - t0) / rate
    print('   by wave index: loop duration ' + ' '.join(f'{dur[:, w].mean():.2f}' for w in range(n_waves)) +
          ' | first half ' + ' '.join(f'{mid[:, w].mean():.2f}' for w in range(n_waves)))
    hw = raw[:, :, 4]
    simd = (hw >> 4) & 3
    wslot = hw & 15
    print('   by SIMD id: ' + ' '.join(f'{s_}:{dur[simd == s_].mean():.2f}(n={int((simd == s_).sum())})' for s_ in range(4)))
    print('   by hw wave slot: ' + ' '.join(f'{k}:{dur[wslot == k].mean():.2f}' for k in sorted(set(wslot.reshape(-1).tolist()))))
    # within a SIMD of one WG: the wave that entered first vs second
    first, second = [], []
    for w_ in range(n_wg):
        for s_ in range(4):
            idx = [k for k in range(n_waves) if simd[w_, k] == s_]
            if len(idx) == 2:
                a_, b_ = sorted(idx, key=lambda k: t0[w_, k])
                first.append(dur[w_, a_]); second.append(dur[w_, b_])
    if first:
        print(f'   two waves on one SIMD: earlier-entered {np.mean(first):.2f} us, later-entered {np.mean(second):.2f} us (pairs {len(first)})')
    wg_end = (t1.max(axis=1) - base) / rate
    order = np.argsort(wg_end)
    print('   slowest WGs (id:loop-end us): ' + ' '.join(f'{i}:{wg_end[i]:.2f}' for i in order[-8:]), flush=True)


def gemms(hidden=4096, ffn=11008, nh=32, nkv=32, vocab=32000):
    g = torch.Generator(device=DEV).manual_seed(1)

    def rnd(n, k):
        return (torch.randn(n, k, generator=g, device=DEV, dtype=torch.float32) * 0.05).to(torch.bfloat16)
    x = rnd(64, hidden)
    xp = gu.pack_x(x)
    # --- gate/up (balanced, fused SwiGLU)
    wps = [gu.pack_planned(1, [rnd(ffn, hidden), rnd(ffn, hidden)], NWG) for _ in range(NBUF)]
    act = torch.zeros(64 * ffn, dtype=torch.bfloat16, device=DEV)
    both('gate/up  k_gemm64r<4,SWIGLU,4,8>', lambda i: lib.la_gemm64r_swiglu(sp(), ptr(wps[i % NBUF]), ptr(xp), ffn, hidden, NWG, ptr(act)),
         2 * ffn * hidden * 2)
    timeline('gate/up', lambda i: lib.la_gemm64r_swiglu(sp(), ptr(wps[i % NBUF]), ptr(xp), ffn, hidden, NWG, ptr(act)), NWG, 8)
    del wps
    # --- qkv (balanced, RoPE + fragment epilogue)
    N = (nh + 2 * nkv) * 128
    wps = [gu.pack_planned(2, [rnd(N, hidden)], NWG) for _ in range(NBUF)]
    pos = torch.arange(64, device=DEV, dtype=torch.int32) + 600
    rc, rs_ = rope_tables(128, 2048, 10000.0, DEV)
    qf = torch.zeros(nh * 8192, dtype=torch.bfloat16, device=DEV)
    kf = torch.zeros(nkv * 8192, dtype=torch.bfloat16, device=DEV)
    vf = torch.zeros(nkv * 8192, dtype=torch.bfloat16, device=DEV)
    both('qkv      k_gemm64r<2,QKV,8,8>', lambda i: lib.la_gemm64r_qkv(sp(), ptr(wps[i % NBUF]), ptr(xp), nh, nkv, hidden, NWG, ptr(pos), ptr(rc),
                                                                       ptr(rs_), ptr(qf), ptr(kf), ptr(vf)), N * hidden * 2)
    timeline('qkv', lambda i: lib.la_gemm64r_qkv(sp(), ptr(wps[i % NBUF]), ptr(xp), nh, nkv, hidden, NWG, ptr(pos), ptr(rc), ptr(rs_), ptr(qf),
                                                 ptr(kf), ptr(vf)), NWG, 8)
    del wps
    # --- o_proj / down_proj (split-K 4 slabs)
    slabs = torch.zeros(8 * 64 * hidden, dtype=torch.float32, device=DEV)
    for name, n, k in (('o_proj   k_gemm64<2,SLAB,8,4> ks4', hidden, nh * 128), ('down     k_gemm64<2,SLAB,8,4> ks4', hidden, ffn)):
        wps = [gu.pack_weight(rnd(n, k)) for _ in range(NBUF * (3 if k <= 4096 else 1))]
        xk = gu.pack_x(rnd(64, k))
        nb = len(wps)
        both(name, lambda i: lib.la_gemm64_slab(sp(), ptr(wps[i % nb]), ptr(xk), n, k, 2, 4, ptr(slabs)), n * k * 2)
        timeline(name.split()[0], lambda i: lib.la_gemm64_slab(sp(), ptr(wps[i % nb]), ptr(xk), n, k, 2, 4, ptr(slabs)), (n // 64) * 4, 4)
        del wps
    # --- lm_head
    wps = [gu.pack_planned(0, [rnd(vocab, hidden)], NWG) for _ in range(NBUF)]
    logits = torch.zeros(64 * vocab, dtype=torch.bfloat16, device=DEV)
    cv = torch.zeros(NWG * 8 * 64, dtype=torch.float32, device=DEV)
    ci = torch.zeros(NWG * 8 * 64, dtype=torch.int32, device=DEV)
    both('lm_head  k_gemm64r<4,LOGITS,4,8>', lambda i: lib.la_gemm64r_logits(sp(), ptr(wps[i % NBUF]), ptr(xp), vocab, hidden, NWG, ptr(logits),
                                                                            ptr(cv), ptr(ci)), vocab * hidden * 2)


def sweep(hidden=4096, ffn=11008, nh=32, nkv=32, vocab=32000):
    """K-skew of the 8-wave kernels (la_debug_set key 1) and the 8-wave variants of the split-K slab GEMMs."""
    g = torch.Generator(device=DEV).manual_seed(1)

    def rnd(n, k):
        return (torch.randn(n, k, generator=g, device=DEV, dtype=torch.float32) * 0.05).to(torch.bfloat16)
    xp = gu.pack_x(rnd(64, hidden))
    wps = [gu.pack_planned(1, [rnd(ffn, hidden), rnd(ffn, hidden)], NWG) for _ in range(NBUF)]
    act = torch.zeros(64 * ffn, dtype=torch.bfloat16, device=DEV)
    for ks in (0, 36, 40, 42, 44, 46, 48):
        check(lib.la_lab_set(1, ks), 'kskew')
        us = timeit(lambda i: lib.la_gemm64r_swiglu(sp(), ptr(wps[i % NBUF]), ptr(xp), ffn, hidden, NWG, ptr(act)))
        print(f'gate/up kskew={ks:2d}: {us:7.2f} us  {2 * ffn * hidden * 2 / us / 1e3:7.1f} GB/s', flush=True)
    check(lib.la_lab_set(1, 44), 'kskew')
    timeline('gate/up kskew=44', lambda i: lib.la_gemm64r_swiglu(sp(), ptr(wps[i % NBUF]), ptr(xp), ffn, hidden, NWG, ptr(act)), NWG, 8)
    del wps
    N = (nh + 2 * nkv) * 128
    wps = [gu.pack_planned(2, [rnd(N, hidden)], NWG) for _ in range(NBUF)]
    pos = torch.arange(64, device=DEV, dtype=torch.int32) + 600
    rc, rs_ = rope_tables(128, 2048, 10000.0, DEV)
    qf = torch.zeros(nh * 8192, dtype=torch.bfloat16, device=DEV)
    kf = torch.zeros(nkv * 8192, dtype=torch.bfloat16, device=DEV)
    vf = torch.zeros(nkv * 8192, dtype=torch.bfloat16, device=DEV)
    for ks in (0, 36, 40, 42, 44, 46, 48):
        check(lib.la_lab_set(1, ks), 'kskew')
        us = timeit(lambda i: lib.la_gemm64r_qkv(sp(), ptr(wps[i % NBUF]), ptr(xp), nh, nkv, hidden, NWG, ptr(pos), ptr(rc), ptr(rs_), ptr(qf),
                                                 ptr(kf), ptr(vf)))
        print(f'qkv     kskew={ks:2d}: {us:7.2f} us  {N * hidden * 2 / us / 1e3:7.1f} GB/s', flush=True)
    del wps
    wps = [gu.pack_planned(0, [rnd(vocab, hidden)], NWG) for _ in range(NBUF)]
    logits = torch.zeros(64 * vocab, dtype=torch.bfloat16, device=DEV)
    cv = torch.zeros(NWG * 8 * 64, dtype=torch.float32, device=DEV)
    ci = torch.zeros(NWG * 8 * 64, dtype=torch.int32, device=DEV)
    for ks in (0, 40, 44, 48):
        check(lib.la_lab_set(1, ks), 'kskew')
        us = timeit(lambda i: lib.la_gemm64r_logits(sp(), ptr(wps[i % NBUF]), ptr(xp), vocab, hidden, NWG, ptr(logits), ptr(cv), ptr(ci)))
        print(f'lm_head kskew={ks:2d}: {us:7.2f} us  {vocab * hidden * 2 / us / 1e3:7.1f} GB/s', flush=True)
    del wps
    slabs = torch.zeros(8 * 64 * hidden, dtype=torch.float32, device=DEV)
    for name, n, k in (('o_proj', hidden, nh * 128), ('down', hidden, ffn)):
        wps = [gu.pack_weight(rnd(n, k)) for _ in range(NBUF * (3 if k <= 4096 else 1))]
        xk = gu.pack_x(rnd(64, k))
        nb = len(wps)
        for var, ksplit in ((0, 4), (3, 4), (4, 4), (3, 2), (3, 8)):
            for ks in ((0,) if var == 0 else (0, 40, 44, 48)):
                check(lib.la_lab_set(1, ks), 'kskew')
                rbv = 2 | (var << 8)
                us = timeit(lambda i: lib.la_gemm64_slab(sp(), ptr(wps[i % nb]), ptr(xk), n, k, rbv, ksplit, ptr(slabs)))
                print(f'{name:6s} variant={var} ksplit={ksplit} kskew={ks:2d}: {us:7.2f} us  {n * k * 2 / us / 1e3:7.1f} GB/s', flush=True)
        del wps
    check(lib.la_lab_set(1, 0), 'kskew')


def prio(hidden=4096, ffn=11008, nh=32, nkv=32, vocab=32000):
    """s_setprio for waves 4..7 of the 8-wave kernels (la_debug_set key 2): does issue priority even out the two waves of a SIMD?"""
    g = torch.Generator(device=DEV).manual_seed(1)

    def rnd(n, k):
        return (torch.randn(n, k, generator=g, device=DEV, dtype=torch.float32) * 0.05).to(torch.bfloat16)
    xp = gu.pack_x(rnd(64, hidden))
    wps = [gu.pack_planned(1, [rnd(ffn, hidden), rnd(ffn, hidden)], NWG) for _ in range(NBUF)]
    act = torch.zeros(64 * ffn, dtype=torch.bfloat16, device=DEV)
    fn = lambda i: lib.la_gemm64r_swiglu(sp(), ptr(wps[i % NBUF]), ptr(xp), ffn, hidden, NWG, ptr(act))
    for pr in (0, 1, 2, 3, 0):
        check(lib.la_lab_set(2, pr), 'prio')
        print(f'gate/up prio_hi={pr}: {timeit(fn):7.2f} us', flush=True)
    check(lib.la_lab_set(2, 1), 'prio')
    timeline('gate/up prio_hi=1', fn, NWG, 8)
    del wps
    N = (nh + 2 * nkv) * 128
    wps = [gu.pack_planned(2, [rnd(N, hidden)], NWG) for _ in range(NBUF)]
    pos = torch.arange(64, device=DEV, dtype=torch.int32) + 600
    rc, rs_ = rope_tables(128, 2048, 10000.0, DEV)
    qf = torch.zeros(nh * 8192, dtype=torch.bfloat16, device=DEV)
    kf = torch.zeros(nkv * 8192, dtype=torch.bfloat16, device=DEV)
    vf = torch.zeros(nkv * 8192, dtype=torch.bfloat16, device=DEV)
    fq = lambda i: lib.la_gemm64r_qkv(sp(), ptr(wps[i % NBUF]), ptr(xp), nh, nkv, hidden, NWG, ptr(pos), ptr(rc), ptr(rs_), ptr(qf), ptr(kf), ptr(vf))
    for pr in (0, 1, 2, 3, 0):
        check(lib.la_lab_set(2, pr), 'prio')
        print(f'qkv     prio_hi={pr}: {timeit(fq):7.2f} us', flush=True)
    del wps
    wps = [gu.pack_planned(0, [rnd(vocab, hidden)], NWG) for _ in range(NBUF)]
    logits = torch.zeros(64 * vocab, dtype=torch.bfloat16, device=DEV)
    cv = torch.zeros(NWG * 8 * 64, dtype=torch.float32, device=DEV)
    ci = torch.zeros(NWG * 8 * 64, dtype=torch.int32, device=DEV)
    fl = lambda i: lib.la_gemm64r_logits(sp(), ptr(wps[i % NBUF]), ptr(xp), vocab, hidden, NWG, ptr(logits), ptr(cv), ptr(ci))
    for pr in (0, 1, 3, 0):
        check(lib.la_lab_set(2, pr), 'prio')
        print(f'lm_head prio_hi={pr}: {timeit(fl):7.2f} us', flush=True)
    del wps
    slabs = torch.zeros(8 * 64 * hidden, dtype=torch.float32, device=DEV)
    wps = [gu.pack_weight(rnd(hidden, nh * 128)) for _ in range(NBUF * 3)]
    nb = len(wps)
    fo = lambda i: lib.la_gemm64_slab(sp(), ptr(wps[i % nb]), ptr(xp), hidden, nh * 128, 2 | (3 << 8), 4, ptr(slabs))
    for pr in (0, 1, 3, 0):
        check(lib.la_lab_set(2, pr), 'prio')
        print(f'o_proj  prio_hi={pr}: {timeit(fo):7.2f} us', flush=True)
    check(lib.la_lab_set(2, 0), 'prio')


def small(hidden=4096, nh=32, nkv=32):
    g = torch.Generator(device=DEV).manual_seed(2)
    h = torch.randn(64, hidden, generator=g, device=DEV).to(torch.bfloat16)
    nw = torch.ones(hidden, device=DEV, dtype=torch.bfloat16)
    xp = torch.zeros(64 * hidden, dtype=torch.bfloat16, device=DEV)
    slabs = torch.randn(4, 64, hidden, generator=g, device=DEV)
    us = timeit(lambda i: lib.la_resid_norm(sp(), ptr(h), ptr(slabs), 4, ptr(nw), hidden, 1e-5, ptr(xp)), 60)
    print(f'resid_norm n_slabs=4: {us:.2f} us', flush=True)
    rc, rs_ = rope_tables(128, 4096 + 128, 10000.0, DEV)
    qf = torch.randn(nh * 8192, generator=g, device=DEV).to(torch.bfloat16)
    kf = torch.randn(nkv * 8192, generator=g, device=DEV).to(torch.bfloat16)
    vf = torch.randn(nkv * 8192, generator=g, device=DEV).to(torch.bfloat16)
    max_keys = 4096 + 64
    NL = 6
    km = torch.randn(NL, nkv * max_keys * 128, generator=g, device=DEV).to(torch.bfloat16)
    vm = torch.randn(NL, nkv * max_keys * 128, generator=g, device=DEV).to(torch.bfloat16)
    rm = torch.from_numpy(np.array([(2 << t) - 1 for t in range(63)] + [-1], dtype=np.int64)).to(DEV)
    out = torch.zeros(64 * nh * 128, dtype=torch.bfloat16, device=DEV)
    for nkeys in (640, 1984, 4032):
        state = torch.zeros(_lib.LA_ST_WORDS, dtype=torch.int32, device=DEV)
        state[0] = nkeys
        for nsplit in (4, 8, 16):
            opart = torch.zeros(nh * nsplit * 64 * 128, dtype=torch.float32, device=DEV)
            mpart = torch.zeros(nh * nsplit * 64, dtype=torch.float32, device=DEV)
            lpart = torch.zeros_like(mpart)
            us = timeit(lambda i: lib.la_tree_attn(sp(), ptr(qf), ptr(km[i % NL]), ptr(vm[i % NL]), ptr(kf), ptr(vf), ptr(rm), ptr(state),
                                                   nh, nkv, max_keys, nsplit, ptr(opart), ptr(mpart), ptr(lpart), ptr(out)), 40)
            kvb = 2 * nkv * 128 * 2 * (nkeys + 64)
            print(f'tree_attn(+combine) nkeys={nkeys} nsplit={nsplit}: {us:.2f} us  ({kvb / us / 1e3:.0f} GB/s KV)', flush=True)
    # phase stamps of the attention kernel at the bench context length
    nkeys, nsplit = 640, 8
    state = torch.zeros(_lib.LA_ST_WORDS, dtype=torch.int32, device=DEV)
    state[0] = nkeys
    opart = torch.zeros(nh * nsplit * 64 * 128, dtype=torch.float32, device=DEV)
    mpart = torch.zeros(nh * nsplit * 64, dtype=torch.float32, device=DEV)
    lpart = torch.zeros_like(mpart)
    nwg, nwv = nh * nsplit, 8
    buf = torch.zeros(nwg * nwv * 8, dtype=torch.int64, device=DEV)
    check(lib.la_lab_set_ptr(0, ptr(buf)), 'debug_set_ptr')
    for i in range(6):
        lib.la_tree_attn(sp(), ptr(qf), ptr(km[i % NL]), ptr(vm[i % NL]), ptr(kf), ptr(vf), ptr(rm), ptr(state), nh, nkv, max_keys, nsplit,
                         ptr(opart), ptr(mpart), ptr(lpart), ptr(out))
    torch.cuda.synchronize()
    check(lib.la_lab_set_ptr(0, None), 'debug_set_ptr')
    t = buf.cpu().numpy().reshape(nwg, nwv, 8).astype(np.float64)
    base = t[:, :, 0].min()
    names = ['entry', 'ranges+q+K issued', 'first tile done', 'loop end', 'merge end', 'exit (par 0)']
    print(f'-- timeline k_tree_attn nkeys={nkeys} nsplit={nsplit}: us since first wave, percentiles 0/10/50/90/100')
    for k, nm in enumerate(names):
        a = t[:, :, k].reshape(-1)
        a = a[a > 0]
        a = (a - base) / 100.0
        print(f'   {nm:20s} ' + ' '.join(f'{np.percentile(a, q):6.2f}' for q in (0, 10, 50, 90, 100)) + f'  (n={len(a)})', flush=True)
    st = torch.zeros(_lib.LA_ST_WORDS, dtype=torch.int32, device=DEV)
    cv = torch.zeros(64 * 8, dtype=torch.float32, device=DEV)
    ci = torch.zeros(64 * 8, dtype=torch.int32, device=DEV)
    us = timeit(lambda i: lib.la_argmax_finalize(sp(), ptr(cv), ptr(ci), 8, ptr(st)), 200)
    print(f'launch floor (argmax_finalize, 64 blocks): {us:.2f} us', flush=True)


def attn(nh=32, nkv=32):
    """tree attention (+ combine) at the 7B shape: K/V tiles straight into registers vs staged once per workgroup through LDS
    (la_debug_set key 10), by context length and key-split count; K/V rotate over 6 layers' worth of cache (> Infinity Cache)."""
    g = torch.Generator(device=DEV).manual_seed(2)
    qf = torch.randn(nh * 8192, generator=g, device=DEV).to(torch.bfloat16)
    kf = torch.randn(nkv * 8192, generator=g, device=DEV).to(torch.bfloat16)
    vf = torch.randn(nkv * 8192, generator=g, device=DEV).to(torch.bfloat16)
    max_keys = 4096 + 64
    NL = 6
    km = torch.randn(NL, nkv * max_keys * 128, generator=g, device=DEV).to(torch.bfloat16)
    vm = torch.randn(NL, nkv * max_keys * 128, generator=g, device=DEV).to(torch.bfloat16)
    rm = torch.from_numpy(np.array([(2 << t) - 1 for t in range(63)] + [-1], dtype=np.int64)).to(DEV)
    out = torch.zeros(64 * nh * 128, dtype=torch.bfloat16, device=DEV)
    for nkeys in (640, 992, 1984, 4032):
        state = torch.zeros(_lib.LA_ST_WORDS, dtype=torch.int32, device=DEV)
        state[0] = nkeys
        for nsplit in (4, 8):
            opart = torch.zeros(nh * nsplit * 64 * 128, dtype=torch.float32, device=DEV)
            mpart = torch.zeros(nh * nsplit * 64, dtype=torch.float32, device=DEV)
            lpart = torch.zeros_like(mpart)
            res = []
            for staged in (0, 1, 0, 1):
                check(lib.la_lab_set(10, staged), 'debug_set')
                res.append(timeit(lambda i: lib.la_tree_attn(sp(), ptr(qf), ptr(km[i % NL]), ptr(vm[i % NL]), ptr(kf), ptr(vf), ptr(rm),
                                                             ptr(state), nh, nkv, max_keys, nsplit, ptr(opart), ptr(mpart), ptr(lpart),
                                                             ptr(out)), 60))
            kvb = 2 * nkv * 128 * 2 * (nkeys + 64)
            d, st = min(res[0], res[2]), min(res[1], res[3])
            print(f'tree_attn(+combine) nkeys={nkeys:5d} nsplit={nsplit}: direct {d:6.2f} us ({kvb / d / 1e3:5.0f} GB/s KV)   '
                  f'staged {st:6.2f} us ({kvb / st / 1e3:5.0f} GB/s KV)', flush=True)
    check(lib.la_lab_set(10, 0), 'debug_set')


if __name__ == '__main__':
    which = sys.argv[1:] or ['gemm', 'small']
    print(f'device {torch.cuda.get_device_name(0)} CUs {NWG}', flush=True)
    if 'gemm' in which:
        gemms()
    if 'small' in which:
        small()
    if 'attn' in which:
        attn()
    if 'sweep' in which:
        sweep()
    if 'prio' in which:
        prio()
