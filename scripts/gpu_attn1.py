import os; os.environ.setdefault('LA_LAB_BUILD', '1')      # A/B script: the lab build (kernel-lab knobs, phase stamps) is the process library
# -*- coding: utf-8 -*-
"""Round 4: the single-launch tree attention (la_attn1.hip, la_debug_set key 17) vs key splits + combine at the Llama-2-7B shape,
by context length; variants of the new kernel (key 18: start rotation off, forced slice counts); phase stamps of one launch.

    python scripts/gpu_attn1.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from painlessinferenceacceleration_amd import _lib                      # noqa: E402
from painlessinferenceacceleration_amd._lib import lib, check           # noqa: E402
from tests.gpu_utils import DEV, ptr, sp                                # noqa: E402


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


def main(nh=32, nkv=32):
    g = torch.Generator(device=DEV).manual_seed(2)
    qf = torch.randn(nh * 8192, generator=g, device=DEV).to(torch.bfloat16)
    kf = torch.randn(nkv * 8192, generator=g, device=DEV).to(torch.bfloat16)
    vf = torch.randn(nkv * 8192, generator=g, device=DEV).to(torch.bfloat16)
    max_keys = 4096 + 64
    NL = 6                                                              # K/V rotate over 6 layers' worth of cache (> Infinity Cache with the 4k runs)
    km = torch.randn(NL, nkv * max_keys * 128, generator=g, device=DEV).to(torch.bfloat16)
    vm = torch.randn(NL, nkv * max_keys * 128, generator=g, device=DEV).to(torch.bfloat16)
    rm = torch.from_numpy(np.array([(2 << t) - 1 for t in range(63)] + [-1], dtype=np.int64)).to(DEV)
    out = torch.zeros(64 * nh * 128, dtype=torch.bfloat16, device=DEV)
    nsplit = 8
    opart = torch.zeros(nh * nsplit * 64 * 128, dtype=torch.float32, device=DEV)
    mpart = torch.zeros(nh * nsplit * 64, dtype=torch.float32, device=DEV)
    lpart = torch.zeros_like(mpart)

    def run(state):
        return lambda i: lib.la_tree_attn(sp(), ptr(qf), ptr(km[i % NL]), ptr(vm[i % NL]), ptr(kf), ptr(vf), ptr(rm), ptr(state), nh, nkv,
                                          max_keys, nsplit, ptr(opart), ptr(mpart), ptr(lpart), ptr(out))
    variants = [('split+combine', 0, 0), ('one launch (SL=2)', 1, 0), ('  no rotation', 1, 1), ('  SL=4', 1, 6), ('  SL=1', 1, 2)]
    for nkeys in (512, 640, 768, 992, 1500, 1984, 4032):
        state = torch.zeros(_lib.LA_ST_WORDS, dtype=torch.int32, device=DEV)
        state[0] = nkeys
        kvb = 2 * nkv * 128 * 2 * (nkeys + 64)
        line = []
        for name, one, var in variants:
            check(lib.la_lab_set(17, one), 'debug_set')
            check(lib.la_lab_set(18, var), 'debug_set')
            us = min(timeit(run(state), 60), timeit(run(state), 60))
            line.append(f'{name.strip()} {us:6.2f}')
        print(f'nkeys={nkeys:5d} ({kvb / 1e6:5.1f} MB K/V): ' + ' | '.join(line), flush=True)
    check(lib.la_lab_set(17, 1), 'debug_set')
    # phase stamps of the single-launch kernel (us since the first wave of the launch started; percentiles over 256 WGs x 8 waves)
    for nkeys, var in ((768, 0), (1984, 0)):
        check(lib.la_lab_set(18, var), 'debug_set')
        state = torch.zeros(_lib.LA_ST_WORDS, dtype=torch.int32, device=DEV)
        state[0] = nkeys
        buf = torch.zeros(nh * 8 * 8 * 8, dtype=torch.int64, device=DEV)
        fn = run(state)
        for i in range(4):
            fn(i)
        torch.cuda.synchronize()
        check(lib.la_lab_set_ptr(0, ptr(buf)), 'debug_set_ptr')
        buf.zero_()
        fn(5)
        torch.cuda.synchronize()
        check(lib.la_lab_set_ptr(0, None), 'debug_set_ptr')
        t = buf.cpu().numpy().reshape(nh * 8, 8, 8).astype(np.float64)
        base = t[:, :, 0].min()
        names = ['entry', 'Q parked, K0 issued', 'first tile done', 'loop end', 'exit', 'first tile: scores done']
        print(f'-- k_tree_attn1 nkeys={nkeys} variant {var}: us since the first wave, percentiles 0/10/50/90/100')
        for k in (0, 1, 5, 2, 3, 4):
            a = t[:, :, k].reshape(-1)
            a = (a[a > 0] - base) / 100.0
            print(f'   {names[k]:26s}' + ' '.join(f'{np.percentile(a, q):6.2f}' for q in (0, 10, 50, 90, 100)) + f'  (n={a.size})')
        per_wave_tiles = (nkeys // 32 + 2 + 7) // 8
        loop = (t[:, :, 3] - t[:, :, 2]) / 100.0
        print(f'   loop after the first tile: median {np.median(loop):.2f} us for ~{per_wave_tiles - 1} more tiles per wave', flush=True)
    check(lib.la_lab_set(18, 0), 'debug_set')


if __name__ == '__main__':
    torch.cuda.set_device(0)
    main()
