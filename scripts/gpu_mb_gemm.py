import os; os.environ.setdefault('LA_LAB_BUILD', '1')      # A/B script: the lab build (kernel-lab knobs, phase stamps) is the process library
# -*- coding: utf-8 -*-
"""The multi-block gate/up (+SwiGLU) and down GEMMs alone at the Llama-2-7B shape through la_mb_gemm: the one-pass wide kernel
(k_gemm_wide) vs the K-split kernel (k_gemm_mb, la_debug_set key 3) for nblk = 4 and 8; weights rotate over 3 images so that no
launch finds its weights in the Infinity Cache.

    python scripts/gpu_mb_gemm.py [time|once]      (once = a single launch per variant, for counter collection)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from painlessinferenceacceleration_amd._lib import check, lib      # noqa: E402
from tests import gpu_utils as gu                                    # noqa: E402
from tests.gpu_utils import DEV, ptr, sp                             # noqa: E402

F, K, NWG, NBUF = int(os.environ.get('MB_F', '11008')), int(os.environ.get('MB_K', '4096')), 256, 3      # MB_F=14336: Mistral / Mixtral, MB_F=13824 MB_K=5120: 13B


def bf(t):
    return t.to(torch.bfloat16)


def bench(fn, trials=7, n=16):
    """min and median over `trials` of the mean time (us) of n back-to-back launches"""
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    res = []
    for _ in range(trials):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / n)
    res.sort()
    return res[0], res[len(res) // 2]


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else 'time'
    g = torch.Generator(device=DEV).manual_seed(0)
    wps = [gu.pack_planned(1, [bf(torch.randn(F, K, generator=g, device=DEV) * 0.05), bf(torch.randn(F, K, generator=g, device=DEV) * 0.05)], NWG)
           for _ in range(NBUF)]
    wd = [gu.pack_weight(bf(torch.randn(K, F, generator=g, device=DEV) * 0.05)) for _ in range(NBUF)]
    act = torch.zeros(8 * 64 * F, dtype=torch.bfloat16, device=DEV)
    slabs = torch.zeros(4 * 512 * K, dtype=torch.float32, device=DEV)
    for nblk in (() if mode in ('parts', 'onceparts') else (4, 8)):
        x = bf(torch.randn(nblk * 64, K, generator=g, device=DEV))
        xp = torch.cat([gu.pack_x(x[b * 64:(b + 1) * 64].contiguous()) for b in range(nblk)])
        a = bf(torch.randn(nblk * 64, F, generator=g, device=DEV))
        ap = torch.cat([gu.pack_x(a[b * 64:(b + 1) * 64].contiguous()) for b in range(nblk)])
        z = None
        for narrow, wmode, dw in ((0, 0, 1), (0, 49, 1), (0, 49 + 8192, 1)):
            check(lib.la_lab_set(3, narrow), 'debug_set')
            check(lib.la_lab_set(6, wmode if wmode else 1), 'debug_set')      # 3 = paired gate/up (8 waves), 17 = fat-wave gate/up (k_gemm_fat, round 5), 49 = fat-wave slab launches too
            check(lib.la_lab_set(24, dw), 'debug_set')          # 1 = round-4 schedule (default), 0 = round-2 schedule

            def gateup(i):
                check(lib.la_mb_gemm(sp(), 1, ptr(wps[i % NBUF]), ptr(xp), F, K, nblk, NWG, 1, ptr(z), 0, ptr(act), ptr(z), ptr(z), ptr(z),
                                     ptr(z), ptr(z), ptr(z), ptr(z), ptr(z), ptr(z), 0, 0), 'mb_gemm')

            def down(i):
                check(lib.la_mb_gemm(sp(), 0, ptr(wd[i % NBUF]), ptr(ap), K, F, nblk, 0, 4, ptr(slabs), 512, ptr(z), ptr(z), ptr(z), ptr(z),
                                     ptr(z), ptr(z), ptr(z), ptr(z), ptr(z), ptr(z), 0, 0), 'mb_gemm')
            for name, fn, flops in (('gate/up', gateup, 2.0 * 2 * F * K * nblk * 64), ('down', down, 2.0 * F * K * nblk * 64)):
                if mode == 'once':
                    fn(0)
                    torch.cuda.synchronize()
                    continue
                us, med = bench(fn)
                print(f'{name:8s} rows {nblk * 64:4d} {"k_gemm_mb  " if narrow else "paired, 8 waves  " if wmode == 3 else "FAT gate/up      " if wmode == 17 else "FAT gate/up+slab " if wmode == 49 else "FAT g/u REG-STAGED" if wmode == 49 + 2048 else "FAT + XCD K map  " if wmode == 49 + 8192 else "FAT, g/u UNPAIRED" if wmode == 113 else "wide (default)   " if dw else "wide, schedule 0 "} min {us:8.2f} us  median {med:8.2f} us  {flops / us / 1e6:7.1f} TFLOP/s', flush=True)
    check(lib.la_lab_set(3, 0), 'debug_set')
    check(lib.la_lab_set(6, 1), 'debug_set')
    check(lib.la_lab_set(24, 1), 'debug_set')
    if mode in ('parts', 'onceparts'):
        # what bounds a stage of the wide kernel: the same launch without MFMAs (1), without the in-loop DMA (2), DMA + barriers only (3)
        nblk = 8
        x = bf(torch.randn(nblk * 64, K, generator=g, device=DEV))
        xp = torch.cat([gu.pack_x(x[b * 64:(b + 1) * 64].contiguous()) for b in range(nblk)])
        z = None
        for dbg in (0, 1, 2, 3, 4, 5, 0):
            check(lib.la_lab_set(4, dbg), 'debug_set')

            def gateup(i):
                check(lib.la_mb_gemm(sp(), 1, ptr(wps[i % NBUF]), ptr(xp), F, K, nblk, NWG, 1, ptr(z), 0, ptr(act), ptr(z), ptr(z), ptr(z),
                                     ptr(z), ptr(z), ptr(z), ptr(z), ptr(z), ptr(z), 0, 0), 'mb_gemm')
            if mode == 'onceparts':           # one launch per build, for counter collection (dispatch order = dbg 0, 1, 2, 3, 4, 5, 0)
                gateup(0)
                torch.cuda.synchronize()
                continue
            us, med = bench(gateup)
            print(f'gate/up 512 rows wide, dbg {dbg}: min {us:8.2f} us  median {med:8.2f} us', flush=True)
        check(lib.la_lab_set(4, 0), 'debug_set')


if __name__ == '__main__':
    main()
