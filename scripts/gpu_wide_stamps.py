import os; os.environ.setdefault('LA_LAB_BUILD', '1')      # A/B script: the lab build (kernel-lab knobs, phase stamps) is the process library
# -*- coding: utf-8 -*-
"""Where a wave of the 512-row gate/up launch (k_gemm_wide<4,4,SWIGLU>, schedule 3) spends its main loop: measurement build DBG = 6
(la_lab_set(4, 6)) sums shader cycles per wave over the stages in four segments — first half, wait for the own DMA pieces, barrier,
second half.  (The wave-priority schedules this script compared in round 4 — profiles/r04_wide_gemm_schedule.txt part 4 — were removed
from the kernel.)  Every s_memtime costs its own scalar round trip, so read the
proportions, not the absolute total; the launch time of the plain build (dbg 0) is printed next to them.

    python scripts/gpu_wide_stamps.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from painlessinferenceacceleration_amd._lib import check, lib      # noqa: E402
from tests import gpu_utils as gu                                    # noqa: E402
from tests.gpu_utils import DEV, ptr, sp                             # noqa: E402

NWG = 256
NAMES = ('first half', 'vmcnt wait', 'barrier', 'second half')


def run(name, F, K, prios):
    g = torch.Generator(device=DEV).manual_seed(0)
    wps = [gu.pack_planned(1, [(torch.randn(F, K, generator=g, device=DEV) * 0.05).to(torch.bfloat16),
                               (torch.randn(F, K, generator=g, device=DEV) * 0.05).to(torch.bfloat16)], NWG) for _ in range(3)]
    act = torch.zeros(8 * 64 * F, dtype=torch.bfloat16, device=DEV)
    nblk = 8
    x = (torch.randn(nblk * 64, K, generator=g, device=DEV)).to(torch.bfloat16)
    xp = torch.cat([gu.pack_x(x[b * 64:(b + 1) * 64].contiguous()) for b in range(nblk)])
    z = None
    buf = torch.zeros(NWG * 8 * 8, dtype=torch.int64, device=DEV)
    check(lib.la_lab_set_ptr(0, ptr(buf)), 'set_ptr')

    def gateup(i):
        check(lib.la_mb_gemm(sp(), 1, ptr(wps[i % 3]), ptr(xp), F, K, nblk, NWG, 1, ptr(z), 0, ptr(act), ptr(z), ptr(z), ptr(z),
                             ptr(z), ptr(z), ptr(z), ptr(z), ptr(z), ptr(z), 0, 0), 'mb_gemm')

    def timed(trials=5, n=16):
        for i in range(3):
            gateup(i)
        torch.cuda.synchronize()
        res = []
        for _ in range(trials):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(n):
                gateup(i)
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) * 1e3 / n)
        return min(res), sorted(res)[len(res) // 2]

    for prio in prios:
        check(lib.la_lab_set(4, 0), 'lab_set')
        us, med = timed()
        check(lib.la_lab_set(4, 6), 'lab_set')
        us6, _ = timed(trials=2, n=4)
        t = buf.cpu().numpy().reshape(NWG, 8, 8).astype(np.float64)
        seg = t[:, :, :4] / np.maximum(t[:, :, 4:5], 1)            # cycles per stage
        lo, hi = seg[:, :4, :].mean(axis=(0, 1)), seg[:, 4:, :].mean(axis=(0, 1))
        print(f'{name} gate/up 512 rows, min {us:7.2f} us  median {med:7.2f} us   (stamped build {us6:.1f} us)')
        print('     cycles per stage   ' + '  '.join(f'{n:>11s}' for n in NAMES) + '        sum')
        print('     waves 0..3         ' + '  '.join(f'{v:11.1f}' for v in lo) + f'  {lo.sum():9.1f}')
        print('     waves 4..7         ' + '  '.join(f'{v:11.1f}' for v in hi) + f'  {hi.sum():9.1f}', flush=True)
    check(lib.la_lab_set(4, 0), 'lab_set')
    check(lib.la_lab_set_ptr(0, 0), 'set_ptr')


def main():
    print('matrix pipe needs 1024 cycles per SIMD and stage (2 waves x 16 MFMAs x 32)')
    run('llama-2-7b', 11008, 4096, (0,))
    run('mistral-7b', 14336, 4096, (0,))


if __name__ == '__main__':
    main()
