import os; os.environ.setdefault('LA_LAB_BUILD', '1')      # A/B script: the lab build (kernel-lab knobs, phase stamps) is the process library
# -*- coding: utf-8 -*-
"""Paired vs unpaired wide launches (la_debug_set key 6) of the slab GEMMs (o_proj, down_proj) and the QKV GEMM at the Llama-2-7B /
Mistral-7B / Llama-2-13B shapes for 256 and 512 rows; weights rotate over 3 images (no launch finds them in the Infinity Cache).

    python scripts/gpu_mb_pair.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from painlessinferenceacceleration_amd._lib import check, lib              # noqa: E402
from painlessinferenceacceleration_amd.llama_engine import rope_tables      # noqa: E402
from scripts.gpu_mb_gemm import bench, bf                                   # noqa: E402
from tests import gpu_utils as gu                                            # noqa: E402
from tests.gpu_utils import DEV, ptr, sp                                     # noqa: E402

NBUF = 3


def main():
    g = torch.Generator(device=DEV).manual_seed(0)
    rc, rs_ = rope_tables(128, 2048, 10000.0, DEV)
    z = None
    for name, hidden, ffn, nh, nkv in (('llama-2-7b', 4096, 11008, 32, 32), ('mistral-7b', 4096, 14336, 32, 8), ('llama-2-13b', 5120, 13824, 40, 40)):
        Nq = (nh + 2 * nkv) * 128
        wq = [gu.pack_planned(2, [bf(torch.randn(Nq, hidden, generator=g, device=DEV) * 0.05)], 256) for _ in range(NBUF)]
        wo = [gu.pack_weight(bf(torch.randn(hidden, nh * 128, generator=g, device=DEV) * 0.05)) for _ in range(NBUF)]
        wd = [gu.pack_weight(bf(torch.randn(hidden, ffn, generator=g, device=DEV) * 0.05)) for _ in range(NBUF)]
        wgu = [gu.pack_planned(1, [bf(torch.randn(ffn, hidden, generator=g, device=DEV) * 0.05), bf(torch.randn(ffn, hidden, generator=g, device=DEV) * 0.05)], 256)
               for _ in range(NBUF)]
        act = torch.zeros(8 * 64 * ffn, dtype=torch.bfloat16, device=DEV)
        slabs = torch.zeros(4 * 512 * hidden, dtype=torch.float32, device=DEV)
        qf = torch.zeros(8 * nh * 8192, dtype=torch.bfloat16, device=DEV)
        kf = torch.zeros(8 * nkv * 8192, dtype=torch.bfloat16, device=DEV)
        vf = torch.zeros(8 * nkv * 8192, dtype=torch.bfloat16, device=DEV)
        for nblk in (4, 8):
            x = bf(torch.randn(nblk * 64, hidden, generator=g, device=DEV))
            xp = torch.cat([gu.pack_x(x[b * 64:(b + 1) * 64].contiguous()) for b in range(nblk)])
            xo = bf(torch.randn(nblk * 64, nh * 128, generator=g, device=DEV))
            xop = torch.cat([gu.pack_x(xo[b * 64:(b + 1) * 64].contiguous()) for b in range(nblk)])
            a = bf(torch.randn(nblk * 64, ffn, generator=g, device=DEV))
            ap = torch.cat([gu.pack_x(a[b * 64:(b + 1) * 64].contiguous()) for b in range(nblk)])
            pos = torch.randint(0, 900, (nblk * 64,), generator=g, device=DEV, dtype=torch.int32)

            def qkv(i):
                check(lib.la_mb_gemm(sp(), 2, ptr(wq[i % NBUF]), ptr(xp), Nq, hidden, nblk, 256, 1, ptr(z), 0, ptr(z), ptr(z), ptr(z), ptr(z),
                                     ptr(pos), ptr(rc), ptr(rs_), ptr(qf), ptr(kf), ptr(vf), nh, nkv), 'mb_gemm')

            def oproj(i):
                check(lib.la_mb_gemm(sp(), 0, ptr(wo[i % NBUF]), ptr(xop), hidden, nh * 128, nblk, 0, 4, ptr(slabs), 512, ptr(z), ptr(z), ptr(z),
                                     ptr(z), ptr(z), ptr(z), ptr(z), ptr(z), ptr(z), ptr(z), 0, 0), 'mb_gemm')

            def down(i):
                check(lib.la_mb_gemm(sp(), 0, ptr(wd[i % NBUF]), ptr(ap), hidden, ffn, nblk, 0, 4, ptr(slabs), 512, ptr(z), ptr(z), ptr(z),
                                     ptr(z), ptr(z), ptr(z), ptr(z), ptr(z), ptr(z), ptr(z), 0, 0), 'mb_gemm')
            def gateup(i):
                check(lib.la_mb_gemm(sp(), 1, ptr(wgu[i % NBUF]), ptr(xp), ffn, hidden, nblk, 256, 1, ptr(z), 0, ptr(act), ptr(z), ptr(z), ptr(z),
                                     ptr(z), ptr(z), ptr(z), ptr(z), ptr(z), ptr(z), 0, 0), 'mb_gemm')
            if len(sys.argv) > 1 and sys.argv[1] == 'epi':
                # share of the epilogue: the paired 512-row launches with and without it (la_debug_set key 4 = 4: measurement build)
                if nblk == 8:
                    for kname, fn in (('qkv', qkv), ('o_proj', oproj), ('down', down)):
                        res = []
                        for dbg in (0, 4, 0, 4):
                            check(lib.la_lab_set(4, dbg), 'debug_set')
                            res.append(bench(fn, trials=5, n=12)[0])
                        check(lib.la_lab_set(4, 0), 'debug_set')
                        print(f'{name:12s} rows {nblk * 64:4d} {kname:7s} paired: full {min(res[0], res[2]):8.2f} us   without epilogue {min(res[1], res[3]):8.2f} us', flush=True)
                continue
            for kname, fn in (('gate/up', gateup), ('qkv', qkv), ('o_proj', oproj), ('down', down)):
                res = []
                for pair in (0, 3, 0, 3):
                    check(lib.la_lab_set(6, pair), 'debug_set')
                    res.append(bench(fn, trials=5, n=12)[0])
                print(f'{name:12s} rows {nblk * 64:4d} {kname:7s} unpaired {min(res[0], res[2]):8.2f} us   paired {min(res[1], res[3]):8.2f} us', flush=True)
        del wq, wo, wd, wgu
        torch.cuda.empty_cache()
    check(lib.la_lab_set(6, 1), 'debug_set')


if __name__ == '__main__':
    main()
