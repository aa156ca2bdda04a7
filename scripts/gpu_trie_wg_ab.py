import os; os.environ.setdefault('LA_LAB_BUILD', '1')      # A/B script: the lab build (kernel-lab knobs, phase stamps) is the process library
"""round 6: one workgroup per query (la_trie_wg.hip) vs one wavefront per query (la_trie_dev.hip) on the round-3 forest; kernel time by HIP
events around the launch (transfers excluded) and the per-phase stamps of both kernels."""
import sys, os, time, random, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
from painlessinferenceacceleration_amd.device_trie import DeviceTrie
from painlessinferenceacceleration_amd.lookahead_cache import LookaheadCache
from painlessinferenceacceleration_amd._lib import lib, check
nr = np.random.RandomState(0); rng = random.Random(0)
cache = LookaheadCache(eos_ids=[None])
phrases = [nr.randint(3, 32000, size=nr.randint(3, 10)).tolist() for _ in range(2000)]
for _ in range(100):
    seq = []
    while len(seq) < 256: seq.extend(phrases[min(int(nr.zipf(1.3)) - 1, 1999)])
    cache.put(seq[:256], branch_length=13, mode='output', idx=-1)
print('forest', cache.stats())
qs = []
for _ in range(256):
    ph = phrases[min(int(nr.zipf(1.3)) - 1, 1999)]; k = rng.randrange(1, len(ph)); qs.append(ph[max(0, k - 2):k])
q1 = [q[-1:] for q in qs]          # 1-token queries: whole per-token trees (thousands of entries)
t0 = time.time()
for q in qs: cache.hier_get_packed(q, 64, 12, 0, 32, 'mix', 0)
print(f'host hier_get: {(time.time() - t0) / 256 * 1e6:.1f} us/query')

def kernel_us(dev, queries, dl, bl, mo, reps=20):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * reps)]
    dev.hier_get(queries, dl, bl, 0, mo, 'mix')
    ts = []
    for r in range(reps):
        dev.sync(); dev._staging_free()
        torch.cuda.synchronize()
        # a weight-sized sweep between launches evicts the image from L2 / MALL as a verify step would
        flush.add_(1)
        torch.cuda.synchronize()
        ev[2 * r].record()
        dev.hier_get_dev(queries, decoding_length=dl, branch_length=bl, min_input_size=0, min_output_size=mo, mode='mix', sync=False)
        ev[2 * r + 1].record()
        torch.cuda.synchronize()
        ts.append(ev[2 * r].elapsed_time(ev[2 * r + 1]) * 1e3)
    return float(np.median(ts)), float(np.min(ts))

flush = torch.zeros(1 << 30, dtype=torch.int32, device='cuda:0')     # 4 GiB
out = {}
for algo in ('wave', 'wg'):
    dev = DeviceTrie(cache, idx=0, algo=algo)
    for name, Q in (('2tok', qs), ('1tok', q1)):
        for B in (1, 8, 64):
            med, mn = kernel_us(dev, Q[:B], 64, 12, 32)
            out[f'{algo}_{name}_B{B}'] = med
            print(f'{algo:4s} {name} B={B:3d}: launch (H2D of the query block + kernel) median {med:7.1f} us  min {mn:7.1f}')
    # phase stamps
    B = 64
    for name, Q in (('2tok', qs), ('1tok', q1)):
        stamps = torch.zeros(B * 8, dtype=torch.int64, device='cuda:0')
        check(lib.la_lab_set_ptr(0, C.c_void_p(stamps.data_ptr())), 'set_ptr')
        dev.hier_get(Q[:B], 64, 12, 0, 32, 'mix')
        check(lib.la_lab_set_ptr(0, None), 'set_ptr')
        st = stamps.cpu().numpy().reshape(B, 8)
        us = lambda a, b: (st[:, b] - st[:, a]) / 100.0
        ok = st[:, 4] > 0
        if ok.sum() == 0: continue
        if algo == 'wg':
            rows, nout = st[:, 5] & 0xffffffff, st[:, 5] >> 32
            print(f'wg   {name} phases over {int(ok.sum())} of {B} queries (us, median / max): match {np.median(us(0,1)[ok]):.1f}/{us(0,1)[ok].max():.1f}  '
                  f'expand {np.median(us(1,2)[ok]):.1f}/{us(1,2)[ok].max():.1f}  cut-offs {np.median(us(2,3)[ok]):.1f}/{us(2,3)[ok].max():.1f}  '
                  f'compact {np.median(us(3,6)[ok]):.1f}/{us(3,6)[ok].max():.1f}  order+emit {np.median(us(6,4)[ok]):.1f}/{us(6,4)[ok].max():.1f}  '
                  f'total {np.median(us(0,4)[ok]):.1f}/{us(0,4)[ok].max():.1f} | live rows median {np.median(rows[ok]):.0f} max {rows[ok].max()}  emitted median {np.median(nout[ok]):.0f}'
                  f'  entries median {np.median(st[ok,7] >> 32):.0f} max {(st[ok,7] >> 32).max()}  candidates median {np.median(st[ok,7] & 0xffffffff):.0f} max {(st[ok,7] & 0xffffffff).max()}')
        else:
            print(f'wave {name} phases over {int(ok.sum())} of {B} queries (us, median / max): match {np.median(us(0,1)[ok]):.1f}/{us(0,1)[ok].max():.1f}  '
                  f'scan {np.median(us(1,2)[ok]):.1f}/{us(1,2)[ok].max():.1f}  cut-offs {np.median(us(2,3)[ok]):.1f}/{us(2,3)[ok].max():.1f}  '
                  f'ordered DFS {np.median(us(3,4)[ok]):.1f}/{us(3,4)[ok].max():.1f}  total {np.median(us(0,4)[ok]):.1f}/{us(0,4)[ok].max():.1f} | live rows median {np.median(st[ok,5]):.0f} max {st[ok,5].max()}  emitted median {np.median(st[ok,6]):.0f}')
# wide trees (workgroup kernel only)
dev = DeviceTrie(cache, idx=0, algo='wg', max_rows=256)
for dl, bl in ((128, 32), (256, 32)):
    for name, Q in (('2tok', qs), ('1tok', q1)):
        med, mn = kernel_us(dev, Q[:8], dl, bl, dl // 2)
        out[f'wg_{name}_dl{dl}_B8'] = med
        print(f'wg   {name} B=8 decoding_length={dl} branch_length={bl}: median {med:7.1f} us  min {mn:7.1f}')
print('RESULT', json.dumps(out))
