import os; os.environ.setdefault('LA_LAB_BUILD', '1')      # A/B script: the lab build (kernel-lab knobs, phase stamps) is the process library
import sys, os, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from painlessinferenceacceleration_amd.device_trie import DeviceTrie
from painlessinferenceacceleration_amd.lookahead_cache import LookaheadCache
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
t0 = time.time()
for q in qs: cache.hier_get_packed(q, 64, 12, 0, 32, 'mix', 0)
th = (time.time() - t0) / 256
t0 = time.time(); dev = DeviceTrie(cache, idx=0); torch.cuda.synchronize(); tm = time.time() - t0
print(f'host hier_get: {th*1e6:.1f} us/query; mirror export+upload: {tm*1e3:.2f} ms for {dev.n_records} nodes')
for B in (1, 8, 64, 256):
    dev.hier_get(qs[:B], 64, 12, 0, 32, 'mix')
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # time only the kernel: re-issue through the class (includes small H2D/D2H); report wall per call as well
    t0 = time.time(); n = 20
    for _ in range(n): dev.hier_get(qs[:B], 64, 12, 0, 32, 'mix')
    w = (time.time() - t0) / n
    print(f'device hier_get B={B}: {w*1e6:.0f} us per call incl. transfers ({w*1e6/B:.1f} us/query)')

# phase stamps of k_trie_hier_get (la_lab_set_ptr(0, buffer): wall_clock64 = 100 MHz ticks)
import ctypes as C
from painlessinferenceacceleration_amd._lib import lib, check
B = 64
stamps = torch.zeros(B * 8, dtype=torch.int64, device='cuda:0')
check(lib.la_lab_set_ptr(0, C.c_void_p(stamps.data_ptr())), 'set_ptr')
dev.hier_get(qs[:B], 64, 12, 0, 32, 'mix')
check(lib.la_lab_set_ptr(0, None), 'set_ptr')
st = stamps.cpu().numpy().reshape(B, 8)
us = lambda a, b: (st[:, b] - st[:, a]) / 100.0
ok = st[:, 4] > 0
print(f'phases over {int(ok.sum())} of {B} queries that reached the DFS (us, median / max): '
      f'match {np.median(us(0,1)[ok]):.1f}/{us(0,1)[ok].max():.1f}  live-subtree scan {np.median(us(1,2)[ok]):.1f}/{us(1,2)[ok].max():.1f}  '
      f'cut-offs {np.median(us(2,3)[ok]):.1f}/{us(2,3)[ok].max():.1f}  ordered DFS {np.median(us(3,4)[ok]):.1f}/{us(3,4)[ok].max():.1f}  '
      f'| live rows median {np.median(st[ok,5]):.0f} max {st[ok,5].max()}  emitted median {np.median(st[ok,6]):.0f}')
