# host trie: cost of a batch step's stream_put calls (8 sequences, random tokens = every branch new) one by one vs la_cache_stream_put_many
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from painlessinferenceacceleration_amd.lookahead_cache import LookaheadCache
rs = np.random.RandomState(0)
def mk():
    c = LookaheadCache(eos_ids=[None])
    for _ in range(50):
        c.put(rs.randint(3, 32000, size=300).tolist(), branch_length=13, mode='output', idx=-1)
    return c
a, b = mk(), None
steps = [[rs.randint(3, 32000, size=rs.randint(1, 14)).tolist() for _ in range(8)] for _ in range(3000)]
c1 = mk(); c2 = mk()
t = time.perf_counter()
for st in steps:
    for i in range(8):
        c1.stream_put(st[i], branch_length=13, final=False, idx=i)
t1 = (time.perf_counter() - t) / len(steps)
t = time.perf_counter()
for st in steps:
    c2.stream_put_many([(i, st[i]) for i in range(8)], branch_length=13, final=False)
t2 = (time.perf_counter() - t) / len(steps)
print('8 stream_put calls %.1f us; one stream_put_many %.1f us' % (t1 * 1e6, t2 * 1e6), c1.stats()['n_nodes'], c2.stats()['n_nodes'])
