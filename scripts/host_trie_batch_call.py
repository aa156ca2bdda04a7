# host trie: 8 per-sample Python-level queries vs ONE la_cache_bat_get_packed call (build container CPU; no GPU needed)
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from painlessinferenceacceleration_amd.lookahead_cache import LookaheadCache
rs = np.random.RandomState(0)
V = 32000
cache = LookaheadCache(eos_ids=[None])
# forest like the bench: 8 sequences x noisy copies of 300-token answers + phrase-bank prompts
phr = [rs.randint(3, V, size=rs.randint(4, 16)).tolist() for _ in range(400)]
seqs = []
for b in range(8):
    truth = sum((phr[i] for i in rs.zipf(1.3, size=40) % 400), [])[:300]
    seqs.append(truth)
    for c in range(6):
        noisy = [t if rs.rand() > 0.1 else int(rs.randint(3, V)) for t in truth]
        cache.put(noisy, branch_length=13, mode='output', idx=-1)
    prompt = sum((phr[i] for i in rs.zipf(1.3, size=60) % 400), [])[:512]
    cache.put(prompt, branch_length=13, mode='input', idx=b)
print(cache.stats())
qs = [[s[50], s[51]] for s in seqs]
def serial():
    out = []
    for b in range(8):
        ids, rm, _, _ = cache.hier_get_packed(qs[b], decoding_length=64, branch_length=12, min_input_size=0, min_output_size=32, mode='mix', idx=b)
        out.append((ids.copy(), rm.copy()))
    return out
def batched():
    return cache.bat_get_packed(qs, decoding_length=64 * 8, branch_length=12, mode='mix', indices=list(range(8)), decoding_mode='hier')
for f in (serial, batched):
    f()
    t = time.perf_counter()
    for _ in range(2000): r = f()
    dt = (time.perf_counter() - t) / 2000
    print(f.__name__, round(dt * 1e6, 1), 'us per 8 queries', [len(x[0]) for x in r][:8])
