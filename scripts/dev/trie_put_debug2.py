import sys, numpy as np, torch
sys.path.insert(0, '.')
from painlessinferenceacceleration_amd.device_trie import DeviceTrie
from painlessinferenceacceleration_amd.lookahead_cache import LookaheadCache
from tests.test_gpu_trie import _host_image
B, bl, eos, stop = 1, 13, [None], []
rs = np.random.RandomState(100 + B)
V = 300
cache = LookaheadCache(eos_ids=eos, stop_words=set(stop))
motifs = [rs.randint(8, V, size=rs.randint(5, 30)).tolist() for _ in range(12)]
for m in motifs[:6]:
    cache.put(m, branch_length=bl, mode='output', idx=-1)
for b in range(B):
    cache.put(rs.randint(8, V, size=40).tolist(), branch_length=bl, mode='input', idx=b)
    cache.stream_put(rs.randint(8, V, size=rs.randint(1, 4)).tolist(), branch_length=bl, final=False, mode='output', idx=b)
dt = DeviceTrie(cache, idxs=list(range(B)), put_vocab=V, cap_slack=200000)
dt.load_stream_buffers()
src = torch.zeros(B * 40, dtype=torch.int32, device='cuda')
cnt = torch.zeros(B, dtype=torch.int32, device='cuda')
def check(tag):
    torch.cuda.synchronize()
    k, tok, cs, cc, cap, fo, fi = _host_image(cache, 1)
    meta = dt.meta.cpu().numpy()
    d = {n_: getattr(dt, n_)[:k].cpu().numpy() for n_ in ('tok', 'cstart', 'ccount', 'ccap')}
    h = {'tok': tok, 'cstart': cs, 'ccount': cc, 'ccap': cap}
    bad = meta[0] != k
    for n_ in d:
        diff = np.nonzero(d[n_] != h[n_])[0]
        if len(diff):
            bad = True
            print(tag, n_, 'first diffs', diff[:10], 'dev', d[n_][diff[:10]], 'host', h[n_][diff[:10]])
    print(tag, 'records', k, 'meta', meta)
    return bad
for step in range(40):
    active = [b for b in range(B) if rs.rand() < 0.85] or [0]
    puts = []
    hs, hc = np.zeros(B * 40, dtype=np.int32), np.zeros(B, dtype=np.int32)
    for k, b in enumerate(active):
        n = int(rs.randint(1, 41)) if rs.rand() < 0.3 else int(rs.randint(1, 9))
        toks = []
        while len(toks) < n:
            toks.extend(motifs[rs.randint(len(motifs))][rs.randint(0, 4):] if rs.rand() < 0.7 else rs.randint(8, V, size=3).tolist())
        toks = toks[:n]
        if rs.rand() < 0.15:
            toks[rs.randint(n)] = -1
        hs[k * 40:k * 40 + n] = toks
        hc[k] = n
        puts.append((b, toks))
    src.copy_(torch.from_numpy(hs)); cnt.copy_(torch.from_numpy(hc))
    dt.stream_put_dev(src.data_ptr(), 40, cnt.data_ptr(), active, bl)
    print('step', step, 'puts', puts, 'replay ok', dt.replay(puts, bl))
    if check('after put %d' % step): break
    if step % 7 == 3:
        cache.put(rs.randint(8, V, size=25).tolist(), branch_length=bl, mode='input', idx=int(rs.randint(B)))
        print('sync', dt.sync())
        if check('after patch %d' % step): break
    if step % 11 == 5:
        cache.reset_input_freqs(int(rs.randint(B)))
        dt.sync()
