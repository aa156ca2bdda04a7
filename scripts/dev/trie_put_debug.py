import sys, numpy as np, torch
sys.path.insert(0, '.')
from painlessinferenceacceleration_amd.device_trie import DeviceTrie
from painlessinferenceacceleration_amd.lookahead_cache import LookaheadCache
from tests.test_gpu_trie import _host_image
rs = np.random.RandomState(101)
V, bl, B = 300, 13, 1
cache = LookaheadCache(eos_ids=[None])
motifs = [rs.randint(8, V, size=rs.randint(5, 30)).tolist() for _ in range(12)]
for m in motifs[:6]:
    cache.put(m, branch_length=bl, mode='output', idx=-1)
cache.stream_put([9, 10], branch_length=bl, final=False, mode='output', idx=0)
dt = DeviceTrie(cache, idxs=[0], put_vocab=V, cap_slack=200000)
dt.load_stream_buffers()
src = torch.zeros(40, dtype=torch.int32, device='cuda'); cnt = torch.zeros(1, dtype=torch.int32, device='cuda')
prev_buf = None
for step in range(40):
    n = int(rs.randint(1, 20))
    toks = []
    while len(toks) < n:
        toks.extend(motifs[rs.randint(len(motifs))][rs.randint(0, 4):] if rs.rand() < 0.7 else rs.randint(8, V, size=3).tolist())
    toks = toks[:n]
    hs = np.zeros(40, dtype=np.int32); hs[:n] = toks
    src.copy_(torch.from_numpy(hs)); cnt.fill_(n)
    torch.cuda.synchronize()
    k0 = int(dt.meta.cpu()[0])
    dt.stream_put_dev(src.data_ptr(), 40, cnt.data_ptr(), [0], bl)
    ok = dt.replay([(0, toks)], bl)
    torch.cuda.synchronize()
    k, tok, cs, cc, cap, fo, fi = _host_image(cache, 1)
    meta = dt.meta.cpu().numpy()
    d = {n_: getattr(dt, n_)[:k].cpu().numpy() for n_ in ('tok', 'cstart', 'ccount', 'ccap', 'fo')}
    h = {'tok': tok, 'cstart': cs, 'ccount': cc, 'ccap': cap, 'fo': fo}
    bad = False
    for n_ in d:
        diff = np.nonzero(d[n_] != h[n_])[0]
        if len(diff):
            bad = True
            print('step', step, n_, 'first diffs', diff[:10], 'dev', d[n_][diff[:10]], 'host', h[n_][diff[:10]])
    print('step', step, 'n', n, 'records', k0, '->', k, 'meta', meta, 'items', dt._items[:2 * min(n, 40)].cpu().numpy().reshape(-1, 2).tolist() if bad else '')
    if bad:
        print('toks', toks)
        # live records: reachable from 0
        parent = {}
        stack = [0]
        while stack:
            u = stack.pop()
            for c in range(cs[u], cs[u] + cc[u]):
                parent[c] = u; stack.append(c)
        for r in np.nonzero(d['fo'] != h['fo'])[0][:4]:
            path = []
            u = int(r)
            while u in parent:
                path.append(int(tok[u])); u = parent[u]
            print('record', r, 'live' if int(r) in parent else 'GARBAGE', 'tok', tok[r], 'path from root', path[::-1], 'cstart', cs[r], 'ccount', cc[r])
        print('buffer before', prev_buf)
        ro = dt.root_of.cpu().numpy()
        dtok, dcs, dcc = (getattr(dt, n_)[:k].cpu().numpy() for n_ in ('tok', 'cstart', 'ccount'))
        for root, path_ in ((48, [227, 105, 174]), (227, [105, 174])):
            u = int(ro[root]); print('root', root, 'root_of', u, 'live' if u in parent else 'GARBAGE', 'host tok', tok[u])
            for t in path_:
                kids = list(range(dcs[u], dcs[u] + dcc[u]))
                nxt = [c for c in kids if dtok[c] == t]
                print('   at', u, 'live' if (u in parent or u == 0) else 'GARBAGE', 'children', kids[:8], '->', nxt)
                if not nxt: break
                u = nxt[0]
        break
    import ctypes as C
    from painlessinferenceacceleration_amd import _lib
    bb = np.zeros(128, dtype=np.int32); nn = C.c_int32()
    _lib.lib.la_cache_stream_buffer(cache._h, 0, 128, bb.ctypes.data_as(_lib.pi32), C.byref(nn))
    prev_buf = bb[:nn.value].tolist()
