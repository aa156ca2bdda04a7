# host-side probe on the GPU box: trie query cost with and without a live HIP context / engine
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import phrase_prompt, noisy_copies
from painlessinferenceacceleration_amd.lookahead_cache import LookaheadCache
V = 32000
g = torch.Generator(); g.manual_seed(0)
order = 3 + torch.randperm(V - 3, generator=g)
perm = torch.arange(V); perm[order] = torch.roll(order, -1); perm = perm.tolist()
prompt = phrase_prompt(1234, 512, V)
t = prompt[-1]; truth = []
for _ in range(1000):
    t = perm[t]; truth.append(t)
def run(tag, between=None):
    cache = LookaheadCache(eos_ids=[None])
    for c in noisy_copies(prompt[-2:] + truth, 12, 0.3, V, seed=99):
        cache.put(c, branch_length=13, mode='output', idx=-1)
    seq = list(prompt); cache.put(seq[1:], branch_length=13, mode='input', idx=0); seq.append(truth[0])
    k, qt, pt = 1, [], []
    while k < 600:
        t0 = time.perf_counter()
        cache.hier_get_packed(seq[-2:], decoding_length=64, branch_length=12, min_input_size=0, min_output_size=32, mode='mix', idx=0)
        qt.append(time.perf_counter() - t0)
        if between: between()
        toks = truth[k:k + 6]; seq.extend(toks); k += 6
        t0 = time.perf_counter(); cache.stream_put(toks, branch_length=13, final=False, idx=0); pt.append(time.perf_counter() - t0)
    print(f'{tag}: query mean {1e3*np.mean(qt):.3f} ms p50 {1e3*np.median(qt):.3f} max {1e3*np.max(qt):.3f}; stream_put mean {1e3*np.mean(pt):.3f} ms', flush=True)
print('affinity cpus', len(os.sched_getaffinity(0)), 'load', os.getloadavg())
run('no gpu context')
x = torch.zeros(1 << 20, device='cuda:0'); torch.cuda.synchronize()
run('hip context alive')
st = torch.cuda.Stream()
def gpu_work():
    with torch.cuda.stream(st):
        y = x * 2
    st.synchronize()
run('sync GPU op between calls', gpu_work)
def gpu_long():
    with torch.cuda.stream(st):
        for _ in range(40): y = x * 2
    st.synchronize()
run('4ms-ish GPU work + sync between calls', gpu_long)
