# -*- coding: utf-8 -*-
"""GPU diagnostic: greedy vs lookahead through the same engine must be bit-identical; find where they diverge."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from painlessinferenceacceleration_amd.llama_engine import LlamaShape, LlamaVerifyEngine, random_weights
from painlessinferenceacceleration_amd.lookahead_cache import LookaheadCache
from painlessinferenceacceleration_amd.modeling_llama import LlamaForCausalLM
from tests.gpu_utils import random_tree

L = int(os.environ.get('DIAG_LAYERS', '4'))
P = int(os.environ.get('DIAG_P', '512'))
shape = LlamaShape(L, 4096, 32, 32, 11008, 32000, 1e-5)
model = LlamaForCausalLM.random_init(shape, seed=0, max_length=1200, eos_token_id=None)
eng = model.engine
rs = np.random.RandomState(0)
prompt = rs.randint(3, 32000, size=P).tolist()
t0 = time.time()
gre = model.greedy_search(torch.tensor([prompt]), P + 40, eos_token_id=None)[0].tolist()
print('greedy 40 tokens', time.time() - t0, 's; first tokens', gre[P:P + 8])
truth = gre[P:]
for rep in range(2):
    eng.reset(); tok0 = eng.prefill(prompt)
    print('rep', rep, 'prefill tok0', tok0, 'truth0', truth[0], 'nkeys', eng.n_keys)
    toks, n = eng.step(np.asarray([tok0], dtype=np.int32), np.array([1], dtype=np.uint64))
    l1 = eng.logits()[0].clone(); st1 = eng.state().cpu().numpy().copy()
    print('  T=1 step ->', toks, 'truth1', truth[1], 'argmax row0', int(l1.float().argmax()), 'state argmax', st1[136])
    eng.reset(); eng.prefill(prompt)
    T = 64
    _, rows = random_tree(rs, T)
    ids = np.concatenate([[tok0], rs.randint(3, 32000, size=T - 1)]).astype(np.int32)
    toks2, n2 = eng.step(ids, rows)
    l2 = eng.logits()[0].clone()
    print('  T=64 random tree ->', toks2, 'row0 bitwise equal to T=1:', bool(torch.equal(l1, l2)),
          'maxdiff', float((l1.float() - l2.float()).abs().max()))
    eng.reset(); eng.prefill(prompt)
    chain = np.asarray([tok0] + truth[1:13], dtype=np.int32)
    rowsc = np.array([(2 << t) - 1 for t in range(13)], dtype=np.uint64)
    toks3, n3 = eng.step(chain, rowsc)
    am = eng.state().cpu().numpy()[136:136 + 13].tolist()
    print('  chain of truth -> accepted', len(toks3), toks3[:5], 'truth', truth[1:6], 'argmax rows', am[:6])
cache = LookaheadCache(eos_ids=[None]); model.lookahead_cache = cache
cache.put(prompt[-2:] + truth, branch_length=13, mode='output', idx=-1)
dk = {'use_lookahead': True, 'decoding_length': 64, 'branch_length': 12, 'stop_words': {}}
out = model.lookahead_generation(torch.tensor([prompt]), stopping_criteria=P + 40, eos_token_id=[None],
                                 return_dict_in_generate=True, decoding_kwargs=dk)
seq = out.sequences[0].tolist()
agree = next((i for i, (a, b) in enumerate(zip(seq, gre)) if a != b), min(len(seq), len(gre)))
print('lookahead vs greedy agree for', agree - P, 'generated tokens; dls', out.kwargs['dls'], 'edls', out.kwargs['edls'])
