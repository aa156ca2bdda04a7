# -*- coding: utf-8 -*-
"""CPU: host logic of pretrained_model.lookahead_generation / greedy_search / generate on an oracle-backed engine,
against the reference's golden run (fp32: token-exact) and against plain greedy."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from painlessinferenceacceleration_amd.lookahead_cache import LookaheadCache
from painlessinferenceacceleration_amd.pretrained_model import LookaheadPreTrainedModel
from tests.oracle_engine import OracleEngine
from tests.tiny_model import load_golden, tiny_shape, tiny_weights


class Model(LookaheadPreTrainedModel):
    def __init__(self, dtype=torch.float32, max_length=512):
        sd = {k: v.to(dtype) for k, v in tiny_weights(0).items()}
        self.engine = OracleEngine(tiny_shape(), sd, max_length=max_length)
        self.generation_config = SimpleNamespace(eos_token_id=2, pad_token_id=0, return_dict_in_generate=False)
        self.lookahead_cache = LookaheadCache()


DK = {'use_lookahead': True, 'decoding_mode': 'hier', 'decoding_length': 64, 'branch_length': 12,
      'max_query_length': 2, 'stop_words': {}}


@pytest.mark.parametrize('tag,dtype', [('fp32', torch.float32), ('bf16', torch.bfloat16)])
def test_loop_reproduces_reference_run(tag, dtype):
    g = load_golden(tag)
    m = Model(dtype)
    prompt = g['prompt'].tolist()
    for r in range(int(g['n_runs'])):
        out = m.lookahead_generation(torch.tensor([prompt]), stopping_criteria=len(prompt) + 96, eos_token_id=2,
                                     return_dict_in_generate=True, decoding_kwargs=dict(DK))
        assert out.sequences[0].tolist() == g[f'r{r}_sequences'].tolist()
        assert out.kwargs['dls'] == g[f'r{r}_dls'].tolist() and out.kwargs['edls'] == g[f'r{r}_edls'].tolist()
        assert len(out.kwargs['fts']) == len(out.kwargs['dls']) and len(out.kwargs['qts']) == len(out.kwargs['dls']) - 1
    gre = m.greedy_search(torch.tensor([prompt]), len(prompt) + 96, eos_token_id=2)[0].tolist()
    assert gre == g['greedy'].tolist()


def test_lookahead_equals_greedy_with_noisy_warmup():
    """The bench.py recipe at toy scale: warm the trie with noisy copies of the greedy continuation; the lookahead
    output must still be exactly the greedy output, and drafts must be accepted."""
    from bench import noisy_copies
    m = Model(torch.float32, max_length=400)
    rs = np.random.RandomState(3)
    prompt = rs.randint(3, 512, size=90).tolist()
    truth = m.greedy_search(torch.tensor([prompt]), 90 + 120, eos_token_id=None)[0].tolist()[90:]
    cache = LookaheadCache(eos_ids=[None])
    m.lookahead_cache = cache
    for c in noisy_copies(prompt[-2:] + truth, 8, 0.12, 512, seed=9):
        cache.put(c, branch_length=13, mode='output', idx=-1)
    out = m.lookahead_generation(torch.tensor([prompt]), stopping_criteria=90 + 120, eos_token_id=[None],
                                 return_dict_in_generate=True, decoding_kwargs=dict(DK))
    seq = out.sequences[0].tolist()
    assert seq[90:90 + 120] == truth[:len(seq) - 90][:120]
    assert np.mean(out.kwargs['edls'][1:]) > 3 and max(out.kwargs['dls']) > 32


def test_generate_front_door_and_unsupported_options():
    m = Model(torch.float32)
    prompt = torch.tensor([load_golden('fp32')['prompt'].tolist()])
    a = m.generate(input_ids=prompt, max_new_tokens=30, decoding_kwargs=dict(DK), eos_token_id=2)
    b = m.generate(input_ids=prompt, max_new_tokens=30, decoding_kwargs={'use_lookahead': False}, eos_token_id=2)
    n = min(a.shape[1], b.shape[1])
    assert a[0, :n].tolist() == b[0, :n].tolist()
    with pytest.raises(NotImplementedError):
        m.generate(input_ids=prompt, max_new_tokens=5, decoding_kwargs=dict(DK), do_sample=True)
    with pytest.raises(NotImplementedError):
        m.lookahead_generation(prompt, logits_processor=[lambda *a: None], stopping_criteria=60,
                               decoding_kwargs=dict(DK))
