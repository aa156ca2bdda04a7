# -*- coding: utf-8 -*-
"""GPU: the reference's benchmark harness (lookahead/benchmarks/benchmark.py methodology: perf_check grid, batch_chat) driven over the
HIP engine — SURVEY 8(f) N2.  The tiny decisive model keeps it fast; scripts/bench_harness.py runs the same harness at the 7B shape
and writes the README-format table committed under profiles/."""
import numpy as np
import pytest
import torch

from painlessinferenceacceleration_amd.benchmark import Benchmark
from painlessinferenceacceleration_amd.lookahead_cache import LookaheadCache
from painlessinferenceacceleration_amd.modeling_llama import LlamaForCausalLM
from tests.tiny_model import noisy_copies, tiny_decisive_weights, tiny_shape

pytestmark = pytest.mark.gpu


def _corpus(n=4, P=40, seed=31):
    rs = np.random.RandomState(seed)
    return [rs.randint(3, tiny_shape().vocab, size=P).tolist() for _ in range(n)]


def test_perf_check_grid_on_the_hip_engine_equals_greedy_with_accepted_drafts(capsys):
    """Benchmark.perf_check over (decoding_length, branch_length) in {32, 64, 128} x {8, 12, 32} — 64-row steps, 128-row trees as two
    chained blocks — on the HIP engine: in EVERY cell the lookahead answers equal the plain-greedy answers (acc 1.0000 = token match
    against save_answers) and drafts are accepted (edl > 1); the speed column is tokens / wall time of generate() incl. prefill."""
    model = LlamaForCausalLM(tiny_shape(), tiny_decisive_weights(0, torch.bfloat16), max_length=512, max_blocks=2, eos_token_id=None)
    model.lookahead_cache = LookaheadCache(eos_ids=[None])
    b = Benchmark(model=model, eos=None)
    prompts = _corpus()
    answers = b.save_answers(prompts, max_new_tokens=96)
    assert all(len(a) == 96 for a in answers)
    warm = []
    for i, (p, a) in enumerate(zip(prompts, answers)):          # bench.py's warm-up: noisy copies -> multi-branch, partially accepted drafts
        warm += noisy_copies(p[-2:] + a, 6, 0.2, tiny_shape().vocab, seed=60 + i)
    res = b.perf_check(prompts, answers=answers, warmup_ids=warm, max_new_tokens=96, sizes=(32, 64, 128), lens=(8, 12, 32))
    out = capsys.readouterr().out
    assert set(res) == {(d, l) for d in (32, 64, 128) for l in (8, 12, 32)} and all(v > 0 for v in res.values())
    lines = [ln for ln in out.splitlines() if ln.startswith('mode:hier bs:1 decoding_length:')]
    assert len(lines) == 9
    widest = 0.0
    for ln in lines:
        edl, dl = (float(x) for x in ln.split('edl:')[1].split('/')[:2])
        assert 'acc:1.0000' in ln and edl > 1.0, ln
        widest = max(widest, dl)
    assert widest > 64.0                                         # the 128-token cells really ran trees wider than one block
    print(out)


def test_batch_chat_on_the_hip_engine_off_and_on_legs_agree():
    """Benchmark.batch_chat (benchmark.py:188-241): lookahead off / on per batch on a trie that learns as it goes — identical outputs,
    at bs=1 (single-sequence engine) and at bs=2 (cursor-batch engine, left-padded prompts of unequal length)."""
    from painlessinferenceacceleration_amd.modeling_llama_batch import LlamaForCausalLM as BatchLlama
    prompts = _corpus(4, 40, seed=33)
    m1 = LlamaForCausalLM(tiny_shape(), tiny_decisive_weights(0, torch.bfloat16), max_length=512, eos_token_id=None)
    m1.lookahead_cache = LookaheadCache(eos_ids=[None])
    r1 = Benchmark(model=m1, eos=None).batch_chat(prompts * 2, max_new_tokens=64, decoding_length=64, branch_length=12, batch_size=1, verbose=False)
    assert r1['identical'] and r1['speed_on'] > 0
    ragged = [p[:40 - 3 * i] for i, p in enumerate(prompts)]
    m2 = BatchLlama(tiny_shape(), tiny_decisive_weights(0, torch.bfloat16), max_length=512, max_batch=2, eos_token_id=None, max_blocks=2)
    m2.lookahead_cache = LookaheadCache(eos_ids=[None])
    r2 = Benchmark(model=m2, eos=None).batch_chat(ragged * 2, max_new_tokens=64, decoding_length=64, branch_length=12, batch_size=2, verbose=False)
    assert r2['identical'] and r2['speed_on'] > 0
