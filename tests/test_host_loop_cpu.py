# -*- coding: utf-8 -*-
"""CPU: host logic of pretrained_model.lookahead_generation / greedy_search / generate on an oracle-backed engine,
against the reference's golden run (fp32: token-exact) and against plain greedy."""
import os
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


@pytest.mark.parametrize('tag,dtype', [('fp32', torch.float32), ('bf16', torch.bfloat16), ('fp16', torch.float16)])
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
    with pytest.raises(NotImplementedError):           # attention maps / hidden states are never materialised by the device path
        m.lookahead_generation(prompt, stopping_criteria=60, output_attentions=True, decoding_kwargs=dict(DK))
    with pytest.raises(NotImplementedError):
        m.lookahead_generation(prompt, stopping_criteria=60, output_hidden_states=True, decoding_kwargs=dict(DK))


def test_output_scores_are_the_tuple_the_reference_returns():
    """`scores` under output_scores + return_dict_in_generate, against the reference's own run (oracle/gen_golden_scores.py): one entry
    per verify step, each the processed logits of the most recent NO-DRAFT step (pretrained_model.py:795, 1195, 1208: the draft branch
    never refreshes model_kwargs['next_tokens_scores']) — with an empty processor list (device accept path) and with
    RepetitionPenaltyLogitsProcessor (host-walked path).  Tokens / dls / edls as in the reference run; scores to 2e-4 (fp32 forward
    of the oracle vs the reference's eager graph).  fresh_scores (extension): draft steps carry the row of their last emitted token."""
    from transformers import LogitsProcessorList, RepetitionPenaltyLogitsProcessor
    from tests.tiny_model import GOLDEN
    g = np.load(os.path.join(GOLDEN, 'llama_tiny_scores_fp32.npz'))
    m = Model(torch.float32)
    prompt = g['bs1_prompt'].tolist()
    for r, procs in enumerate([None, LogitsProcessorList([RepetitionPenaltyLogitsProcessor(float(g['penalty']))])]):
        out = m.lookahead_generation(torch.tensor([prompt]), logits_processor=procs, stopping_criteria=len(prompt) + 64, eos_token_id=2,
                                     return_dict_in_generate=True, output_scores=True, decoding_kwargs=dict(DK))
        assert out.sequences[0].tolist() == g[f'bs1_r{r}_sequences'].tolist()
        assert out.kwargs['dls'] == g[f'bs1_r{r}_dls'].tolist() and out.kwargs['edls'] == g[f'bs1_r{r}_edls'].tolist()
        ref = g[f'bs1_r{r}_scores']
        assert len(out.scores) == ref.shape[0] == len(out.kwargs['dls'])
        got = torch.cat(out.scores, 0).float().numpy()
        assert got.shape == ref.shape
        fin = np.isfinite(ref)
        assert np.array_equal(fin, np.isfinite(got))
        assert np.abs(got[fin] - ref[fin]).max() < 2e-4
        dls = out.kwargs['dls']
        assert any(d > 1 for d in dls)
        for i, d in enumerate(dls):                      # a step with drafts repeats the previous entry
            if d > 1:
                assert np.array_equal(got[i], got[i - 1])
    # without return_dict_in_generate nothing is collected (pretrained_model.py:1092); generate() forwards the flag
    seq = m.lookahead_generation(torch.tensor([prompt]), stopping_criteria=len(prompt) + 8, eos_token_id=2, output_scores=True,
                                 decoding_kwargs=dict(DK))
    assert torch.is_tensor(seq)
    m2 = Model(torch.float32)
    out = m2.generate(input_ids=torch.tensor([prompt]), max_new_tokens=64, eos_token_id=2, return_dict_in_generate=True,
                      output_scores=True, decoding_kwargs=dict(DK))
    assert len(out.scores) == len(out.kwargs['dls']) and out.sequences[0].tolist() == g['bs1_r0_sequences'].tolist()
    # extension: per-step scores of the step's own last pick
    m3 = Model(torch.float32)
    m3.generate(input_ids=torch.tensor([prompt]), max_new_tokens=64, eos_token_id=2, decoding_kwargs=dict(DK))     # warm the trie
    dk = dict(DK); dk['fresh_scores'] = True
    out = m3.lookahead_generation(torch.tensor([prompt]), stopping_criteria=len(prompt) + 64, eos_token_id=2,
                                  return_dict_in_generate=True, output_scores=True, decoding_kwargs=dk)
    assert out.sequences[0].tolist() == g['bs1_r0_sequences'].tolist()
    seq, pos = out.sequences[0].tolist(), len(prompt)
    assert max(out.kwargs['dls']) > 1
    for sc, e in zip(out.scores, out.kwargs['edls']):
        pos += e
        assert int(torch.argmax(sc[0])) == seq[pos - 1]          # the row the step's last token was picked from


def test_batch_output_scores_are_the_tuple_the_reference_returns():
    """Batch loop: the reference appends model_kwargs['next_tokens_scores'] once per loop iteration and only its PREFILL branch ever
    writes it (pretrained_model_batch.py:789, 807, 1247, 1263) — [bs, vocab] of the last prompt rows, repeated.  Against the reference's own
    run (oracle/gen_golden_scores.py, 3 left-padded prompts); single-block and multi-block engines."""
    from painlessinferenceacceleration_amd.pretrained_model_batch import LookaheadPreTrainedModel as BatchMixin
    from tests.oracle_engine import OracleBatchEngine
    from tests.tiny_model import GOLDEN
    g = np.load(os.path.join(GOLDEN, 'llama_tiny_scores_fp32.npz'))
    ids, am = torch.from_numpy(g['b3pad_ids']), torch.from_numpy(g['b3pad_am'])
    ref = g['b3pad_scores']
    for max_blocks in (0, 4):
        class BModel(BatchMixin):
            def __init__(self):
                self.engine = OracleBatchEngine(tiny_shape(), tiny_weights(0), max_length=256, n_slots=4, max_blocks=max_blocks)
                self.generation_config = SimpleNamespace(eos_token_id=2, pad_token_id=0, return_dict_in_generate=False)
                self.lookahead_cache = LookaheadCache()
        m = BModel()
        out = m.lookahead_generation(ids, stopping_criteria=ids.shape[1] + 48, eos_token_id=2, pad_token_id=0, return_dict_in_generate=True,
                                     output_scores=True, attention_mask=am, decoding_kwargs=dict(DK))
        assert out.sequences.tolist() == g['b3pad_sequences'].tolist()
        assert out.kwargs['dls'] == g['b3pad_dls'].tolist() and out.kwargs['edls'] == g['b3pad_edls'].tolist()
        assert len(out.scores) == ref.shape[0]
        got = torch.stack(out.scores, 0).float().numpy()
        assert got.shape == ref.shape and np.abs(got - ref).max() < 2e-4
        assert all(np.array_equal(got[i], got[0]) for i in range(len(got)))


# ------------------------------------------------------------------------------------------------ batch twin
def test_batch_loop_reproduces_reference_batch_run():
    """pretrained_model_batch.lookahead_generation (product host loop + native trie) on an oracle-backed slot engine
    against the reference batch run (fp32: sequences, dls, edls exact; cases whose budget fits a 64-row block)."""
    import os
    from painlessinferenceacceleration_amd.pretrained_model_batch import LookaheadPreTrainedModel as BatchMixin
    from tests.oracle_engine import OracleBatchEngine
    from tests.tiny_model import GOLDEN

    class BModel(BatchMixin):
        def __init__(self):
            self.engine = OracleBatchEngine(tiny_shape(), tiny_weights(0), max_length=256, n_slots=4)
            self.generation_config = SimpleNamespace(eos_token_id=2, pad_token_id=0, return_dict_in_generate=False)
            self.lookahead_cache = LookaheadCache()

    g = np.load(os.path.join(GOLDEN, 'llama_tiny_batch_fp32.npz'))
    for name in ('b2', 'b3pad', 'b4'):
        bs, dl, max_new = [int(x) for x in g[f'{name}_cfg']]
        ids, am = torch.from_numpy(g[f'{name}_ids']), torch.from_numpy(g[f'{name}_am'])
        m = BModel()
        for r in range(2):
            dk = dict(DK); dk['decoding_length'] = dl
            out = m.lookahead_generation(ids, stopping_criteria=ids.shape[1] + max_new, eos_token_id=2, pad_token_id=0,
                                         return_dict_in_generate=True, attention_mask=am, decoding_kwargs=dk)
            assert out.sequences.tolist() == g[f'{name}_r{r}_sequences'].tolist(), (name, r)
            assert out.kwargs['dls'] == g[f'{name}_r{r}_dls'].tolist(), (name, r)
            assert out.kwargs['edls'] == g[f'{name}_r{r}_edls'].tolist(), (name, r)
    # budgets that can exceed a 64-row block once samples retire (decoding_length 128 / 256): the draft budget is clamped
    # (with a warning), tokens still equal the reference's; draft sizes may differ
    import warnings
    for name in ('b3pad128', 'b4w256'):
        bs, dl, max_new = [int(x) for x in g[f'{name}_cfg']]
        m = BModel()
        dk = dict(DK); dk['decoding_length'] = dl
        ids, am = torch.from_numpy(g[f'{name}_ids']), torch.from_numpy(g[f'{name}_am'])
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            out = m.lookahead_generation(ids, stopping_criteria=ids.shape[1] + max_new, eos_token_id=2, pad_token_id=0,
                                         return_dict_in_generate=True, attention_mask=am, decoding_kwargs=dk)
        ref, got = g[f'{name}_r0_sequences'], out.sequences.numpy()
        P = ids.shape[1]
        for b in range(bs):
            n = min(int((ref[b, P:] != 0).sum()), int((got[b, P:] != 0).sum()))
            assert n >= max_new - 13 and got[b, P:P + n].tolist() == ref[b, P:P + n].tolist(), (name, b)
    gre = m.greedy_search(ids, ids.shape[1] + 20, attention_mask=am, eos_token_id=2)
    assert gre[:, :ids.shape[1] + 20].tolist() == [r[:ids.shape[1] + 20] for r in g['b4w256_r0_sequences'].tolist()]

    # round 4: with a multi-block engine the budget is NOT clamped (a sample's tree may span up to 4 blocks of the pass,
    # mstep_trees), and the same two cases reproduce the reference run exactly — sequences, dls, edls, request after request
    class BModel4(BatchMixin):
        def __init__(self):
            self.engine = OracleBatchEngine(tiny_shape(), tiny_weights(0), max_length=256, n_slots=4, max_blocks=4)
            self.generation_config = SimpleNamespace(eos_token_id=2, pad_token_id=0, return_dict_in_generate=False)
            self.lookahead_cache = LookaheadCache()

    for name in ('b3pad128', 'b4w256'):
        bs, dl, max_new = [int(x) for x in g[f'{name}_cfg']]
        ids, am = torch.from_numpy(g[f'{name}_ids']), torch.from_numpy(g[f'{name}_am'])
        m = BModel4()
        for r in range(2):
            dk = dict(DK); dk['decoding_length'] = dl
            with warnings.catch_warnings():
                warnings.simplefilter('error')                 # no clamp warning on this path
                out = m.lookahead_generation(ids, stopping_criteria=ids.shape[1] + max_new, eos_token_id=2, pad_token_id=0,
                                             return_dict_in_generate=True, attention_mask=am, decoding_kwargs=dk)
            assert out.sequences.tolist() == g[f'{name}_r{r}_sequences'].tolist(), (name, r)
            assert out.kwargs['dls'] == g[f'{name}_r{r}_dls'].tolist(), (name, r)
            assert out.kwargs['edls'] == g[f'{name}_r{r}_edls'].tolist(), (name, r)


def test_batch_loop_sequential_processor_path_reproduces_reference():
    """Batch loop with RepetitionPenaltyLogitsProcessor(1.3) — the sequential accept path (forward-only step, host walk with
    the processors over the padded row + accepted tokens, host-decided commit) — against the reference batch run with the same
    processor (pretrained_model_batch.py:814-931): sequences / dls / edls exact, on the shared-block engine surface (bstep /
    bcommit) and, for the 256-token budget, one block per sample (mstep / mcommit)."""
    import os
    from transformers import LogitsProcessorList, RepetitionPenaltyLogitsProcessor
    from painlessinferenceacceleration_amd.pretrained_model_batch import LookaheadPreTrainedModel as BatchMixin
    from tests.oracle_engine import OracleBatchEngine
    from tests.tiny_model import GOLDEN

    class BModel(BatchMixin):
        def __init__(self, max_blocks):
            self.engine = OracleBatchEngine(tiny_shape(), tiny_weights(0), max_length=256, n_slots=4, max_blocks=max_blocks)
            self.generation_config = SimpleNamespace(eos_token_id=2, pad_token_id=0, return_dict_in_generate=False)
            self.lookahead_cache = LookaheadCache()

    g = np.load(os.path.join(GOLDEN, 'llama_tiny_batch_fp32_rep.npz'))
    procs = LogitsProcessorList([RepetitionPenaltyLogitsProcessor(float(g['penalty']))])
    for name, max_blocks in (('b2', 0), ('b3pad', 0), ('b3pad', 4), ('b3pad256', 4)):
        bs, dl, max_new = [int(x) for x in g[f'{name}_cfg']]
        ids, am = torch.from_numpy(g[f'{name}_ids']), torch.from_numpy(g[f'{name}_am'])
        m = BModel(max_blocks)
        for r in range(2):
            dk = dict(DK); dk['decoding_length'] = dl
            out = m.lookahead_generation(ids, logits_processor=procs, stopping_criteria=ids.shape[1] + max_new, eos_token_id=2,
                                         pad_token_id=0, return_dict_in_generate=True, attention_mask=am, decoding_kwargs=dk)
            assert out.sequences.tolist() == g[f'{name}_r{r}_sequences'].tolist(), (name, r)
            assert out.kwargs['dls'] == g[f'{name}_r{r}_dls'].tolist(), (name, r)
            assert out.kwargs['edls'] == g[f'{name}_r{r}_edls'].tolist(), (name, r)
    # the generate() front door: repetition_penalty= builds the same processor; without lookahead the host-pick greedy loop
    # emits the same tokens (lookahead is lossless under processors too)
    m = BModel(0)
    ids, am = torch.from_numpy(g['b3pad_ids']), torch.from_numpy(g['b3pad_am'])
    a = m.generate(input_ids=ids, attention_mask=am, max_new_tokens=20, decoding_kwargs=dict(DK), eos_token_id=2, pad_token_id=0,
                   repetition_penalty=float(g['penalty']))
    b = m.generate(input_ids=ids, attention_mask=am, max_new_tokens=20, decoding_kwargs={'use_lookahead': False}, eos_token_id=2,
                   pad_token_id=0, repetition_penalty=float(g['penalty']))
    n = ids.shape[1] + 20
    assert a[:, :n].tolist() == b[:, :n].tolist() == [row[:n] for row in g['b3pad_r0_sequences'].tolist()]


def test_batch_sequential_path_with_identity_processor_equals_default_path():
    """A processor that returns the scores unchanged sends the batch loop down the sequential accept path (forward-only step,
    host walk, host commit); tokens, dls and edls must equal the default path's (device-style accept) on the same prompts —
    including eos stops in the middle of an accepted chain and the max_length limit — on both engine surfaces."""
    import os
    from painlessinferenceacceleration_amd.pretrained_model_batch import LookaheadPreTrainedModel as BatchMixin
    from tests.oracle_engine import OracleBatchEngine
    from tests.tiny_model import GOLDEN

    class Identity(object):
        calls = 0

        def __call__(self, input_ids, scores):
            Identity.calls += 1
            assert input_ids.dim() == 2 and scores.dim() == 2 and input_ids.shape[0] == scores.shape[0] == 1
            return scores

    class BModel(BatchMixin):
        def __init__(self, max_blocks):
            self.engine = OracleBatchEngine(tiny_shape(), tiny_weights(0), max_length=256, n_slots=4, max_blocks=max_blocks)
            self.generation_config = SimpleNamespace(eos_token_id=2, pad_token_id=0, return_dict_in_generate=False)
            self.lookahead_cache = LookaheadCache()

    g = np.load(os.path.join(GOLDEN, 'llama_tiny_batch_fp32.npz'))
    ids, am = torch.from_numpy(g['b3pad_ids']), torch.from_numpy(g['b3pad_am'])
    ref_seq = g['b3pad_r0_sequences']
    P = ids.shape[1]
    eos = int(ref_seq[0, P + 9])                 # a token sample 0 generates early: it stops there, the others go on
    for max_blocks, dl in ((0, 64), (4, 256)):
        outs = []
        for procs in (None, [Identity()]):
            m = BModel(max_blocks)
            runs = []
            for r in range(2):
                dk = dict(DK); dk['decoding_length'] = dl
                out = m.lookahead_generation(ids, logits_processor=procs, stopping_criteria=P + 41, eos_token_id=eos, pad_token_id=0,
                                             return_dict_in_generate=True, attention_mask=am, decoding_kwargs=dk)
                runs.append((out.sequences.tolist(), out.kwargs['dls'], out.kwargs['edls']))
            outs.append(runs)
        assert outs[0] == outs[1], (max_blocks, dl)
        seq0 = outs[0][0][0][0]
        assert eos in seq0[P:P + 12] and len([t for t in seq0[P:] if t != 0]) <= 12 + 13
    assert Identity.calls > 50


@pytest.mark.parametrize('sequential', [False, True])
def test_batch_loop_with_wide_per_sample_trees_equals_greedy(sequential):
    """Per-sample budgets wider than a 64-row block (bat_get's rule gives a sample (decoding_length // bs) // bs rows of any size,
    lookahead_cache.py:534-541): the drafts come from the host trie's hier walk with multi-word row masks, the step packs every
    sample's tree as ceil(T / 64) blocks of one pass (mstep_trees) and groups the samples by the engine's block budget.  On the
    oracle engine: the output equals plain greedy decoding, trees of more than 64 rows are drafted, and the reference's own
    budget rule (no per_sample_budget) reaches the same path at decoding_length = 65 bs^2."""
    from painlessinferenceacceleration_amd.pretrained_model_batch import LookaheadPreTrainedModel as BatchMixin
    from tests.oracle_engine import OracleBatchEngine
    from tests.tiny_model import noisy_copies, tiny_decisive_weights

    class BModel(BatchMixin):
        def __init__(self, max_blocks):
            self.engine = OracleBatchEngine(tiny_shape(), tiny_decisive_weights(0, torch.float32), max_length=256, n_slots=2, max_blocks=max_blocks)
            self.generation_config = type('G', (), {'pad_token_id': 0, 'eos_token_id': None, 'return_dict_in_generate': True})()
            self.lookahead_cache = LookaheadCache(eos_ids=[])

    shape = tiny_shape()
    rs = np.random.RandomState(5)
    B, P, n_new = 2, 12, 70
    ids = rs.randint(3, shape.vocab, size=(B, P)).astype(np.int64)
    m = BModel(4)
    truth = m.greedy_search(torch.from_numpy(ids), P + n_new, eos_token_id=None)[:, P:].tolist()
    procs = None
    if sequential:
        from transformers import LogitsProcessorList, MinLengthLogitsProcessor
        procs = LogitsProcessorList([MinLengthLogitsProcessor(1, eos_token_id=1)])
    for dk in ({'use_lookahead': True, 'decoding_length': 100, 'branch_length': 30, 'stop_words': {}, 'per_sample_budget': True},
               {'use_lookahead': True, 'decoding_length': 100 * B * B, 'branch_length': 30, 'stop_words': {}}):
        m = BModel(4)
        for b in range(B):              # bench.py's warm-up: noisy copies of the continuation -> many branches below every prefix
            for c in noisy_copies(ids[b, -2:].tolist() + truth[b], 10, 0.3, shape.vocab, seed=70 + b):
                m.lookahead_cache.put(c, branch_length=31, mode='output', idx=-1)
        widths = []
        for rep in range(2):
            out = m.lookahead_generation(torch.from_numpy(ids), stopping_criteria=P + n_new, eos_token_id=[None], pad_token_id=0,
                                         return_dict_in_generate=True, decoding_kwargs=dict(dk), logits_processor=procs)
            got = out.sequences.cpu().numpy()
            for b in range(B):
                assert got[b, P:P + n_new].tolist() == truth[b][:n_new], (rep, b)
            widths.append(max(out.kwargs['dls']))
        assert max(widths) > 64, widths


def test_benchmark_harness_perf_check_and_trie_loop(capsys):
    """painlessinferenceacceleration_amd.benchmark.Benchmark (methodology of lookahead/benchmarks/benchmark.py): warm_up +
    perf_check over a (decoding_length, branch_length) grid on the oracle-backed model, and the trie-only timing loop."""
    from painlessinferenceacceleration_amd.benchmark import Benchmark
    m = Model(torch.float32)
    g = load_golden('fp32')
    prompt = g['prompt'].tolist()
    b = Benchmark(model=m, eos=2)
    answers = b.save_answers([prompt], max_new_tokens=24)
    assert answers[0] == g['greedy'].tolist()[len(prompt):len(prompt) + 24][:len(answers[0])]
    res = b.perf_check([prompt, prompt], answers=[answers[0], answers[0]], warmup_ids=answers, max_new_tokens=24,
                       sizes=(1, 16), lens=(0, 4))
    out = capsys.readouterr().out
    assert set(res) == {(1, 0), (16, 0), (16, 4)} and all(v > 0 for v in res.values())      # (1, 4) skipped: dl < bl * bs
    line = [ln for ln in out.splitlines() if ln.startswith('mode:hier bs:1 decoding_length:16 branch_length:4')][0]
    assert 'edl:' in line and 'speed:' in line and 'acc:1.0000' in line      # lookahead answers == greedy answers
    edl = float(line.split('edl:')[1].split('/')[0])
    assert edl > 1.5                                                        # warmed trie: multi-token accepts
    r = Benchmark.perf_check_trie(LookaheadCache(), [answers[0]] * 3, [prompt], [answers[0]], decoding_length=16,
                                  branch_length=4, edl=4, verbose=False)
    assert r['gets'] == 6 and r['put_tokens'] == len(prompt) + len(answers[0])


def test_loop_with_logits_processors_reproduces_reference_run():
    """Sequential accept path of the product host loop (forward-only step, host walk, commit) on the oracle-backed engine
    against the reference run with RepetitionPenaltyLogitsProcessor(1.3); generate(repetition_penalty=...) builds the
    same list; greedy_search applies it too."""
    import os
    from transformers import LogitsProcessorList, RepetitionPenaltyLogitsProcessor
    from tests.tiny_model import GOLDEN
    g = np.load(os.path.join(GOLDEN, 'llama_tiny_fp32_rep.npz'))
    m = Model(torch.float32)
    prompt = g['prompt'].tolist()
    procs = LogitsProcessorList([RepetitionPenaltyLogitsProcessor(float(g['penalty']))])
    for r in range(2):
        out = m.lookahead_generation(torch.tensor([prompt]), logits_processor=procs, stopping_criteria=len(prompt) + 64,
                                     eos_token_id=2, return_dict_in_generate=True, decoding_kwargs=dict(DK))
        assert out.sequences[0].tolist() == g[f'r{r}_sequences'].tolist()
        assert out.kwargs['dls'] == g[f'r{r}_dls'].tolist() and out.kwargs['edls'] == g[f'r{r}_edls'].tolist()
    m2 = Model(torch.float32)
    out = m2.generate(input_ids=torch.tensor([prompt]), max_new_tokens=64, repetition_penalty=float(g['penalty']),
                      eos_token_id=2, return_dict_in_generate=True, decoding_kwargs=dict(DK))
    assert out.sequences[0].tolist() == g['r0_sequences'].tolist()
    plain = m2.generate(input_ids=torch.tensor([prompt]), max_new_tokens=64, repetition_penalty=float(g['penalty']),
                        eos_token_id=2, decoding_kwargs={'use_lookahead': False})
    assert plain[0].tolist() == g['r0_sequences'].tolist()          # lookahead with processors == plain decoding with them
    s = m2.generate(input_ids=torch.tensor([prompt]), max_new_tokens=8, do_sample=True, eos_token_id=2,
                    decoding_kwargs=dict(DK))
    assert s.shape[1] >= len(prompt) + 1


def test_generate_called_like_the_reference_example():
    """examples/llama_example.py:39-69: the [False, False, True, True] loop with a stop-word SET, position_ids / use_cache
    keywords, and the LookaheadGenerationConfig route."""
    from painlessinferenceacceleration_amd.lookahead_generation_utils import LookaheadGenerationConfig
    m = Model(torch.float32)
    g = load_golden('fp32')
    input_ids = torch.tensor([g['prompt'].tolist()])
    outs = []
    for use_lookahead in [False, False, True, True]:
        decoding_kwargs = {"use_lookahead": use_lookahead, "debug_lookahead": False, "decoding_length": 64,
                           "branch_length": 12, "stop_words": set([5, 9, 11])}
        outputs = m.generate(input_ids=input_ids, attention_mask=torch.ones_like(input_ids), position_ids=None,
                             pad_token_id=2, eos_token_id=2, use_cache=True, max_new_tokens=40, repetition_penalty=1.0,
                             do_sample=False, decoding_kwargs=decoding_kwargs)
        outs.append(outputs[0, input_ids.size(-1):].tolist())
    assert outs[0] == outs[1] == outs[2][:len(outs[0])] and outs[3][:len(outs[0])] == outs[0]
    cfg = LookaheadGenerationConfig(use_lookahead=True, decoding_length=64, branch_length=12, max_new_tokens=40,
                                    eos_token_id=2, pad_token_id=2)
    out = m.generate(input_ids=input_ids, generation_config=cfg)
    assert out[0, input_ids.size(-1):].tolist()[:len(outs[0])] == outs[0]


def test_generate_forwards_what_the_reference_generate_builds():
    """generate() front door == the reference's (common/pretrained_model.py:349-372, 403, 428-441): config-derived processors +
    the caller's logits_processor list, MaxLengthCriteria + the caller's stopping_criteria, both handed to the lookahead loop;
    warpers (temperature / top_k / top_p) only in the sampling mode.  Called the way examples/llama_example.py:51-60 and
    benchmarks/benchmark.py:282-300 call it, plus the two list arguments."""
    from transformers import (LogitsProcessorList, MaxLengthCriteria, NoRepeatNGramLogitsProcessor, RepetitionPenaltyLogitsProcessor,
                              StoppingCriteria, StoppingCriteriaList)
    from painlessinferenceacceleration_amd.lookahead_generation_utils import resolve_generate_args
    g = load_golden('fp32')
    prompt = g['prompt'].tolist()
    input_ids = torch.tensor([prompt])
    ref_seq = g['r0_sequences'].tolist()
    stop_tok = ref_seq[len(prompt) + 17]                     # a token the plain run emits: generation must end right there

    class StopOnToken(StoppingCriteria):
        def __init__(self):
            self.calls = 0

        def __call__(self, input_ids, scores, **kw):
            self.calls += 1
            return bool((input_ids[0, len(prompt):] == stop_tok).any())

    # (1) user stopping_criteria reaches the lookahead loop through generate(), merged behind the length criterion
    m = Model(torch.float32)
    crit = StopOnToken()
    out = m.generate(input_ids=input_ids, attention_mask=torch.ones_like(input_ids), max_new_tokens=96, eos_token_id=2, pad_token_id=0,
                     stopping_criteria=StoppingCriteriaList([crit]), decoding_kwargs=dict(DK), return_dict_in_generate=True)
    seq = out.sequences[0].tolist()
    first = ref_seq.index(stop_tok, len(prompt))
    assert crit.calls >= 1 and stop_tok in seq[len(prompt):] and seq[:first + 1] == ref_seq[:first + 1]
    assert len(seq) <= first + 1 + 13                        # stopped at the verify step that emitted it (<= branch_length + 1 tokens later)
    direct = Model(torch.float32).lookahead_generation(input_ids, stopping_criteria=StoppingCriteriaList(
        [MaxLengthCriteria(max_length=len(prompt) + 96), StopOnToken()]), eos_token_id=2, pad_token_id=0,
        return_dict_in_generate=True, decoding_kwargs=dict(DK))
    assert direct.sequences[0].tolist() == seq and direct.kwargs['dls'] == out.kwargs['dls']
    # ... and the plain mode honours it too
    plain = Model(torch.float32).generate(input_ids=input_ids, max_new_tokens=96, eos_token_id=2, stopping_criteria=StoppingCriteriaList([StopOnToken()]),
                                          decoding_kwargs={'use_lookahead': False})
    assert plain[0].tolist() == ref_seq[:first + 1]
    # (2) repetition_penalty + no_repeat_ngram_size (config-derived) + a user processor list: the same processors, in transformers'
    # order, as a direct lookahead_generation call; a duplicate type is refused as transformers refuses it
    class Bias(torch.nn.Module):
        def forward(self, ids, scores):
            scores = scores.clone(); scores[:, 7] += 0.25
            return scores

    bias = Bias()
    a = Model(torch.float32).generate(input_ids=input_ids, max_new_tokens=48, eos_token_id=2, repetition_penalty=1.3, no_repeat_ngram_size=3,
                                      logits_processor=LogitsProcessorList([bias]), decoding_kwargs=dict(DK))
    procs = LogitsProcessorList([RepetitionPenaltyLogitsProcessor(1.3), NoRepeatNGramLogitsProcessor(3), bias])
    b = Model(torch.float32).lookahead_generation(input_ids, logits_processor=procs, stopping_criteria=len(prompt) + 48, eos_token_id=2,
                                                  decoding_kwargs=dict(DK))
    assert a[0].tolist() == b[0].tolist()
    with pytest.raises(ValueError):
        Model(torch.float32).generate(input_ids=input_ids, max_new_tokens=8, repetition_penalty=1.3,
                                      logits_processor=LogitsProcessorList([RepetitionPenaltyLogitsProcessor(1.1)]), decoding_kwargs=dict(DK))
    # (3) resolver: precedence keyword > generation_config > model defaults; warpers only for do_sample; pad defaults to eos
    cfg = SimpleNamespace(max_new_tokens=5, temperature=0.7, top_k=50, top_p=0.9, do_sample=True, eos_token_id=[2, 9])
    ga, rest = resolve_generate_args(SimpleNamespace(eos_token_id=2, pad_token_id=None, repetition_penalty=1.1), 10, generation_config=cfg,
                                     max_new_tokens=7, attention_mask='am', use_cache=True)
    assert ga.max_length == 17 and ga.eos_token_id == [2, 9] and ga.pad_token_id == 2 and ga.do_sample
    assert [type(w).__name__ for w in ga.logits_warper] == ['TemperatureLogitsWarper', 'TopKLogitsWarper', 'TopPLogitsWarper']
    assert [type(w).__name__ for w in ga.logits_processor] == ['RepetitionPenaltyLogitsProcessor']
    assert [type(c).__name__ for c in ga.stopping_criteria] == ['MaxLengthCriteria'] and rest == {'attention_mask': 'am', 'use_cache': True}
    ga, _ = resolve_generate_args(None, 10, temperature=0.5, top_k=3)
    assert len(ga.logits_warper) == 0 and ga.max_length == 30
    # (4) sampling mode with top_k = 1 is greedy: the warpers act on the plain path
    s = Model(torch.float32).generate(input_ids=input_ids, max_new_tokens=24, eos_token_id=2, do_sample=True, top_k=1, temperature=0.7,
                                      decoding_kwargs={'use_lookahead': False})
    assert s[0].tolist() == ref_seq[:len(prompt) + 24] or s[0].tolist() == g['greedy'].tolist()[:len(prompt) + 24]


class DecisiveModel(LookaheadPreTrainedModel):
    def __init__(self, dtype=torch.float32, max_length=512):
        from tests.tiny_model import tiny_decisive_weights
        self.engine = OracleEngine(tiny_shape(), tiny_decisive_weights(0, dtype), max_length=max_length)
        self.generation_config = SimpleNamespace(eos_token_id=2, pad_token_id=0, return_dict_in_generate=False)
        self.lookahead_cache = LookaheadCache(eos_ids=[2])


@pytest.mark.parametrize('suffix', ['', '_par', '_one', '_dl128'])
def test_product_loop_draft_modes_reproduce_reference_runs(suffix):
    """decoding_mode 'hier' / 'par' / 'one' (selected at pretrained_model.py:712-723, retrieval lookahead_cache.py:408-517)
    through the PRODUCT host loop and the native trie against the reference's recorded runs with a noisy warm trie
    (oracle/gen_golden_noisy.py): tokens, dls, edls.  'par' needs the surviving-branches accept walk (a shared prefix is
    duplicated across chains): the reference accepts 5 tokens in the first tree step where a child-of-current walk stops at 4."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', f'llama_tiny_noisy{suffix}_fp32.npz'))
    m = DecisiveModel(torch.float32)
    dl, bl = int(g['decoding_length']), int(g['branch_length'])       # _dl128: trees of up to 128 rows, up to 33 tokens per step
    for c in g['copies'].tolist():
        m.lookahead_cache.put(c, branch_length=bl + 1, mode='output', idx=-1)
    prompt = g['prompt'].tolist()
    for r in range(int(g['n_runs'])):
        dk = dict(DK, decoding_mode=str(g['decoding_mode']), decoding_length=dl, branch_length=bl)
        out = m.lookahead_generation(torch.tensor([prompt]), stopping_criteria=len(prompt) + int(g['max_new']), eos_token_id=2,
                                     pad_token_id=0, return_dict_in_generate=True, decoding_kwargs=dk)
        assert out.sequences[0].tolist() == g[f'r{r}_sequences'].tolist(), (suffix, r)
        assert out.kwargs['dls'] == g[f'r{r}_dls'].tolist() and out.kwargs['edls'] == g[f'r{r}_edls'].tolist(), (suffix, r)


def test_custom_stopping_criteria_are_evaluated_every_step():
    """pretrained_model.py:1225-1226: `stopping_criteria(input_ids, scores)` after every verify step — a user criterion ends the
    request at the end of the step in which it first holds (the whole accepted chunk is kept), the max-length criterion of
    the same list still bounds it, and the request flushes the trie as a normal end does."""
    from transformers import MaxLengthCriteria, StoppingCriteria, StoppingCriteriaList
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'llama_tiny_noisy_fp32.npz'))
    prompt = g['prompt'].tolist()
    P, max_new = len(prompt), int(g['max_new'])

    def fresh():
        m = DecisiveModel(torch.float32)
        for c in g['copies'].tolist():
            m.lookahead_cache.put(c, branch_length=13, mode='output', idx=-1)
        return m
    base = fresh().lookahead_generation(torch.tensor([prompt]), stopping_criteria=P + max_new, eos_token_id=2, pad_token_id=0,
                                        return_dict_in_generate=True, decoding_kwargs=dict(DK))
    seq, edls = base.sequences[0].tolist(), base.kwargs['edls']
    target = seq[P + 30]
    first = next(i for i in range(P, len(seq)) if seq[i] == target)
    ends = np.cumsum(edls) + P                              # sequence length after every step
    want_len = int(next(e for e in ends if e > first))

    calls = []

    class StopOnToken(StoppingCriteria):
        def __call__(self, input_ids, scores, **kw):
            calls.append(int(input_ids.shape[1]))
            return bool((input_ids[0, P:] == target).any())
    sc = StoppingCriteriaList([MaxLengthCriteria(max_length=P + max_new), StopOnToken()])
    m = fresh()
    out = m.lookahead_generation(torch.tensor([prompt]), stopping_criteria=sc, eos_token_id=2, pad_token_id=0,
                                 return_dict_in_generate=True, decoding_kwargs=dict(DK))
    got = out.sequences[0].tolist()
    assert got == seq[:want_len] and len(got) < len(seq)
    assert calls == [int(e) for e in ends if e <= want_len]          # once per step, on the sequence so far
    assert m.lookahead_cache.stats()['n_dirty_input_trees'] == 0     # final flush ran (reset_input_freqs)
    # a list with only the length criterion behaves as the plain integer
    out2 = fresh().lookahead_generation(torch.tensor([prompt]), eos_token_id=2, pad_token_id=0, return_dict_in_generate=True,
                                        stopping_criteria=StoppingCriteriaList([MaxLengthCriteria(max_length=P + max_new)]),
                                        decoding_kwargs=dict(DK))
    assert out2.sequences[0].tolist() == seq


def test_load_hf_checkpoint_skips_consolidated_safetensors(tmp_path):
    """Mistral / Mixtral repositories ship consolidated.safetensors (other key names) NEXT to the HF-named shards: without an
    index file only the HF files may be read — reading both doubles host memory and leaves keys nobody consumes."""
    import transformers
    from safetensors.torch import save_file
    from painlessinferenceacceleration_amd.llama_engine import load_hf_checkpoint
    cfg = transformers.LlamaConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2, vocab_size=100)
    cfg.save_pretrained(tmp_path)
    save_file({'model.embed_tokens.weight': torch.zeros(100, 64)}, str(tmp_path / 'model.safetensors'))
    save_file({'tok_embeddings.weight': torch.ones(100, 64)}, str(tmp_path / 'consolidated.safetensors'))
    save_file({'layers.0.w': torch.ones(4)}, str(tmp_path / 'consolidated.00.safetensors'))
    _, sd = load_hf_checkpoint(str(tmp_path))
    assert sorted(sd) == ['model.embed_tokens.weight']


def test_batch_loop_overlapped_trie_update_is_lossless_and_flushes():
    """decoding_kwargs['overlap_trie_update'] (extension): the trie update of step k runs after the verify pass of step k + 1 has
    been queued (LlamaVerifyEngine.mstep_async / mstep_finish) — drafts see a step's tokens one step later.  Tokens stay exactly the
    greedy ones, the trie ends up with the same n-grams as the in-order run (every deferred put is applied before the final flush),
    and a second request on the warmed cache reproduces the first."""
    from painlessinferenceacceleration_amd.pretrained_model_batch import LookaheadPreTrainedModel as BatchMixin
    from tests.oracle_engine import OracleBatchEngine
    from tests.tiny_model import noisy_copies, tiny_decisive_weights

    class BModel(BatchMixin):
        def __init__(self):
            self.engine = OracleBatchEngine(tiny_shape(), tiny_decisive_weights(0, torch.float32), max_length=256, n_slots=3, max_blocks=3)
            self.generation_config = type('G', (), {'pad_token_id': 0, 'eos_token_id': None, 'return_dict_in_generate': True})()
            self.lookahead_cache = LookaheadCache(eos_ids=[])

    shape = tiny_shape()
    rs = np.random.RandomState(8)
    B, P, n_new = 3, 14, 50
    ids = rs.randint(3, shape.vocab, size=(B, P)).astype(np.int64)
    truth = BModel().greedy_search(torch.from_numpy(ids), P + n_new, eos_token_id=None)[:, P:].tolist()
    stats = {}
    for overlap in (False, True):
        m = BModel()
        calls = []
        inner_async, inner_put = m.engine.mstep_async, m.lookahead_cache.stream_put_many
        m.engine.mstep_async = lambda blocks, eager=False: (calls.append('async'), inner_async(blocks))[1]
        m.lookahead_cache.stream_put_many = lambda *a, **k: (calls.append('put'), inner_put(*a, **k))[1]
        for b in range(B):
            for c in noisy_copies(ids[b, -2:].tolist() + truth[b], 6, 0.3, shape.vocab, seed=20 + b):
                m.lookahead_cache.put(c, branch_length=13, mode='output', idx=-1)
        dk = {'use_lookahead': True, 'decoding_length': 64, 'branch_length': 12, 'stop_words': {}, 'per_sample_budget': True,
              'overlap_trie_update': overlap}
        for rep in range(2):
            out = m.lookahead_generation(torch.from_numpy(ids), stopping_criteria=P + n_new, eos_token_id=[None], pad_token_id=0,
                                         return_dict_in_generate=True, decoding_kwargs=dict(dk))
            got = out.sequences.cpu().numpy()
            for b in range(B):
                assert got[b, P:P + n_new].tolist() == truth[b][:n_new], (overlap, rep, b)
            assert np.mean(out.kwargs['edls'][B:]) > 2
        stats[overlap] = m.lookahead_cache.stats()['n_nodes']
        # in-order: the update of the first tokens precedes the first pass; overlapped: every put follows the pass it hides under (and
        # the last one is applied before the final flush)
        assert calls[0] == ('async' if overlap else 'put') and 'async' in calls, calls[:6]
        stats[('puts', overlap)] = calls.count('put')
    assert stats[False] == stats[True] and stats[('puts', False)] == stats[('puts', True)]


def test_benchmark_harness_load_prompts_batch_chat_and_profile_hooks(tmp_path, capsys):
    """The rest of the reference harness (benchmarks/benchmark.py:79-100 load_prompts, :188-241 batch_chat, :397-441 profile hooks)
    on the oracle-backed model: a jsonl corpus round trip (token-id prompts, answers, warm-up ids, the max_length filter), batch_chat's
    off/on legs with identical outputs and the reference's log lines, cProfile over chat() and over the trie loop, torch.profiler."""
    from painlessinferenceacceleration_amd.benchmark import Benchmark
    m = DecisiveModel(torch.float32, max_length=400)
    rs = np.random.RandomState(12)
    prompts = [rs.randint(3, 512, size=n).tolist() for n in (30, 41, 36, 300)]
    b = Benchmark(model=m, eos=None)
    answers = b.save_answers(prompts[:3], max_new_tokens=40)
    b.save_prompts(str(tmp_path / 'q.jsonl'), prompts, answers=answers + [None])
    b.save_prompts(str(tmp_path / 'w.jsonl'), prompts[:3], answers=answers, ids=answers)
    b2 = Benchmark(model=m, eos=None)
    b2.load_prompts(str(tmp_path / 'q.jsonl'), str(tmp_path / 'w.jsonl'), max_length=100)
    assert b2.prompts == prompts[:3] and len(b2.answers) == 4 and b2.warmup_ids == answers and b2.warmup_prompts == prompts[:3]
    b2.warm_up(b2.warmup_ids, branch_length=12)
    r = b2.batch_chat(b2.prompts, max_new_tokens=40, decoding_length=32, branch_length=12, erase=False, batch_size=1)
    out = capsys.readouterr().out
    assert r['identical'] and r['speed_on'] > 0 and r['speed_off'] > 0
    on = [ln for ln in out.splitlines() if ln.startswith('lookahead:On ')]
    assert len(on) == 3 and all('edl:' in ln and 'speedup:' in ln for ln in on) and 'speed:' in out.splitlines()[-1]
    assert np.mean([float(ln.split('edl:')[1].split('/')[0]) for ln in on]) > 2.0          # the warmed trie's drafts were accepted
    txt = b2.naive_profile(b2.prompts[:1], use_lookahead=True, count=8, max_new_tokens=16, decoding_length=32, branch_length=12)
    assert 'function calls' in txt
    txt = b2.naive_profile_trie(LookaheadCache(), answers, prompts[:3], answers, decoding_length=32, branch_length=12, edl=4, count=8)
    assert 'function calls' in txt
    prof = b2.torch_profile(use_lookahead=True, trace_dir=str(tmp_path / 'prof'), prompts=b2.prompts * 2, max_new_tokens=8,
                            decoding_length=32, branch_length=12)
    assert prof is not None and os.path.isdir(str(tmp_path / 'prof'))
