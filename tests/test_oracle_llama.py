# -*- coding: utf-8 -*-
"""Pin the model-side ORACLE (oracle/llama_oracle.py) against vectors recorded from the reference classes
(oracle/gen_golden_model.py: LlamaForCausalLM + LookaheadPreTrainedModel.lookahead_generation run on CPU)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import llama_oracle as lo
from oracle.trie_oracle import TrieOracle
from tests.tiny_model import GOLDEN, load_golden, tiny_shape, tiny_weights


def _rows_to_mask(rows):
    T = len(rows)
    return np.array([[(int(r) >> j) & 1 for j in range(T)] for r in rows], dtype=np.int64)


def test_accept_scan_matches_reference_vectors():
    vecs = json.load(open(os.path.join(GOLDEN, 'accept_scan.json')))
    assert len(vecs) >= 40
    partial = 0
    for v in vecs:
        mask = _rows_to_mask(v['rows'])
        toks, rows = lo.accept_scan(v['ids'], mask, v['argmax'])
        assert toks == v['next_token_list'], v
        assert lo.kv_keep_positions(v['context_length'], len(v['ids']) - 1, rows) == v['kept_kv'], v
        assert v['edls'] == [len(toks)] and v['dls'] == [len(v['ids'])]
        partial += 1 < len(toks) < len(v['ids'])
    assert partial >= 5      # the vectors exercise partial acceptance, not only 0 / all


def test_survey_worked_example():
    """Tree(1) with [1,2,3],[1,2,4]; argmax rows {0->2, 1->4, 3->9}; ctx=5."""
    mask = _rows_to_mask([1, 3, 7, 11])
    toks, rows = lo.accept_scan([1, 2, 3, 4], mask, [2, 4, 0, 9])
    assert toks == [2, 4, 9] and rows == [0, 1, 3]
    assert lo.kv_keep_positions(5, 3, rows) == [0, 1, 2, 3, 4, 5, 7]


@pytest.mark.parametrize('tag,dtype', [('fp32', torch.float32), ('bf16', torch.bfloat16), ('fp16', torch.float16)])
def test_oracle_loop_matches_reference_generation(tag, dtype):
    g = load_golden(tag)
    torch.set_num_threads(4)
    model = lo.OracleLlama(tiny_shape(), tiny_weights(0, torch.float32))
    model.w = {k: v.to(dtype) for k, v in model.w.items()}
    model.dtype = dtype
    cache = TrieOracle()
    prompt = g['prompt'].tolist()
    max_length = len(prompt) + 96
    for r in range(int(g['n_runs'])):
        rec = []
        out = lo.lookahead_generate(model, cache, prompt, max_length, eos_token_id=2, record=rec)
        assert out['sequences'] == g[f'r{r}_sequences'].tolist()
        assert out['dls'] == g[f'r{r}_dls'].tolist() and out['edls'] == g[f'r{r}_edls'].tolist()
        assert len(rec) == int(g[f'r{r}_nsteps'])
        for i, st in enumerate(rec):
            assert st['ids'] == g[f'r{r}_s{i}_ids'].tolist(), (r, i)
            assert st['next'] == g[f'r{r}_s{i}_next'].tolist(), (r, i)
            assert st['argmax'] == g[f'r{r}_s{i}_argmax'].tolist(), (r, i)
            if f'r{r}_s{i}_rows' in g.files:
                assert st['rows'] == [int(x) for x in g[f'r{r}_s{i}_rows']], (r, i)
    if dtype == torch.float32:
        n_new = len(g['greedy']) - len(prompt)
        assert lo.greedy_generate(model, prompt, n_new) == g['greedy'].tolist()
        assert out['sequences'][:len(g['greedy'])] == g['greedy'].tolist()[:len(out['sequences'])]


def test_oracle_forward_logits_match_reference_sample():
    g = load_golden('fp32')
    model = lo.OracleLlama(tiny_shape(), tiny_weights(0, torch.float32))
    prompt = g['prompt'].tolist()
    P = len(prompt)
    logits, _ = model.forward(torch.tensor(prompt), torch.tril(torch.ones((P, P), dtype=torch.long)), None)
    ref = g['r0_s0_logits']
    assert np.abs(logits[:, :64].numpy() - ref).max() < 2e-4 * max(1.0, np.abs(ref).max())


# ------------------------------------------------------------------------------------------ batch twin
def _load_batch(tag):
    return np.load(os.path.join(GOLDEN, f'llama_tiny_batch_{tag}.npz'))


@pytest.mark.parametrize('tag,dtype', [('fp32', torch.float32), ('bf16', torch.bfloat16)])
def test_oracle_batch_loop_matches_reference_batch_generation(tag, dtype):
    """pretrained_model_batch.lookahead_generation on modeling_llama_batch (bs 2/3/4, left padding, budgets 64/128/256):
    sequences, dls, edls and every step's cursors / batch indices / padded draft ids / emitted tokens."""
    g = _load_batch(tag)
    torch.set_num_threads(4)
    sd = {k: v.to(dtype) for k, v in tiny_weights(0, torch.float32).items()}
    model = lo.OracleLlamaBatch(tiny_shape(), sd)
    for name in g['cases'].tolist():
        bs, dl, max_new = [int(x) for x in g[f'{name}_cfg']]
        ids, am = g[f'{name}_ids'], g[f'{name}_am']
        cache = TrieOracle()
        for r in range(2):
            rec = []
            out = lo.lookahead_generate_batch(model, cache, ids, am, ids.shape[1] + max_new, eos_token_id=2, pad_token_id=0,
                                              decoding_length=dl, branch_length=12, record=rec)
            assert out['sequences'].tolist() == g[f'{name}_r{r}_sequences'].tolist(), (name, r)
            assert out['dls'] == g[f'{name}_r{r}_dls'].tolist() and out['edls'] == g[f'{name}_r{r}_edls'].tolist(), (name, r)
            assert len(rec) + 1 == int(g[f'{name}_r{r}_nsteps'])
            for i, st in enumerate(rec, start=1):
                assert st['cursors'] == g[f'{name}_r{r}_s{i}_cursors'].tolist(), (name, r, i)
                assert st['bidx'] == g[f'{name}_r{r}_s{i}_bidx'].tolist(), (name, r, i)
                assert st['ids'] == g[f'{name}_r{r}_s{i}_ids'].tolist(), (name, r, i)
                nx = g[f'{name}_r{r}_s{i}_next']
                assert st['next'] == [[int(t) for t in row if t >= 0] for row in nx], (name, r, i)
                key = f'{name}_r{r}_s{i}_logits'
                if key in g.files:
                    ref = g[key]
                    tol = 2e-4 if dtype == torch.float32 else 0.0
                    assert np.abs(st['logits'][:, :, :64] - ref).max() <= tol * max(1.0, np.abs(ref).max()), (name, r, i)


def test_oracle_batch_equals_per_sample_bs1_semantics():
    """Each sample of a batch run generates exactly what plain greedy decoding of that sample generates (fp32):
    the property the batch HIP path is held to at sizes without a reference run."""
    g = _load_batch('fp32')
    model = lo.OracleLlama(tiny_shape(), tiny_weights(0, torch.float32))
    ids, am = g['b3pad_ids'], g['b3pad_am']
    seqs = g['b3pad_r0_sequences']
    P = ids.shape[1]
    for b in range(ids.shape[0]):
        prompt = ids[b][am[b] == 1].tolist()
        n_gen = int((seqs[b, P:] != 0).sum())
        gre = lo.greedy_generate(model, prompt, n_gen)
        assert gre[len(prompt):] == seqs[b, P:P + n_gen].tolist(), b


# ------------------------------------------------------------------------------------- GQA / sparse MoE (a15c)
@pytest.mark.parametrize('tag,dtype', [('fp32', torch.float32), ('bf16', torch.bfloat16)])
@pytest.mark.parametrize('kind', ['mixtral', 'mistral'])
def test_oracle_gqa_and_moe_match_reference_forward(kind, tag, dtype):
    """Reference MixtralForCausalLM / MistralForCausalLM (cache-free forward under a prompt+tree rank-4 mask) against
    the oracle forward: GQA repeat_kv attention, rope_theta, router softmax/top-2/renormalise, expert order."""
    from tests.tiny_model import TINY_GQA, TINY_MOE, moe_shape, moe_weights
    g = np.load(os.path.join(GOLDEN, f'moe_tiny_{tag}.npz'))
    cfg = TINY_MOE if kind == 'mixtral' else TINY_GQA
    sd = {k: v.to(dtype) for k, v in moe_weights(cfg, 0, torch.float32).items()}
    model = lo.OracleLlama(moe_shape(cfg), sd)
    for case in range(3):
        ids, mask, ref = g[f'{kind}_{case}_ids'], g[f'{kind}_{case}_mask'].astype(np.int64), g[f'{kind}_{case}_logits']
        logits, _ = model.forward(torch.from_numpy(ids), torch.from_numpy(mask), None)
        got = logits.float().numpy()
        if dtype == torch.float32:
            assert np.abs(got - ref).max() < 2e-4 * np.abs(ref).max(), (kind, case)
        else:
            # bf16: same rounding points; allow a few ulps where CPU kernels pick different accumulation orders
            assert np.abs(got - ref).max() <= 2 ** -5 * 2 and (got.argmax(-1) == ref.argmax(-1)).mean() > 0.97, (kind, case)
        if kind == 'mixtral' and dtype == torch.float32:
            rl = g[f'{kind}_{case}_router1']
            assert np.abs(model.last_router_logits.float().numpy() - rl).max() < 1e-4 * max(1.0, np.abs(rl).max())


def test_oracle_sequential_processor_path_matches_reference():
    """RepetitionPenaltyLogitsProcessor(1.3): the reference's sequential accept walk (pretrained_model.py:825-875)."""
    from transformers import LogitsProcessorList, RepetitionPenaltyLogitsProcessor
    g = np.load(os.path.join(GOLDEN, 'llama_tiny_fp32_rep.npz'))
    model = lo.OracleLlama(tiny_shape(), tiny_weights(0, torch.float32))
    cache = TrieOracle()
    prompt = g['prompt'].tolist()
    procs = LogitsProcessorList([RepetitionPenaltyLogitsProcessor(float(g['penalty']))])
    for r in range(2):
        out = lo.lookahead_generate(model, cache, prompt, len(prompt) + 64, eos_token_id=2, logits_processor=procs)
        assert out['sequences'] == g[f'r{r}_sequences'].tolist()
        assert out['dls'] == g[f'r{r}_dls'].tolist() and out['edls'] == g[f'r{r}_edls'].tolist()


@pytest.mark.parametrize('tag,dtype,suffix', [(t, d, sfx) for sfx in ('', '_par', '_one', '_dl128') for t, d in (('fp32', torch.float32), ('bf16', torch.bfloat16))] +
                         [('fp16', torch.float16, '')])          # round 4: the reference's own dtype, hier 64 / 12
def test_oracle_loop_matches_reference_run_with_partial_accepts(tag, dtype, suffix):
    """oracle/gen_golden_noisy.py: the reference loop on the decisive tiny model with a NOISY warm trie — multi-branch trees,
    23 partially accepted steps in the first request.  Tokens, dls, edls and every step's draft ids / row masks / emitted
    tokens of the oracle loop over the trie oracle must equal the recording.  Variants: decoding_mode 'par' / 'one'
    (lookahead_cache.py:441-517) and the reference's best published setting decoding_length=128, branch_length=32
    (lookahead/README.md:100: trees of up to 128 rows, up to 33 tokens accepted per step)."""
    from tests.tiny_model import tiny_decisive_weights
    g = np.load(os.path.join(GOLDEN, f'llama_tiny_noisy{suffix}_{tag}.npz'))
    dm, dl, bl = str(g['decoding_mode']), int(g['decoding_length']), int(g['branch_length'])
    torch.set_num_threads(4)
    model = lo.OracleLlama(tiny_shape(), tiny_decisive_weights(0, dtype))
    cache = TrieOracle(eos_ids=[2])
    for c in g['copies'].tolist():
        cache.put(c, branch_length=bl + 1, mode='output', idx=-1)
    prompt = g['prompt'].tolist()
    max_length = len(prompt) + int(g['max_new'])
    partial = 0
    for r in range(int(g['n_runs'])):
        rec = []
        out = lo.lookahead_generate(model, cache, prompt, max_length, eos_token_id=2, record=rec, decoding_mode=dm,
                                    decoding_length=dl, branch_length=bl)
        assert out['sequences'] == g[f'r{r}_sequences'].tolist()
        assert out['dls'] == g[f'r{r}_dls'].tolist() and out['edls'] == g[f'r{r}_edls'].tolist()
        assert len(rec) == int(g[f'r{r}_nsteps'])
        for i, st in enumerate(rec):
            assert st['next'] == g[f'r{r}_s{i}_next'].tolist(), (r, i)
            if f'r{r}_s{i}_ids' in g.files:
                assert st['ids'] == g[f'r{r}_s{i}_ids'].tolist(), (r, i)
                want = [int(x) for x in g[f'r{r}_s{i}_rows']]
                if f'r{r}_s{i}_rows_hi' in g.files:
                    want = [lo_ | (int(hi) << 64) for lo_, hi in zip(want, g[f'r{r}_s{i}_rows_hi'])]
                assert [int(x) for x in st['rows']] == want, (r, i)
        partial += sum(1 < e < bl + 1 for e in out['edls'][1:])
    assert partial >= 20
    if dl > 64:
        assert max(max(g[f'r{r}_dls'].tolist()) for r in range(int(g['n_runs']))) > 64      # the fixture really has wide trees


def test_attention_scale_in_fp16_needs_the_division():
    """fp16 build (LA_DTYPE 1): fp16(x / sqrt(128)) and fp16(x * fp32(1 / sqrt(128))) differ on 52 of the 65536 bit patterns, so
    attn_scale() of csrc/la_common.h divides in that build — fp32 division of the upcast value, one rounding, which is what torch
    does for `half_tensor / python_float`."""
    import math
    bits = np.arange(65536, dtype=np.uint16)
    x = bits.view(np.float16)
    fin = np.isfinite(x)
    with np.errstate(invalid='ignore'):
        x32 = x.astype(np.float32)
        div = (x32 / np.float32(math.sqrt(128))).astype(np.float16)
        mul = (x32 * np.float32(0.088388346135616302490234375)).astype(np.float16)
    assert int(((div.view(np.uint16) != mul.view(np.uint16)) & fin).sum()) == 52
    t = (torch.from_numpy(x.copy()) / math.sqrt(128)).numpy()
    assert int(((t.view(np.uint16) != div.view(np.uint16)) & fin).sum()) == 0          # torch == the fp32 division


def test_attention_scale_as_multiply_is_exact():
    """The bf16 attention kernels compute bf16(scores / sqrt(head_dim)) (modeling_llama.py:270) as bf16(x * fp32(1 / sqrt(head_dim)))
    (attn_scale / la_qk_scale, csrc/la_common.h): the two agree for every finite bf16 x at every head_dim the engine accepts a model
    with (even, 8 .. 128; la_llama_create repeats this check for the value it is given), so the multiply is a bit-exact restatement."""
    import math
    import numpy as np
    import torch
    bits = (np.arange(65536, dtype=np.uint32) << 16)
    x = torch.from_numpy(bits.view(np.float32).copy())
    fin = torch.isfinite(x)
    div = (x / 11.313708498984761).to(torch.bfloat16).view(torch.int16)
    mul = (x * torch.tensor(np.float32(0.088388346135616302490234375))).to(torch.bfloat16).view(torch.int16)
    assert bool((div[fin] == mul[fin]).all())
    xb = x[fin].to(torch.bfloat16)
    for hd in range(8, 129, 2):
        ref = (xb / math.sqrt(hd)).view(torch.int16)                          # what the reference's eager graph does
        mul = (xb.float() * torch.tensor(np.float32(1.0 / math.sqrt(hd)))).to(torch.bfloat16).view(torch.int16)
        assert bool((ref == mul).all()), hd


@pytest.mark.parametrize('name', ['hd64', 'hd96'])
def test_oracle_matches_reference_at_narrow_heads(name):
    """head_dim 64 / 96 (LlamaAttention is shape-generic, modeling_llama.py:189-308): the oracle against the reference's own run
    (oracle/gen_golden_headdim.py) — logits of the prefill and of the recorded tree forwards (fp32, 2e-4), then tokens / dls / edls of
    two consecutive requests.  These vectors are what pins the padded-lane path of the engine (tests/test_gpu_e2e.py) to the reference."""
    g = np.load(os.path.join(GOLDEN, f'llama_tiny_{name}_fp32.npz'))
    hidden, nh, nkv = [int(x) for x in g['cfg']]
    over = dict(hidden=hidden, n_heads=nh, n_kv_heads=nkv)
    shape = tiny_shape(**over)
    assert shape.head_dim == int(name[2:])
    torch.set_num_threads(4)
    model = lo.OracleLlama(shape, tiny_weights(0, torch.float32, cfg=over))
    prompt = g['prompt'].tolist()
    P = len(prompt)
    cache = TrieOracle()
    for r in range(2):
        rec = []
        out = lo.lookahead_generate(model, cache, prompt, P + int(g['max_new']), eos_token_id=2, record=rec)
        assert out['sequences'] == g[f'r{r}_sequences'].tolist()
        assert out['dls'] == g[f'r{r}_dls'].tolist() and out['edls'] == g[f'r{r}_edls'].tolist()
        # replay the recorded forwards: the keys committed before step i are the sequence's first kv tokens
        seq = g[f'r{r}_sequences'].tolist()
        for i in g[f'r{r}_steps'].tolist():
            ids, kv, ref = g[f'r{r}_s{i}_ids'].tolist(), int(g[f'r{r}_s{i}_kv']), g[f'r{r}_s{i}_logits']
            if kv == 0:
                T = len(ids)
                logits, _ = model.forward(torch.tensor(ids), torch.tril(torch.ones((T, T), dtype=torch.long)), None)
            else:
                _, past = model.forward(torch.tensor(seq[:kv]), torch.tril(torch.ones((kv, kv), dtype=torch.long)), None)
                rows = [int(x) for x in g[f'r{r}_s{i}_rows']]
                T = len(ids)
                tree = torch.tensor([[(rows[a] >> b) & 1 for b in range(T)] for a in range(T)], dtype=torch.long)
                full = torch.cat([torch.ones((T, kv), dtype=torch.long), tree], 1)
                logits, _ = model.forward(torch.tensor(ids), full, past)
            assert np.abs(logits[:, :64].numpy() - ref).max() < 2e-4 * max(1.0, np.abs(ref).max()), (r, i)


def test_oracle_batch_sequential_processor_path_matches_reference():
    """Batch twin with RepetitionPenaltyLogitsProcessor(1.3): the per-sample sequential accept walk of
    pretrained_model_batch.py:814-931 (processors see the padded row up to the accepted token) and the batch-wise processor
    call of the prefill (:783); bs 2 / 3 with left padding, budgets 64 / 256, two consecutive requests each."""
    from transformers import LogitsProcessorList, RepetitionPenaltyLogitsProcessor
    g = np.load(os.path.join(GOLDEN, 'llama_tiny_batch_fp32_rep.npz'))
    torch.set_num_threads(4)
    model = lo.OracleLlamaBatch(tiny_shape(), tiny_weights(0, torch.float32))
    procs = LogitsProcessorList([RepetitionPenaltyLogitsProcessor(float(g['penalty']))])
    partial = 0
    for name in g['cases'].tolist():
        bs, dl, max_new = [int(x) for x in g[f'{name}_cfg']]
        ids, am = g[f'{name}_ids'], g[f'{name}_am']
        cache = TrieOracle()
        for r in range(2):
            out = lo.lookahead_generate_batch(model, cache, ids, am, ids.shape[1] + max_new, eos_token_id=2, pad_token_id=0,
                                              decoding_length=dl, branch_length=12, logits_processor=procs)
            assert out['sequences'].tolist() == g[f'{name}_r{r}_sequences'].tolist(), (name, r)
            assert out['dls'] == g[f'{name}_r{r}_dls'].tolist() and out['edls'] == g[f'{name}_r{r}_edls'].tolist(), (name, r)
            partial += sum(1 for d, e in zip(out['dls'], out['edls']) if 1 < e < min(d, 13))
    assert partial > 0          # the fixture exercises partially accepted trees, not only full chains
