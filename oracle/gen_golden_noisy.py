# -*- coding: utf-8 -*-
"""ORACLE tooling: an end-to-end golden run of the REFERENCE loop with PARTIAL accepts and multi-branch trees
(build container only; /root/reference imported in place through the shim of oracle/gen_golden_model.py).

The plain tiny random model has ~3-ulp top-2 gaps, so its golden run (llama_tiny_*.npz) cannot be compared token for token
with a bf16 GPU run beyond a few steps, and its warm-trie request accepts every draft (edls = 13, 13, ...).  Here the tiny
model is the decisive "permutation LM" (tests/tiny_model.py::tiny_decisive_weights) and the reference trie is warmed, as
Benchmark.warm_up does (benchmarks/benchmark.py:159-169), with NOISY copies of the model's own greedy continuation: drafts are
multi-branch, some branches are wrong, accepts are partial — and every token / dls / edls of the reference's
lookahead_generation (common/pretrained_model.py:947-1268) is reproducible bit for bit by any correct implementation.

Writes tests/golden/llama_tiny_noisy_{fp32,bf16}.npz: prompt, greedy continuation, warm-up copies, and per request the
sequences, dls, edls and per-step draft ids / row masks / emitted tokens.

Round 3: the same recording for the other draft shapes the reference serves — decoding_mode 'par' / 'one'
(lookahead_cache.py:441-517, selected at pretrained_model.py:712-723) as llama_tiny_noisy_{par,one}_{tag}.npz, and the
reference's best published setting decoding_length=128, branch_length=32 (lookahead/README.md:100) as
llama_tiny_noisy_dl128_{tag}.npz (row masks wider than 64 bits are stored as two uint64 words per row: rows_lo / rows_hi).
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gen_golden_model as gm            # noqa: E402
from oracle.tiny import TINY, noisy_copies, tiny_decisive_weights      # noqa: E402  (build-free: no product import)

OUT = os.path.join(ROOT, 'tests', 'golden')
MAX_NEW, RHO, COPIES, BL, DL = 120, 0.3, 8, 12, 64
# (file suffix, decoding_mode, decoding_length, branch_length, max_new)
VARIANTS = [('', 'hier', 64, 12, 120), ('_par', 'par', 64, 12, 120), ('_one', 'one', 64, 12, 120), ('_dl128', 'hier', 128, 32, 160)]


def run(dtype, tag, suffix='', decoding_mode='hier', DL=DL, BL=BL, MAX_NEW=MAX_NEW):
    from transformers import LogitsProcessorList, MaxLengthCriteria, StoppingCriteriaList
    LookaheadCache, LPM, LlamaForCausalLM = gm.import_reference()
    model = gm.build_reference_model(LlamaForCausalLM, torch.float32)
    missing, unexpected = model.load_state_dict(tiny_decisive_weights(0, torch.float32), strict=False)
    assert not unexpected
    model = model.to(dtype)
    for mod in model.modules():
        if hasattr(mod, 'inv_freq'):
            mod.inv_freq = 1.0 / (mod.base ** (torch.arange(0, mod.dim, 2, dtype=torch.int64).float() / mod.dim))
    prompt = gm.tiny_prompt(seed=4321, n=40)
    # ground truth: plain greedy of the reference model
    seq = list(prompt)
    with torch.no_grad():
        o = model(input_ids=torch.tensor([seq]), use_cache=True)
        past = o.past_key_values
        for _ in range(MAX_NEW + 40):
            t = int(torch.argmax(o.logits[0, -1].float()))
            seq.append(t)
            o = model(input_ids=torch.tensor([[t]]), past_key_values=past, use_cache=True)
            past = o.past_key_values
    truth = seq[len(prompt):]
    copies = noisy_copies(prompt[-2:] + truth, COPIES, RHO, TINY['vocab'], seed=99)
    steps = []
    orig_upd = model._lookahead_update_model_kwargs_for_generation

    def rec_upd(outputs, model_kwargs, **kw):
        mk = orig_upd(outputs, model_kwargs, **kw)
        dk = mk['decoding_kwargs']
        st = {'next': [int(x) for x in mk['next_token_list'][0]], 'ids': None, 'rows': None}
        if 'decoding_masks' in dk and dk.get('dls') and dk['dls'][-1] > 1:
            m = np.asarray(dk['decoding_masks']).astype(np.int64)
            st['ids'] = [int(x) for x in dk['decoding_ids']]
            st['rows'] = [int(sum(int(b) << j for j, b in enumerate(r[:64]))) for r in m]
            st['rows_hi'] = [int(sum(int(b) << j for j, b in enumerate(r[64:128]))) for r in m]
        steps.append(st)
        return mk
    model._lookahead_update_model_kwargs_for_generation = rec_upd
    model.lookahead_cache = LookaheadCache(eos_ids=[2])
    if decoding_mode == 'par':
        # REFERENCE DEFECT: par_get builds its mask with np.ones(...) (float64, lookahead_cache.py:480); the accept step then
        # slices a list with float cumsums (pretrained_model.py:816-819) and raises TypeError — 'par' cannot complete one verify
        # step in the reference as published.  The recording casts the mask to int64 (what hier_get returns) and changes
        # nothing else; the draft ids / mask VALUES are the reference's.
        _par = model.lookahead_cache.par_get

        def par_get_int(*a, **k):
            ids_, m_, sz_ = _par(*a, **k)
            return ids_, np.asarray(m_).astype(np.int64), sz_
        model.lookahead_cache.par_get = par_get_int
    for c in copies:
        model.lookahead_cache.put(c, branch_length=BL + 1, mode='output', idx=-1)
    runs = []
    for rep in range(2):                 # the second request also sees what the first one put into the trie
        steps.clear()
        ids = torch.tensor([prompt], dtype=torch.long)
        dk = {'use_lookahead': True, 'decoding_mode': decoding_mode, 'decoding_length': DL, 'branch_length': BL, 'max_query_length': 2,
              'stop_words': {}}
        with torch.no_grad():
            out = model.lookahead_generation(ids, logits_processor=LogitsProcessorList(),
                                             stopping_criteria=StoppingCriteriaList([MaxLengthCriteria(max_length=len(prompt) + MAX_NEW)]),
                                             pad_token_id=0, eos_token_id=2, return_dict_in_generate=True,
                                             attention_mask=torch.ones_like(ids), decoding_kwargs=dk, use_cache=True)
        runs.append({'sequences': out.sequences[0].tolist(), 'dls': list(out.kwargs['dls']), 'edls': list(out.kwargs['edls']),
                     'steps': [dict(s) for s in steps]})
    save = {'prompt': np.array(prompt), 'truth': np.array(truth), 'copies': np.array(copies), 'n_runs': np.array(len(runs)),
            'max_new': np.array(MAX_NEW), 'decoding_length': np.array(DL), 'branch_length': np.array(BL),
            'decoding_mode': np.array(decoding_mode)}
    for r, run_ in enumerate(runs):
        save[f'r{r}_sequences'] = np.array(run_['sequences'])
        save[f'r{r}_dls'] = np.array(run_['dls'])
        save[f'r{r}_edls'] = np.array(run_['edls'])
        save[f'r{r}_nsteps'] = np.array(len(run_['steps']))
        for i, st in enumerate(run_['steps']):
            save[f'r{r}_s{i}_next'] = np.array(st['next'])
            if st['ids'] is not None:
                save[f'r{r}_s{i}_ids'] = np.array(st['ids'])
                save[f'r{r}_s{i}_rows'] = np.array(st['rows'], dtype=np.uint64)
                if DL > 64:
                    save[f'r{r}_s{i}_rows_hi'] = np.array(st['rows_hi'], dtype=np.uint64)
    np.savez_compressed(os.path.join(OUT, f'llama_tiny_noisy{suffix}_{tag}.npz'), **save)
    for r, run_ in enumerate(runs):
        e = run_['edls'][1:]
        d = run_['dls'][1:]
        print(tag, suffix, f'run {r}: steps {len(e)} dls {d[:10]} edls {e[:10]} partial accepts {sum(1 < x < BL + 1 for x in e)} '
                   f'full {sum(x == BL + 1 for x in e)} single {sum(x == 1 for x in e)} == greedy {run_["sequences"] == (prompt + truth)[:len(run_["sequences"])]}')
    return runs


if __name__ == '__main__':
    torch.manual_seed(0)
    torch.set_num_threads(8)
    only = sys.argv[1:]          # optional: suffixes to (re)generate, e.g. _par _dl128 ('' = the original hier 64/12 run)
    for suffix, dm, dl, bl, mn in VARIANTS:
        if only and (suffix or 'hier') not in only:
            continue
        a = run(torch.float32, 'fp32', suffix, dm, dl, bl, mn)
        b = run(torch.bfloat16, 'bf16', suffix, dm, dl, bl, mn)
        assert [r['sequences'] for r in a] == [r['sequences'] for r in b] and [r['edls'] for r in a] == [r['edls'] for r in b], \
            'the decisive model must decode identically in fp32 and bf16'
        if suffix == '':         # round 4: the reference's own dtype (benchmarks/llama_benchmark.py:27, examples/llama_example.py:19)
            h = run(torch.float16, 'fp16', suffix, dm, dl, bl, mn)
            assert [r['sequences'] for r in a] == [r['sequences'] for r in h] and [r['edls'] for r in a] == [r['edls'] for r in h], \
                'the decisive model must decode identically in fp32 and fp16'
