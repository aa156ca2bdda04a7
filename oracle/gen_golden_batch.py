# -*- coding: utf-8 -*-
"""ORACLE tooling: golden vectors for the batch (bs>1, cursor) twin of the path, produced by running the REFERENCE
classes in place (build container only): common/pretrained_model_batch.py:1002-1330 (loop), :664-759 (prepare
inputs / bat_get), :767-935 (per-sample accept scan), :937-980 (_early_stop) on top of
models/llama/modeling_llama_batch.py (fused QKV, rotation-table RoPE, pre-allocated KV written at cursors).

Writes tests/golden/llama_tiny_batch_{fp32,bf16}.npz: for each case (batch size, prompt lengths / left padding,
decoding_length) the padded prompts, attention masks, final sequences, dls / edls and, per decode step, the cursors,
batch indices, draft id lists and emitted tokens, plus a sample of the logits of the first decode steps.
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.gen_golden_model import OUT, TINY, import_reference, tiny_prompt, tiny_weights  # noqa: E402

# (name, batch size, valid prompt lengths (left-padded to the longest), decoding_length, max_new)
CASES = [
    ('b2', 2, [40, 40], 64, 64),
    ('b3pad', 3, [40, 33, 25], 64, 48),
    ('b4', 4, [24, 24, 24, 24], 64, 40),
    ('b3pad128', 3, [40, 33, 25], 128, 48),      # budget > 64 rows once samples retire: oracle-only case
    ('b4w256', 4, [24, 24, 24, 24], 256, 40),
]


def build_reference_batch_model(dtype):
    import_reference()
    from transformers import GenerationConfig, LlamaConfig
    from lookahead.models.llama.modeling_llama_batch import LlamaForCausalLM
    c = TINY
    cfg = LlamaConfig(vocab_size=c['vocab'], hidden_size=c['hidden'], intermediate_size=c['ffn'],
                      num_hidden_layers=c['n_layers'], num_attention_heads=c['n_heads'],
                      num_key_value_heads=c['n_kv_heads'], rms_norm_eps=c['rms_eps'], max_position_embeddings=2048,
                      pad_token_id=0, bos_token_id=1, eos_token_id=2, tie_word_embeddings=False)
    cfg.rope_scaling = None
    cfg.rope_theta = 10000.0
    cfg.pretraining_tp = 1
    model = LlamaForCausalLM(cfg).eval()
    missing, unexpected = model.load_state_dict(tiny_weights(0, torch.float32), strict=False)
    assert not unexpected and all('rotary' in m or 'inv_freq' in m or 'cached' in m for m in missing), (missing, unexpected)
    model = model.to(dtype)
    model.generation_config = GenerationConfig(pad_token_id=0, eos_token_id=2)
    return model


def case_prompts(bs, lengths):
    P = max(lengths)
    ids = np.zeros((bs, P), dtype=np.int64)
    am = np.zeros((bs, P), dtype=np.int64)
    for b, n in enumerate(lengths):
        ids[b, P - n:] = tiny_prompt(seed=1234 + 7 * b, n=n)
        am[b, P - n:] = 1
    return ids, am


def run(dtype, tag):
    from transformers import LogitsProcessorList, MaxLengthCriteria, StoppingCriteriaList
    LookaheadCache = import_reference()[0]
    model = build_reference_batch_model(dtype)
    save = {'cases': np.array([c[0] for c in CASES])}
    for name, bs, lengths, dl, max_new in CASES:
        ids, am = case_prompts(bs, lengths)
        P = ids.shape[1]
        steps = []
        orig_upd = model._lookahead_update_model_kwargs_for_generation
        orig_fwd = model.forward

        def rec_fwd(*a, **kw):
            out = orig_fwd(*a, **kw)
            rec_fwd.last = out.logits.float().numpy().copy()
            return out

        def rec_upd(outputs, model_kwargs, **kw):
            dk = model_kwargs['decoding_kwargs']
            prefill = model_kwargs.get('past_key_values', None) is None
            pre = {'cursors': list(dk.get('decoding_cursors') or []), 'batch_indices': list(dk.get('batch_indices') or []),
                   'ids': [list(x) for x in dk.get('decoding_ids', [])] if not prefill else []}
            mk = orig_upd(outputs, model_kwargs, **kw)
            pre['next'] = [list(map(int, x)) for x in mk['next_token_list']]
            pre['prefill'] = prefill
            pre['logits'] = rec_fwd.last
            steps.append(pre)
            return mk
        model.forward = rec_fwd
        model._lookahead_update_model_kwargs_for_generation = rec_upd
        model.lookahead_cache = LookaheadCache()
        runs = []
        for rep in range(2):            # second request: trie warmed by the first
            steps.clear()
            dk = {'use_lookahead': True, 'decoding_mode': 'hier', 'decoding_length': dl, 'branch_length': 12,
                  'max_query_length': 2, 'stop_words': {}}
            with torch.no_grad():
                out = model.lookahead_generation(torch.from_numpy(ids), logits_processor=LogitsProcessorList(),
                                                 stopping_criteria=StoppingCriteriaList([MaxLengthCriteria(max_length=P + max_new)]),
                                                 pad_token_id=0, eos_token_id=2, return_dict_in_generate=True,
                                                 attention_mask=torch.from_numpy(am), decoding_kwargs=dk, use_cache=True)
            runs.append((out.sequences.numpy().copy(), list(out.kwargs['dls']), list(out.kwargs['edls']), [dict(s) for s in steps]))
        model.forward = orig_fwd
        model._lookahead_update_model_kwargs_for_generation = orig_upd
        # attention layers cache a fused weight + a sequence-length counter on first use: rebuild for the next case
        save[f'{name}_ids'] = ids
        save[f'{name}_am'] = am
        save[f'{name}_cfg'] = np.array([bs, dl, max_new])
        for r, (seq, dls, edls, sts) in enumerate(runs):
            save[f'{name}_r{r}_sequences'] = seq
            save[f'{name}_r{r}_dls'] = np.array(dls)
            save[f'{name}_r{r}_edls'] = np.array(edls)
            save[f'{name}_r{r}_nsteps'] = np.array(len(sts))
            for i, st in enumerate(sts):
                if st['prefill']:
                    continue
                save[f'{name}_r{r}_s{i}_cursors'] = np.array(st['cursors'])
                save[f'{name}_r{r}_s{i}_bidx'] = np.array(st['batch_indices'])
                width = max(len(x) for x in st['ids'])
                save[f'{name}_r{r}_s{i}_ids'] = np.array(st['ids']).reshape(len(st['ids']), width)
                nx = np.full((len(st['next']), 16), -1, dtype=np.int64)
                for k, t in enumerate(st['next']):
                    nx[k, :len(t)] = t
                save[f'{name}_r{r}_s{i}_next'] = nx
                if i <= 3:
                    save[f'{name}_r{r}_s{i}_logits'] = st['logits'][:, :, :64].astype(np.float32)
        print(tag, name, 'dls', runs[-1][1][:16], 'edls', runs[-1][2][:16], 'steps', len(runs[-1][3]))
    np.savez_compressed(os.path.join(OUT, f'llama_tiny_batch_{tag}.npz'), **save)


if __name__ == '__main__':
    torch.manual_seed(0)
    torch.set_num_threads(8)
    run(torch.float32, 'fp32')
    run(torch.bfloat16, 'bf16')
