# -*- coding: utf-8 -*-
"""ORACLE tooling: golden vectors for the model side, produced by running the REFERENCE classes
(build container only; /root/reference is imported in place behind the small transformers-5.x shim of
SURVEY Appendix A, nothing is copied).

Writes
  tests/golden/llama_tiny_fp32.npz / llama_tiny_bf16.npz
      reference LlamaForCausalLM (models/llama/modeling_llama.py) + LookaheadPreTrainedModel.lookahead_generation
      (common/pretrained_model.py:947-1268) on a tiny seeded Llama: sequences, dls, edls, per-step draft ids /
      row masks / argmax rows / emitted tokens, KV lengths, and sampled logits.
  tests/golden/accept_scan.json
      reference _lookahead_update_model_kwargs_for_generation (pretrained_model.py:764-892) on random trees with
      forced argmax rows: next_token_list, logit_indices, kept KV positions.
The tiny model's weights are a pure function of a numpy seed (tiny_weights below), so no weights are stored.
"""
import json
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'tests', 'golden')

TINY = dict(n_layers=2, hidden=256, n_heads=2, n_kv_heads=2, ffn=512, vocab=512, rms_eps=1e-5)


def tiny_weights(seed=0, dtype=torch.float32, std=0.08):
    """HF-named state dict of the tiny model from numpy's MT19937 (stable across platforms)."""
    rs = np.random.RandomState(seed)
    c = TINY
    hd = c['hidden'] // c['n_heads']

    def w(n, k):
        return torch.from_numpy((rs.standard_normal((n, k)) * std).astype(np.float32)).to(dtype)

    sd = {'model.embed_tokens.weight': w(c['vocab'], c['hidden'])}
    for i in range(c['n_layers']):
        p = f'model.layers.{i}.'
        sd[p + 'self_attn.q_proj.weight'] = w(c['n_heads'] * hd, c['hidden'])
        sd[p + 'self_attn.k_proj.weight'] = w(c['n_kv_heads'] * hd, c['hidden'])
        sd[p + 'self_attn.v_proj.weight'] = w(c['n_kv_heads'] * hd, c['hidden'])
        sd[p + 'self_attn.o_proj.weight'] = w(c['hidden'], c['n_heads'] * hd)
        sd[p + 'mlp.gate_proj.weight'] = w(c['ffn'], c['hidden'])
        sd[p + 'mlp.up_proj.weight'] = w(c['ffn'], c['hidden'])
        sd[p + 'mlp.down_proj.weight'] = w(c['hidden'], c['ffn'])
        sd[p + 'input_layernorm.weight'] = torch.from_numpy((1.0 + 0.1 * rs.standard_normal(c['hidden'])).astype(np.float32)).to(dtype)
        sd[p + 'post_attention_layernorm.weight'] = torch.from_numpy((1.0 + 0.1 * rs.standard_normal(c['hidden'])).astype(np.float32)).to(dtype)
    sd['model.norm.weight'] = torch.from_numpy((1.0 + 0.1 * rs.standard_normal(c['hidden'])).astype(np.float32)).to(dtype)
    sd['lm_head.weight'] = w(c['vocab'], c['hidden'])
    return sd


def tiny_prompt(seed=1234, n=40):
    rs = np.random.RandomState(seed)
    phrases = [rs.randint(3, TINY['vocab'], size=rs.randint(3, 8)).tolist() for _ in range(12)]
    out = []
    while len(out) < n:
        out.extend(phrases[rs.randint(0, len(phrases))])
    return out[:n]


def import_reference():
    sys.path.insert(0, '/root/reference/lookahead')

    def _ga(n):
        if n.startswith('__'):
            raise AttributeError(n)
        return type(n, (object,), {})
    for name in ('transformers.generation.beam_constraints', 'transformers.generation.beam_search'):
        m = types.ModuleType(name); m.__getattr__ = _ga; m.__file__ = '<stub>'; sys.modules[name] = m
    import transformers.generation.utils as gu
    for n in ('GreedySearchEncoderDecoderOutput', 'GreedySearchDecoderOnlyOutput', 'GreedySearchOutput', 'SampleOutput'):
        if not hasattr(gu, n):
            setattr(gu, n, type(n, (object,), {}))
    from lookahead.common.lookahead_cache import LookaheadCache
    from lookahead.common.pretrained_model import LookaheadPreTrainedModel
    from lookahead.models.llama.modeling_llama import LlamaForCausalLM
    return LookaheadCache, LookaheadPreTrainedModel, LlamaForCausalLM


def build_reference_model(LlamaForCausalLM, dtype):
    from transformers import LlamaConfig, GenerationConfig
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
    assert not unexpected and all('rotary' in m or 'inv_freq' in m for m in missing), (missing, unexpected)
    model = model.to(dtype)
    # from_pretrained(torch_dtype=...) (benchmarks/llama_benchmark.py:25-29) leaves the non-persistent rotary
    # inv_freq buffer in fp32; a blanket .to(bf16) would round it, so put the fp32 values back.
    for mod in model.modules():
        if hasattr(mod, 'inv_freq'):
            mod.inv_freq = 1.0 / (mod.base ** (torch.arange(0, mod.dim, 2, dtype=torch.int64).float() / mod.dim))
    model.generation_config = GenerationConfig(pad_token_id=0, eos_token_id=2)
    model._extract_past_from_model_output = lambda outputs, standardize_cache_format=False: outputs.past_key_values
    return model


def run_reference_generation(dtype, tag, max_new=96, warm=True):
    from transformers import LogitsProcessorList, StoppingCriteriaList, MaxLengthCriteria
    LookaheadCache, LPM, LlamaForCausalLM = import_reference()
    model = build_reference_model(LlamaForCausalLM, dtype)
    prompt = tiny_prompt()
    steps = []
    orig_forward = model.forward

    def rec_forward(*a, **kw):
        out = orig_forward(*a, **kw)
        steps.append({'ids': kw['input_ids'][0].tolist(), 'mask_shape': list(kw['attention_mask'].shape),
                      'tree_rows': None, 'argmax': torch.argmax(out.logits[0].float(), -1).tolist(),
                      'logits_sample': out.logits[0, :, :64].float().numpy().copy(),
                      'kv_in': 0 if kw.get('past_key_values') is None else int(kw['past_key_values'][0][0].shape[2])})
        return out
    model.forward = rec_forward
    orig_upd = model._lookahead_update_model_kwargs_for_generation

    def rec_upd(outputs, model_kwargs, **kw):
        mk = orig_upd(outputs, model_kwargs, **kw)
        st = steps[-1]
        st['next'] = [int(x) for x in (mk['next_token_list'][0] if isinstance(mk['next_token_list'][0], list) else mk['next_token_list'])]
        st['kv_out'] = int(mk['past_key_values'][0][0].shape[2])
        dk = mk['decoding_kwargs']
        if 'decoding_masks' in dk and len(dk.get('decoding_ids', [])) == len(st['ids']) and st['kv_in'] > 0:
            m = np.asarray(dk['decoding_masks']).astype(np.int64)
            st['tree_rows'] = [int(sum(int(b) << j for j, b in enumerate(r))) for r in m]
            st['sizes'] = [int(x) for x in dk.get('sizes', [])]
        return mk
    model._lookahead_update_model_kwargs_for_generation = rec_upd

    runs = []
    model.lookahead_cache = LookaheadCache()
    for rep in range(2 if warm else 1):      # second request runs on the trie warmed by the first
        steps.clear()
        ids = torch.tensor([prompt], dtype=torch.long)
        dk = {'use_lookahead': True, 'decoding_mode': 'hier', 'decoding_length': 64, 'branch_length': 12,
              'max_query_length': 2, 'stop_words': {}}
        with torch.no_grad():
            out = model.lookahead_generation(ids, logits_processor=LogitsProcessorList(),
                                             stopping_criteria=StoppingCriteriaList([MaxLengthCriteria(max_length=len(prompt) + max_new)]),
                                             pad_token_id=0, eos_token_id=2, return_dict_in_generate=True,
                                             attention_mask=torch.ones_like(ids), decoding_kwargs=dk, use_cache=True)
        runs.append({'sequences': out.sequences[0].tolist(), 'dls': list(out.kwargs['dls']), 'edls': list(out.kwargs['edls']),
                     'steps': [dict(s) for s in steps]})
    # plain greedy with the same model (reference lookahead == greedy in fp32)
    model.forward = orig_forward
    seq = list(prompt)
    with torch.no_grad():
        o = model(input_ids=torch.tensor([seq]), use_cache=True)
        past = o.past_key_values
        for _ in range(max_new):
            t = int(torch.argmax(o.logits[0, -1].float()))
            seq.append(t)
            if t == 2:
                break
            o = model(input_ids=torch.tensor([[t]]), past_key_values=past, use_cache=True)
            past = o.past_key_values
    save = {'prompt': np.array(prompt), 'greedy': np.array(seq), 'n_runs': np.array(len(runs))}
    for r, run in enumerate(runs):
        save[f'r{r}_sequences'] = np.array(run['sequences'])
        save[f'r{r}_dls'] = np.array(run['dls'])
        save[f'r{r}_edls'] = np.array(run['edls'])
        save[f'r{r}_nsteps'] = np.array(len(run['steps']))
        for i, st in enumerate(run['steps']):
            save[f'r{r}_s{i}_ids'] = np.array(st['ids'])
            save[f'r{r}_s{i}_argmax'] = np.array(st['argmax'])
            save[f'r{r}_s{i}_next'] = np.array(st['next'])
            save[f'r{r}_s{i}_kv'] = np.array([st['kv_in'], st['kv_out']])
            if st['tree_rows'] is not None:
                save[f'r{r}_s{i}_rows'] = np.array(st['tree_rows'], dtype=np.uint64)
            if i < 4:
                save[f'r{r}_s{i}_logits'] = st['logits_sample'].astype(np.float32)
    np.savez_compressed(os.path.join(OUT, f'llama_tiny_{tag}.npz'), **save)
    print(tag, 'dls', runs[-1]['dls'][:12], 'edls', runs[-1]['edls'][:12], 'greedy==lookahead:',
          [r['sequences'][:len(seq)] == seq[:len(r['sequences'])] for r in runs])


def accept_scan_vectors():
    """Reference accept scan on random trees (built through the reference trie) with forced argmax rows."""
    LookaheadCache, LPM, _ = import_reference()
    rs = np.random.RandomState(7)
    out = []

    class Fake(object):
        def _extract_past_from_model_output(self, outputs, standardize_cache_format=False):
            return outputs.past_key_values
        _update_cache = LPM._update_cache
        _update_cache_with_axis_2 = LPM._update_cache_with_axis_2

    from transformers import LogitsProcessorList
    for case in range(60):
        cache = LookaheadCache(eos_ids=[None])
        vocab = int(rs.choice([6, 12, 40]))
        for _ in range(rs.randint(1, 12)):
            cache.put(rs.randint(0, vocab, size=rs.randint(2, 30)).tolist(), branch_length=13, mode='output', idx=-1)
        q = rs.randint(0, vocab, size=2).tolist()
        ids, mask, sizes = cache.hier_get(q, decoding_length=int(rs.choice([8, 16, 64])), branch_length=12,
                                          min_output_size=8, mode='mix', idx=0)
        T = len(ids)
        if T < 2:
            continue
        # argmax rows: follow a random root-to-somewhere path with probability, else random tokens
        par = [-1] * T
        for i in range(1, T):
            par[i] = int(np.nonzero(mask[i, :i])[0][-1])
        am = rs.randint(0, vocab, size=T).tolist()
        cur = 0
        while rs.rand() < 0.8:
            kids = [j for j in range(1, T) if par[j] == cur]
            if not kids:
                break
            nxt = kids[rs.randint(0, len(kids))]
            am[cur] = ids[nxt]
            cur = nxt
        ctx = int(rs.randint(3, 20))
        V = vocab + 3
        logits = torch.full((1, T, V), -5.0)
        for t in range(T):
            logits[0, t, am[t]] = 5.0
        kv = torch.arange(ctx - 1 + T, dtype=torch.float32)[None, None, :, None].expand(1, 1, -1, 2).contiguous()
        outputs = types.SimpleNamespace(logits=logits, past_key_values=((kv, kv.clone()),))
        dk = {'decoding_ids': list(ids), 'decoding_masks': mask, 'dls': [], 'edls': [], 'decoding_qids': q, 'sizes': sizes}
        input_ids = torch.zeros((1, ctx), dtype=torch.long)
        mk = LPM._lookahead_update_model_kwargs_for_generation(Fake(), outputs, {'decoding_kwargs': dk},
                                                               logits_processor=LogitsProcessorList(), input_ids=input_ids)
        kept = mk['past_key_values'][0][0][0, 0, :, 0].long().tolist()
        out.append({'ids': [int(x) for x in ids], 'rows': [int(sum(int(b) << j for j, b in enumerate(r))) for r in mask],
                    'argmax': [int(x) for x in am], 'context_length': ctx,
                    'next_token_list': [int(x) for x in mk['next_token_list'][0]],
                    'kept_kv': kept, 'dls': dk['dls'], 'edls': dk['edls']})
    # the SURVEY §8a worked example
    with open(os.path.join(OUT, 'accept_scan.json'), 'w') as f:
        json.dump(out, f, separators=(',', ':'))
    print('accept vectors:', len(out))


if __name__ == '__main__':
    torch.manual_seed(0)
    torch.set_num_threads(8)
    accept_scan_vectors()
    run_reference_generation(torch.float32, 'fp32')
    run_reference_generation(torch.bfloat16, 'bf16')
    run_reference_generation(torch.float16, 'fp16')      # round 4: the dtype the reference's examples / benchmarks load
