# -*- coding: utf-8 -*-
"""ORACLE tooling (build container only): golden vectors of the REFERENCE for heads narrower than 128 features.

The reference's LlamaAttention is shape-generic (models/llama/modeling_llama.py:189-308: head_dim = hidden_size // num_heads, softmax
scale 1 / sqrt(head_dim), rotary pairs (d, d + head_dim / 2)); the MI355X kernels lay every head out as a 128-feature lane and run a
narrower head zero-padded inside it (include/lookahead_hip.h, la_head_lane_map).  These vectors pin that path — through the oracle — to
the reference's own forward: a tiny seeded Llama (plain N(0, 0.08) weights: attention matters to every logit) at head_dim 64 and 96,
the reference's lookahead_generation on it (fp32: tokens / dls / edls are reproducible; two requests, the second on a warm trie) and,
for the first forwards of each request, the draft ids / tree rows the reference fed and 64 columns of the logits it got back.

Writes tests/golden/llama_tiny_hd64_fp32.npz, llama_tiny_hd96_fp32.npz.
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gen_golden_model as gm            # noqa: E402
from oracle.tiny import TINY, tiny_weights           # noqa: E402  (build-free: no product import)

OUT = os.path.join(ROOT, 'tests', 'golden')
# name -> overrides of oracle.tiny.TINY (head_dim = hidden / n_heads; the reference's Llama is MHA-only)
CONFIGS = {'hd64': dict(hidden=256, n_heads=4, n_kv_heads=4), 'hd96': dict(hidden=384, n_heads=4, n_kv_heads=4)}
MAX_NEW = 72


def build(LlamaForCausalLM, over):
    from transformers import GenerationConfig, LlamaConfig
    c = dict(TINY); c.update(over)
    cfg = LlamaConfig(vocab_size=c['vocab'], hidden_size=c['hidden'], intermediate_size=c['ffn'], num_hidden_layers=c['n_layers'],
                      num_attention_heads=c['n_heads'], num_key_value_heads=c['n_kv_heads'], rms_norm_eps=c['rms_eps'],
                      max_position_embeddings=2048, pad_token_id=0, bos_token_id=1, eos_token_id=2, tie_word_embeddings=False)
    cfg.rope_scaling = None
    cfg.rope_theta = 10000.0
    cfg.pretraining_tp = 1
    model = LlamaForCausalLM(cfg).eval()
    missing, unexpected = model.load_state_dict(tiny_weights(0, torch.float32, cfg=over), strict=False)
    assert not unexpected and all('rotary' in m or 'inv_freq' in m for m in missing), (missing, unexpected)
    for mod in model.modules():
        if hasattr(mod, 'inv_freq'):
            assert mod.dim == c['hidden'] // c['n_heads']
    model.generation_config = GenerationConfig(pad_token_id=0, eos_token_id=2)
    model._extract_past_from_model_output = lambda outputs, standardize_cache_format=False: outputs.past_key_values
    return model


def run(name, over):
    from transformers import LogitsProcessorList, MaxLengthCriteria, StoppingCriteriaList
    LookaheadCache, _, LlamaForCausalLM = gm.import_reference()
    model = build(LlamaForCausalLM, over)
    prompt = gm.tiny_prompt()
    steps = []
    orig_forward = model.forward

    def rec_forward(*a, **kw):
        out = orig_forward(*a, **kw)
        steps.append({'ids': kw['input_ids'][0].tolist(), 'rows': None,
                      'logits': out.logits[0, :, :64].float().numpy().copy(),
                      'kv_in': 0 if kw.get('past_key_values') is None else int(kw['past_key_values'][0][0].shape[2])})
        return out
    model.forward = rec_forward
    orig_upd = model._lookahead_update_model_kwargs_for_generation

    def rec_upd(outputs, model_kwargs, **kw):
        mk = orig_upd(outputs, model_kwargs, **kw)
        st = steps[-1]
        dk = mk['decoding_kwargs']
        if 'decoding_masks' in dk and len(dk.get('decoding_ids', [])) == len(st['ids']) and st['kv_in'] > 0:
            m = np.asarray(dk['decoding_masks']).astype(np.int64)
            st['rows'] = [int(sum(int(b) << j for j, b in enumerate(r))) for r in m]
        return mk
    model._lookahead_update_model_kwargs_for_generation = rec_upd
    model.lookahead_cache = LookaheadCache()
    save = {'prompt': np.array(prompt), 'cfg': np.array([over['hidden'], over['n_heads'], over['n_kv_heads']]), 'max_new': np.array(MAX_NEW)}
    for r in range(2):
        steps.clear()
        ids = torch.tensor([prompt], dtype=torch.long)
        dk = {'use_lookahead': True, 'decoding_mode': 'hier', 'decoding_length': 64, 'branch_length': 12, 'max_query_length': 2,
              'stop_words': {}}
        with torch.no_grad():
            out = model.lookahead_generation(ids, logits_processor=LogitsProcessorList(),
                                             stopping_criteria=StoppingCriteriaList([MaxLengthCriteria(max_length=len(prompt) + MAX_NEW)]),
                                             pad_token_id=0, eos_token_id=2, return_dict_in_generate=True,
                                             attention_mask=torch.ones_like(ids), decoding_kwargs=dk, use_cache=True)
        save[f'r{r}_sequences'] = np.array(out.sequences[0].tolist())
        save[f'r{r}_dls'] = np.array(out.kwargs['dls'])
        save[f'r{r}_edls'] = np.array(out.kwargs['edls'])
        kept = 0
        for i, st in enumerate(steps):
            # the prefill and the first forwards that carried a draft tree
            if i == 0 or (st['rows'] is not None and len(st['ids']) > 1 and kept < 4):
                save[f'r{r}_s{i}_ids'] = np.array(st['ids'])
                save[f'r{r}_s{i}_kv'] = np.array(st['kv_in'])
                save[f'r{r}_s{i}_logits'] = st['logits'].astype(np.float32)
                if st['rows'] is not None:
                    save[f'r{r}_s{i}_rows'] = np.array(st['rows'], dtype=np.uint64)
                    kept += 1
        save[f'r{r}_steps'] = np.array(sorted(int(k.split('_')[1][1:]) for k in save if k.startswith(f'r{r}_s') and k.endswith('_ids')))
        print(name, 'run', r, 'dls', out.kwargs['dls'][:14], 'edls', out.kwargs['edls'][:14], 'recorded steps', save[f'r{r}_steps'].tolist())
    np.savez_compressed(os.path.join(OUT, f'llama_tiny_{name}_fp32.npz'), **save)


if __name__ == '__main__':
    torch.manual_seed(0)
    torch.set_num_threads(8)
    for name, over in CONFIGS.items():
        run(name, over)
