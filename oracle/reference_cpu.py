# -*- coding: utf-8 -*-
"""ORACLE tooling (test infrastructure, not product): the REFERENCE ITSELF as the CPU leg of bench.py.

When /root/reference is present (the build container; never the GPU box) `reference_lookahead_loop` imports the reference's own
classes in place — LookaheadCache (lookahead/lookahead/common/lookahead_cache.py), LookaheadPreTrainedModel.lookahead_generation
(common/pretrained_model.py:947-1268) and LlamaForCausalLM (models/llama/modeling_llama.py) — behind the transformers-5.x import shim
of SURVEY Appendix A, loads the SAME HF-named state dict the GPU run uses, warms its trie with the same copies and times its
verify steps.  bench.py reports it as `cpu_baseline.kind = "reference"`; where the reference is absent it falls back to the port
(oracle/llama_oracle.py::lookahead_generate, kind "port"), which tests/test_oracle_llama.py pins token for token to this loop.
Nothing is copied from the reference; nothing here is imported by the product package."""
import os
import sys
import time
import types

import numpy as np
import torch

REFERENCE_ROOT = '/root/reference/lookahead'


def cpu_model_name():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, 'lookahead', 'common', 'lookahead_cache.py'))


def import_reference():
    """-> (LookaheadCache, LookaheadPreTrainedModel, LlamaForCausalLM) of the reference, imported where they lie."""
    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    def _ga(n):
        if n.startswith('__'):
            raise AttributeError(n)
        return type(n, (object,), {})
    for name in ('transformers.generation.beam_constraints', 'transformers.generation.beam_search'):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__getattr__ = _ga
            m.__file__ = '<stub>'
            sys.modules[name] = m
    import transformers.generation.utils as gu
    for n in ('GreedySearchEncoderDecoderOutput', 'GreedySearchDecoderOnlyOutput', 'GreedySearchOutput', 'SampleOutput'):
        if not hasattr(gu, n):
            setattr(gu, n, type(n, (object,), {}))
    from lookahead.common.lookahead_cache import LookaheadCache
    from lookahead.common.pretrained_model import LookaheadPreTrainedModel
    from lookahead.models.llama.modeling_llama import LlamaForCausalLM
    return LookaheadCache, LookaheadPreTrainedModel, LlamaForCausalLM


def build_reference_llama(shape, state_dict, dtype=torch.bfloat16):
    """The reference's LlamaForCausalLM with `state_dict` (HF names) at `dtype`; rotary inv_freq kept in fp32 as from_pretrained leaves it."""
    from transformers import GenerationConfig, LlamaConfig
    _, _, LlamaForCausalLM = import_reference()
    cfg = LlamaConfig(vocab_size=shape.vocab, hidden_size=shape.hidden, intermediate_size=shape.ffn, num_hidden_layers=shape.n_layers,
                      num_attention_heads=shape.n_heads, num_key_value_heads=shape.n_kv_heads, rms_norm_eps=shape.rms_eps,
                      max_position_embeddings=4096, pad_token_id=0, bos_token_id=1, eos_token_id=2, tie_word_embeddings=False)
    cfg.rope_scaling = None
    cfg.rope_theta = float(getattr(shape, 'rope_theta', 10000.0))
    cfg.pretraining_tp = 1
    # built on the meta device and filled by assignment: no random init and no fp32 detour for a 7B model
    with torch.device('meta'):
        model = LlamaForCausalLM(cfg).eval()
    missing, unexpected = model.load_state_dict({k: v.to(dtype) for k, v in state_dict.items()}, strict=False, assign=True)
    assert not unexpected and all('rotary' in m or 'inv_freq' in m for m in missing), (missing, unexpected)
    for mod in model.modules():                            # non-persistent buffers are not in the state dict: re-create them (fp32, as from_pretrained leaves them)
        if hasattr(mod, 'inv_freq'):
            mod.inv_freq = 1.0 / (mod.base ** (torch.arange(0, mod.dim, 2, dtype=torch.int64).float() / mod.dim))
    left = [n for n, t in list(model.named_parameters()) + list(model.named_buffers()) if t.is_meta]
    assert not left, left
    model.generation_config = GenerationConfig(pad_token_id=0, eos_token_id=2)
    model._extract_past_from_model_output = lambda outputs, standardize_cache_format=False: outputs.past_key_values
    return model


def reference_lookahead_loop(shape, sd_cpu, prompt, copies, branch_length, decoding_length, verify_steps=5, threads=None,
                             dtype=torch.bfloat16):
    """bench.py's `cpu_baseline` leg with kind "reference": the reference's transformers CPU path on the full model, same prompt and
    trie warm-up as the GPU run, `verify_steps` verify steps after the prefill.  Accepted tok/s = sum(edls[1:]) / sum(fts[1:]) — the
    reference's own counters (pretrained_model.py:1256-1266)."""
    from transformers import LogitsProcessorList, MaxLengthCriteria, StoppingCriteriaList
    LookaheadCache, _, _ = import_reference()
    ncpu = os.cpu_count() or 1
    nt = threads or min(ncpu, 16)
    torch.set_num_threads(nt)
    model = build_reference_llama(shape, sd_cpu, dtype)
    cache = LookaheadCache(eos_ids=[None])
    for c in copies:
        cache.put([int(t) for t in c], branch_length=branch_length + 1, mode='output', idx=-1)
    model.lookahead_cache = cache
    # stop after `verify_steps` verify steps: count forward calls and cap max_length at what they can emit at most
    n_calls = {'n': 0}
    orig = model.forward

    def counted(*a, **kw):
        n_calls['n'] += 1
        return orig(*a, **kw)
    model.forward = counted

    class _StepCap(object):                                 # StoppingCriteria duck type: stop once prefill + verify_steps forwards ran
        def __call__(self, input_ids, scores, **kw):
            return n_calls['n'] >= verify_steps + 1
    P = len(prompt)
    ids = torch.tensor([list(prompt)], dtype=torch.long)
    t0 = time.time()
    with torch.no_grad():
        out = model.lookahead_generation(
            ids, logits_processor=LogitsProcessorList(),
            stopping_criteria=StoppingCriteriaList([MaxLengthCriteria(max_length=P + (verify_steps + 1) * (branch_length + 1) + 2), _StepCap()]),
            pad_token_id=0, eos_token_id=None, return_dict_in_generate=True, attention_mask=torch.ones_like(ids),
            decoding_kwargs={'use_lookahead': True, 'decoding_mode': 'hier', 'decoding_length': decoding_length, 'branch_length': branch_length,
                             'max_query_length': 2, 'stop_words': {}, 'debug_lookahead': False},
            use_cache=True)
    wall = time.time() - t0
    kw = out.kwargs
    fts, edls, dls = list(kw['fts']), list(kw['edls']), list(kw['dls'])
    t_dec, n_acc = float(sum(fts[1:])), int(sum(edls[1:]))
    return {'value': round(n_acc / max(t_dec, 1e-9), 3), 'unit': 'tokens/s', 'cores': nt, 'kind': 'reference',
            'ms_per_step': round(1e3 * t_dec / max(len(fts) - 1, 1), 1), 'dtype': str(dtype).replace('torch.', ''), 'cpu_model': cpu_model_name(),
            'host_cores': ncpu, 'verify_steps': len(fts) - 1, 'mean_accept_len': round(n_acc / max(len(fts) - 1, 1), 3),
            'mean_draft_len': round(float(np.mean(dls[1:])) if len(dls) > 1 else 0.0, 2),
            'prefill_s': round(float(fts[0]), 2), 'wall_s': round(wall, 1),
            'tokens': out.sequences[0, P:].tolist(),
            'sample': f'the reference itself (lookahead_generation + LookaheadCache + models/llama/modeling_llama.py, imported from {REFERENCE_ROOT}) on the '
                      f'full {shape.n_layers}-layer model: prefill of {P} tokens + {len(fts) - 1} verify steps, {str(dtype).replace("torch.", "")}, {nt} threads'}
