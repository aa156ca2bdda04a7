# -*- coding: utf-8 -*-
"""from_hf — the bridge to "the repo's existing transformers model wrappers" (north_star).  Tiny transformers Llama / Mistral /
Mixtral models are built from configs HERE (random init; transformers 5.x naming, fused Mixtral experts, rope_theta inside
rope_parameters) and pushed through the bridge (LlamaShape.from_hf + legacy_state_dict):

  * CPU (not gpu): the fp32 oracle fed by the bridge == the HF eager forward in fp32 to 2e-3 — config mapping, parameter naming,
    GQA grouping, RoPE base, RMSNorm flavour and the expert split are exact, with no bf16 noise in the way;
  * -m gpu: the engine built by <Wrapper>.from_hf == the bf16 oracle on the same bridged weights (tiny-model tolerance), and
    generate() runs through the wrapper."""
import numpy as np
import pytest
import torch

from oracle import llama_oracle as lo
from painlessinferenceacceleration_amd.llama_engine import LlamaShape, legacy_state_dict

TOL, TOL_TAIL, TAIL_FRAC = 2e-2, 3e-2, 0.10          # the stated 2e-2 (TOL) per row + the tiny seeded model's documented tail, see tests/test_gpu_mblock.py


def _hf(kind):
    import transformers
    torch.manual_seed(7)
    common = dict(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                  rms_norm_eps=1e-5, max_position_embeddings=1024, pad_token_id=0, bos_token_id=1, eos_token_id=2,
                  tie_word_embeddings=False, attn_implementation='eager')
    if kind == 'llama':
        cfg = transformers.LlamaConfig(num_key_value_heads=2, **common)
        m = transformers.LlamaForCausalLM(cfg)
    elif kind == 'mistral':
        common.update(hidden_size=512, num_attention_heads=4)
        cfg = transformers.MistralConfig(num_key_value_heads=2, sliding_window=4096, **common)
        m = transformers.MistralForCausalLM(cfg)
    else:
        cfg = transformers.MixtralConfig(num_key_value_heads=1, num_local_experts=8, num_experts_per_tok=2, sliding_window=None,
                                         rope_theta=1e6, **common)
        m = transformers.MixtralForCausalLM(cfg)
    m = m.eval()
    for p in m.parameters():           # HF init std 0.02 on 256 dims gives logits ~1e-2: scale up to the tiny-model regime
        if p.dim() >= 2:
            p.data.mul_(4.0)
    m = m.to(torch.bfloat16)             # weights rounded to bf16 once; the fp32 legs upcast them
    _restore_inv_freq(m)
    return m


def _restore_inv_freq(m):
    """from_pretrained(torch_dtype=...) leaves the non-persistent rotary inv_freq buffer in fp32; a blanket .to(bf16) rounds it
    (oracle/gen_golden_model.py makes the same repair on the reference classes)."""
    c = m.config
    rp = getattr(c, 'rope_parameters', None) or {}
    theta = getattr(c, 'rope_theta', None) or rp.get('rope_theta', 10000.0)
    dim = getattr(c, 'head_dim', None) or c.hidden_size // c.num_attention_heads
    inv = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.int64).float() / dim))
    for mod in m.modules():
        if hasattr(mod, 'inv_freq'):
            mod.inv_freq = inv.clone()
            if hasattr(mod, 'original_inv_freq'):
                mod.original_inv_freq = inv.clone()


def _prompt(seed, n):
    return np.random.RandomState(seed).randint(3, 512, size=n).tolist()


@pytest.mark.parametrize('kind', ['llama', 'mistral', 'mixtral'])
def test_bridge_shape_mapping_and_fp32_oracle_equals_hf_eager(kind):
    import copy
    m = _hf(kind)
    c = m.config
    s = LlamaShape.from_hf(c)
    assert (s.n_layers, s.hidden, s.n_heads, s.n_kv_heads, s.ffn, s.vocab) == \
        (c.num_hidden_layers, c.hidden_size, c.num_attention_heads, c.num_key_value_heads, c.intermediate_size, c.vocab_size)
    assert s.head_dim == 128 and s.rms_eps == c.rms_norm_eps
    assert s.n_experts == (8 if kind == 'mixtral' else 0) and s.norm_cast_first == (kind != 'llama')
    assert s.rope_theta == (1e6 if kind == 'mixtral' else 10000.0)
    sd = {k: v.float() for k, v in legacy_state_dict(m.state_dict(), s).items()}
    if kind == 'mixtral':
        assert 'model.layers.1.block_sparse_moe.experts.7.w2.weight' in sd and sd['model.layers.0.block_sparse_moe.gate.weight'].shape == (8, 256)
    P = 48
    prompt = _prompt(3, P)
    with torch.no_grad():
        m32 = copy.deepcopy(m).float()
        _restore_inv_freq(m32)
        ref = m32(input_ids=torch.tensor([prompt])).logits[0].float()
    got, _ = lo.OracleLlama(s, sd).forward(torch.tensor(prompt), torch.tril(torch.ones((P, P), dtype=torch.long)), None)
    err = (got.float() - ref).abs().max(1).values / ref.abs().max(1).values
    assert float(err.max()) < 2e-3, (kind, err)


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['llama', 'mistral', 'mixtral'])
def test_from_hf_engine_matches_bf16_oracle_on_bridged_weights(kind):
    from painlessinferenceacceleration_amd.modeling_llama import LlamaForCausalLM
    from tests.test_gpu_e2e import _check_rows
    m = _hf(kind)
    model = LlamaForCausalLM.from_hf(m, max_length=256)
    s = LlamaShape.from_hf(m.config)
    oracle = lo.OracleLlama(s, {k: v.to(torch.bfloat16) for k, v in legacy_state_dict(m.state_dict(), s).items()})
    P = 50
    prompt = _prompt(4, P)
    ref, _ = oracle.forward(torch.tensor(prompt), torch.tril(torch.ones((P, P), dtype=torch.long)), None)
    model.engine.prefill(prompt)
    got = model.engine.logits()[:P].float().cpu()
    if kind == 'mixtral':
        # an expert flip on a near-tie router row moves that row and the rows attending to it (tests/test_gpu_moe.py): the bulk of
        # the rows must sit at bf16 noise
        err = (got - ref.float()).abs().max(1).values / ref.float().abs().max(1).values
        assert float(err.median()) < TOL and float((err < 2 * TOL_TAIL).float().mean()) >= 0.7, err
    else:
        _check_rows(got, ref, range(P), f'from_hf {kind}', tol=TOL, tail_tol=TOL_TAIL, tail_frac=TAIL_FRAC)
    out = model.generate(input_ids=torch.tensor([prompt]), max_new_tokens=8,
                         decoding_kwargs={'use_lookahead': True, 'decoding_length': 64, 'branch_length': 12})
    assert out.shape[1] == P + 8
    assert model.generation_config.eos_token_id == 2


@pytest.mark.gpu
def test_from_pretrained_front_door_runs_the_reference_example_sequence(tmp_path):
    """examples/llama_example.py:19-69 with only the import changed: from_pretrained(model_dir, cache_dir, torch_dtype,
    low_cpu_mem_usage, device_map) on a checkpoint directory written by transformers' save_pretrained, then generate() with the
    example's keyword arguments, lookahead off / off / on / on.  All four replies must be the HF model's own greedy continuation."""
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples'))
    from generate_from_checkpoint import synthetic_checkpoint
    from painlessinferenceacceleration_amd.modeling_llama import LlamaForCausalLM
    from painlessinferenceacceleration_amd.modeling_llama_batch import LlamaForCausalLM as BatchLlama
    import transformers
    d = synthetic_checkpoint(str(tmp_path), layers=2, hidden=256, heads=2, ffn=512, vocab=512)
    hf = transformers.LlamaForCausalLM.from_pretrained(d, torch_dtype=torch.float32).eval()
    model = LlamaForCausalLM.from_pretrained(d, cache_dir='../', torch_dtype=torch.float16, low_cpu_mem_usage=True,
                                             device_map={'': 'cuda:0'}, max_length=512)
    # torch_dtype=torch.float16 — the reference example's own setting — now runs the float16 build of the library (round 4; before, fp16
    # checkpoints were silently computed in bfloat16)
    assert model.config.vocab_size == 512 and model.eval() is model and model.dtype == torch.float16
    assert LlamaForCausalLM.from_pretrained(d, torch_dtype=torch.bfloat16, max_length=128).dtype == torch.bfloat16
    input_ids = torch.randint(3, 512, (1, 20), generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref = hf.generate(input_ids, max_new_tokens=64, do_sample=False, pad_token_id=0, eos_token_id=None)[0].tolist()
    stop_words = set([5, 6])
    for use_lookahead in (False, False, True, True):
        out = model.generate(input_ids=input_ids.cuda(), attention_mask=torch.ones_like(input_ids).cuda(), position_ids=None,
                             pad_token_id=2, eos_token_id=None, use_cache=True, max_new_tokens=64, repetition_penalty=1.0,
                             do_sample=False, decoding_kwargs={'use_lookahead': use_lookahead, 'debug_lookahead': False,
                                                               'decoding_length': 64, 'branch_length': 12, 'stop_words': stop_words})
        assert out.device.type == 'cuda' and out[0].tolist() == ref, use_lookahead
    # the batch wrapper (benchmarks/llama_benchmark.py:25-29: device_map='auto')
    bm = BatchLlama.from_pretrained(d, cache_dir='../', torch_dtype=torch.float16, low_cpu_mem_usage=True, device_map='auto',
                                    max_length=512, max_batch=2)
    two = torch.cat([input_ids, input_ids.flip(1)], 0)
    with torch.no_grad():
        ref2 = [hf.generate(two[i:i + 1], max_new_tokens=32, do_sample=False, pad_token_id=0, eos_token_id=None)[0].tolist() for i in range(2)]
    out2 = bm.generate(input_ids=two.cuda(), attention_mask=torch.ones_like(two).cuda(), max_new_tokens=32, pad_token_id=2,
                       eos_token_id=None, decoding_kwargs={'use_lookahead': True, 'decoding_length': 64, 'branch_length': 12})
    assert [o.tolist() for o in out2] == ref2


def test_family_wrappers_check_the_family_and_map_the_sliding_window():
    """modeling_mixtral.py (the surface of the reference's models/mistral + models/mixtral drivers): the wrapper refuses a checkpoint
    of the other family, leaves the sliding window OFF by default (the reference's lookahead path feeds the full mask),
    and maps sliding_window='config' / an integer / kv_ring onto the engine's shape.  Config objects only: no engine is built."""
    from painlessinferenceacceleration_amd.modeling_mixtral import MistralForCausalLM, MixtralForCausalLM
    mis, mix = _hf('mistral').config, _hf('mixtral').config
    s = MistralForCausalLM._shape_of(mis, {})
    assert s.n_experts == 0 and s.norm_cast_first and s.sliding_window == 0
    kw = {'sliding_window': 'config', 'kv_ring': True, 'max_length': 256}
    s = MistralForCausalLM._shape_of(mis, kw)
    assert s.sliding_window == 4096 and 'sliding_window' not in kw and kw['kv_ring'] is True      # kv_ring travels on to the engine
    assert MistralForCausalLM._shape_of(mis, {'sliding_window': 100}).sliding_window == 100
    s = MixtralForCausalLM._shape_of(mix, {'sliding_window': 'config'})
    assert s.n_experts == 8 and s.top_k == 2 and s.rope_theta == 1e6 and s.sliding_window == 0     # the checkpoint has no window
    with pytest.raises(ValueError):
        MixtralForCausalLM._shape_of(mis, {})
    with pytest.raises(ValueError):
        MistralForCausalLM._shape_of(mix, {})
    with pytest.raises(ValueError):
        MistralForCausalLM._shape_of(mis, {'kv_ring': True})
    with pytest.raises(ValueError):
        MistralForCausalLM._shape_of(mis, {'sliding_window': -5})


@pytest.mark.gpu
def test_mistral_wrapper_window_and_ring_generate_like_the_full_cache_inside_the_window():
    """MistralForCausalLM.from_hf(..., sliding_window=W, kv_ring=True): while the context stays inside the window the windowed ring
    engine must emit exactly what the default (full-mask, linear cache) wrapper emits; MixtralForCausalLM.from_hf runs the MoE path."""
    from painlessinferenceacceleration_amd.modeling_mixtral import MistralForCausalLM, MixtralForCausalLM
    m = _hf('mistral')
    prompt = torch.tensor([_prompt(9, 40)])
    dk = {'use_lookahead': True, 'decoding_length': 64, 'branch_length': 12, 'stop_words': {}}
    full = MistralForCausalLM.from_hf(m, max_length=256)
    ring = MistralForCausalLM.from_hf(m, max_length=256, sliding_window=200, kv_ring=True)
    assert ring.engine.kv_ring and ring.shape.sliding_window == 200 and full.shape.sliding_window == 0
    a = full.generate(input_ids=prompt, max_new_tokens=60, decoding_kwargs=dict(dk), eos_token_id=None)
    b = ring.generate(input_ids=prompt, max_new_tokens=60, decoding_kwargs=dict(dk), eos_token_id=None)
    assert a[0].tolist() == b[0].tolist()
    moe = MixtralForCausalLM.from_hf(_hf('mixtral'), max_length=256)
    out = moe.generate(input_ids=prompt, max_new_tokens=16, decoding_kwargs=dict(dk), eos_token_id=None)
    assert out.shape[1] == 40 + 16
