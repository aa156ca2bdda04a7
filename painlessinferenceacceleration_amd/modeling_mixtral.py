# -*- coding: utf-8 -*-
"""Mistral / Mixtral wrappers on the MI355X verify engine: the surface of lookahead/lookahead/models/mistral/
modeling_mistral.py and models/mixtral/modeling_mixtral.py (both bs=1 drivers in the reference).  Same engine as Llama:
grouped-query attention is native to the attention / QKV kernels, the Mixtral MLP is the sparse-MoE path of
la_llama_step (router fused into the post-attention norm, one gated GEMM pair per expert, weighted bf16 accumulation in
expert order), RMSNorm uses the Mistral flavour.

What the family wrappers add to LlamaForCausalLM:
* the checkpoint's family is checked (a dense checkpoint under MixtralForCausalLM, or an MoE one under MistralForCausalLM, is a
  loading mistake the Llama wrapper would silently run);
* `sliding_window`: as in the reference, the window is NOT applied on the lookahead path by default (mistral/modeling_mistral.py:
  979-983 feeds the full rank-4 mask).  `sliding_window='config'` (the checkpoint's value, Mistral-7B-v0.1: 4096) or an
  integer turns on the transformers mask rule (visible iff pos_row - pos_key <= window) in the attention kernels, and
  `kv_ring=True` on top of it keeps a sequence's KV cache as a ring of window + one step of rows — memory O(window) instead of
  O(max_length), BASELINE config 3 (bitwise equal to the windowed full cache, tests/test_gpu_e2e.py).
"""
from .llama_engine import LlamaShape
from .modeling_llama import LlamaForCausalLM


def _window_of(cfg, sliding_window):
    """0 (off: the reference's behaviour), 'config' (the checkpoint's own window, 0 when it has none) or a positive integer."""
    if sliding_window in (None, 0, False):
        return 0
    if sliding_window == 'config':
        return int(getattr(cfg, 'sliding_window', 0) or 0)
    w = int(sliding_window)
    if w < 0:
        raise ValueError(f'sliding_window={sliding_window}: 0 / None (full attention), "config" or a positive window')
    return w


class _MistralFamily(LlamaForCausalLM):
    _moe = None            # True: the checkpoint must carry experts; False: it must not

    @classmethod
    def _shape_of(cls, cfg, kw):
        shape = LlamaShape.from_hf(cfg)
        if cls._moe is not None and bool(shape.n_experts) != cls._moe:
            kind = getattr(cfg, 'model_type', type(cfg).__name__)
            raise ValueError(f'{cls.__name__}: the checkpoint is a {kind} model with {shape.n_experts} experts per layer — '
                             f'load it with {"MixtralForCausalLM" if shape.n_experts else "MistralForCausalLM / LlamaForCausalLM"}')
        shape.sliding_window = _window_of(cfg, kw.pop('sliding_window', 0))
        if kw.get('kv_ring') and not shape.sliding_window:
            raise ValueError('kv_ring=True needs a sliding window (sliding_window="config" or an integer): the ring holds window + one step of rows')
        return shape

    @classmethod
    def random_init(cls, shape=None, sliding_window=0, **kw):
        shape = shape or cls._default_shape()
        if sliding_window:
            shape.sliding_window = _window_of(shape, sliding_window)
        return super().random_init(shape, **kw)


class MistralForCausalLM(_MistralFamily):
    _moe = False
    _default_shape = staticmethod(LlamaShape.mistral_7b)


class MixtralForCausalLM(_MistralFamily):
    _moe = True
    _default_shape = staticmethod(LlamaShape.mixtral_8x7b)
