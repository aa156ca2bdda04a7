# -*- coding: utf-8 -*-
"""Mistral / Mixtral wrappers on the MI355X verify engine: the surface of lookahead/lookahead/models/mistral/
modeling_mistral.py and models/mixtral/modeling_mixtral.py (both bs=1 drivers in the reference).  Same engine as Llama:
grouped-query attention is native to the attention / QKV kernels, the Mixtral MLP is the sparse-MoE path of
la_llama_step (router fused into the post-attention norm, one gated GEMM pair per expert, weighted bf16 accumulation in
expert order), RMSNorm uses the Mistral flavour.  As in the reference, the sliding window is NOT applied on the
lookahead path (mistral/modeling_mistral.py:979-983 feeds the full rank-4 mask)."""
from .llama_engine import LlamaShape
from .modeling_llama import LlamaForCausalLM


class MistralForCausalLM(LlamaForCausalLM):
    @classmethod
    def random_init(cls, shape=None, **kw):
        return super().random_init(shape or LlamaShape.mistral_7b(), **kw)


class MixtralForCausalLM(LlamaForCausalLM):
    @classmethod
    def random_init(cls, shape=None, **kw):
        return super().random_init(shape or LlamaShape.mixtral_8x7b(), **kw)
