# -*- coding: utf-8 -*-
"""painlessinferenceacceleration_amd — MI355X-native LOOKAHEAD trie-draft / tree-verify decoding.

Drop-in surface (reference: alipay/PainlessInferenceAcceleration, lookahead/):
    LookaheadCache                         lookahead/common/lookahead_cache.py
    LookaheadPreTrainedModel.lookahead_generation   lookahead/common/pretrained_model.py
    LlamaForCausalLM                       lookahead/models/llama/modeling_llama.py
    modeling_llama_batch.LlamaForCausalLM  lookahead/models/llama/modeling_llama_batch.py (bs>1, cursor batch)
All compute lives in liblookahead_hip.so (csrc/, hand-written gfx950 HIP + the native trie).
"""
from ._lib import LIB_PATH, LookaheadHipError  # noqa: F401  (raises at import if the library is not built)
from .lookahead_cache import LookaheadCache  # noqa: F401
from .lookahead_generation_utils import (GenerationMode, LookaheadDecoderOnlyOutput,  # noqa: F401
                                         LookaheadGenerationConfig)

__all__ = ['LookaheadCache', 'GenerationMode', 'LookaheadDecoderOnlyOutput', 'LookaheadGenerationConfig',
           'LookaheadHipError', 'LIB_PATH']


def __getattr__(name):          # torch-dependent pieces are imported lazily
    if name in ('LlamaForCausalLM',):
        from .modeling_llama import LlamaForCausalLM
        return LlamaForCausalLM
    if name in ('BatchLlamaForCausalLM',):
        from .modeling_llama_batch import LlamaForCausalLM
        return LlamaForCausalLM
    if name in ('LookaheadPreTrainedModel',):
        from .pretrained_model import LookaheadPreTrainedModel
        return LookaheadPreTrainedModel
    if name in ('LlamaVerifyEngine', 'LlamaShape'):
        from . import llama_engine
        return getattr(llama_engine, name)
    raise AttributeError(name)
