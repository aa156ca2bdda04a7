# -*- coding: utf-8 -*-
"""Batch LlamaForCausalLM on the MI355X verify engine: the wrapper surface of
lookahead/lookahead/models/llama/modeling_llama_batch.py (generate / lookahead_generation over [bs, P] prompts with
per-sample cursors) with the forward living in liblookahead_hip.so (la_llama_bstep)."""
from types import SimpleNamespace

import torch

from .llama_engine import LlamaShape, LlamaVerifyEngine, legacy_state_dict, random_weights
from .lookahead_cache import LookaheadCache
from .pretrained_model_batch import LookaheadPreTrainedModel


class LlamaForCausalLM(LookaheadPreTrainedModel):
    def __init__(self, shape, state_dict, max_length=2048, max_batch=8, device='cuda:0', eos_token_id=2, pad_token_id=0,
                 attn_split=0, gemm_cfg=None, consume_state_dict=False, balanced=True, max_blocks=None, kv_ring=False):
        self.shape = shape
        self.engine = LlamaVerifyEngine(shape, state_dict, max_length=max_length, device=device, attn_split=attn_split,
                                        gemm_cfg=gemm_cfg, consume_state_dict=consume_state_dict, balanced=balanced,
                                        n_slots=max_batch,
                                        max_blocks=min(max_batch, 8) if max_blocks is None else max_blocks, kv_ring=kv_ring)
        self.generation_config = SimpleNamespace(eos_token_id=eos_token_id, pad_token_id=pad_token_id,
                                                 return_dict_in_generate=False)
        self.config = SimpleNamespace(is_encoder_decoder=False, vocab_size=shape.vocab)
        self.lookahead_cache = LookaheadCache()
        self.device = torch.device(device)

    @classmethod
    def from_hf(cls, hf_model, **kw):
        shape = LlamaShape.from_hf(hf_model.config)
        kw.setdefault('eos_token_id', getattr(hf_model.config, 'eos_token_id', 2))
        kw.setdefault('pad_token_id', getattr(hf_model.config, 'pad_token_id', 0) or 0)
        return cls(shape, legacy_state_dict(hf_model.state_dict(), shape), **kw)

    @classmethod
    def random_init(cls, shape, seed=0, device='cuda:0', decisive=False, **kw):
        return cls(shape, random_weights(shape, seed=seed, device=device, decisive=decisive), device=device,
                   consume_state_dict=True, **kw)
