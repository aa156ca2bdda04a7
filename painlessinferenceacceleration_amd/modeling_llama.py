# -*- coding: utf-8 -*-
"""LlamaForCausalLM on the MI355X verify engine: the model-wrapper surface of
lookahead/lookahead/models/llama/modeling_llama.py (generate / lookahead_generation / lookahead_cache) with the
forward living in liblookahead_hip.so."""
from types import SimpleNamespace

import torch

from .llama_engine import LlamaShape, LlamaVerifyEngine, legacy_state_dict, load_hf_checkpoint, random_weights
from .lookahead_cache import LookaheadCache
from .pretrained_model import LookaheadPreTrainedModel


class LlamaForCausalLM(LookaheadPreTrainedModel):
    def __init__(self, shape, state_dict, max_length=2048, device='cuda:0', eos_token_id=2, pad_token_id=0,
                 attn_split=0, gemm_cfg=None, consume_state_dict=False, balanced=True, fuse=0, max_blocks=0, kv_ring=False, dtype=None):
        self.shape = shape
        self.engine = LlamaVerifyEngine(shape, state_dict, max_length=max_length, device=device,
                                        attn_split=attn_split, gemm_cfg=gemm_cfg,
                                        consume_state_dict=consume_state_dict, balanced=balanced, fuse=fuse,
                                        max_blocks=max_blocks, kv_ring=kv_ring, dtype=dtype)
        self.generation_config = SimpleNamespace(eos_token_id=eos_token_id, pad_token_id=pad_token_id,
                                                 return_dict_in_generate=False)
        self.config = SimpleNamespace(is_encoder_decoder=False, vocab_size=shape.vocab)
        self.lookahead_cache = LookaheadCache()
        self.device = torch.device(device)

    @classmethod
    def _shape_of(cls, cfg, kw):
        """LlamaShape of a transformers config.  Family wrappers (modeling_mixtral.py) override this to check the checkpoint's family and to
        consume their own keywords (sliding_window=...) from kw before the engine sees it."""
        return LlamaShape.from_hf(cfg)

    @classmethod
    def from_hf(cls, hf_model, **kw):
        """Wrap a transformers LlamaForCausalLM (weights are repacked into HBM; the HF module is not used afterwards)."""
        shape = cls._shape_of(hf_model.config, kw)
        kw.setdefault('eos_token_id', getattr(hf_model.config, 'eos_token_id', 2))
        kw.setdefault('pad_token_id', getattr(hf_model.config, 'pad_token_id', 0) or 0)
        return cls(shape, legacy_state_dict(hf_model.state_dict(), shape), **kw)

    @classmethod
    def from_pretrained(cls, model_dir, *unused_args, device_map=None, torch_dtype=None, max_length=4096, **kw):
        """The reference examples' front door (examples/llama_example.py:19-24, benchmarks/llama_benchmark.py:25-29):
        LlamaForCausalLM.from_pretrained(model_dir, cache_dir=..., torch_dtype=..., low_cpu_mem_usage=True, device_map=...).
        The checkpoint's tensors are repacked straight into HBM (no nn.Module is built); the engine computes in `torch_dtype` —
        torch.float16 (the reference's own setting) or torch.bfloat16, each its own build of the library; None = the checkpoint's
        dtype (fp32 checkpoints run as bfloat16).  device_map picks the GPU ({"": "cuda:0"} / "auto" / None ->
        cuda:0).  max_length = KV capacity in tokens (prompt + generation)."""
        for k in ('cache_dir', 'low_cpu_mem_usage', 'trust_remote_code', 'revision', 'use_safetensors', 'attn_implementation'):
            kw.pop(k, None)
        device = 'cuda:0'
        if isinstance(device_map, dict) and device_map:
            device = str(next(iter(device_map.values())))
        elif isinstance(device_map, str) and device_map.startswith('cuda'):
            device = device_map
        if device.isdigit():
            device = f'cuda:{device}'
        cfg, sd = load_hf_checkpoint(model_dir)
        shape = cls._shape_of(cfg, kw)
        kw.setdefault('eos_token_id', getattr(cfg, 'eos_token_id', 2))
        kw.setdefault('pad_token_id', getattr(cfg, 'pad_token_id', 0) or 0)
        model = cls(shape, legacy_state_dict(sd, shape), max_length=max_length, device=device, consume_state_dict=True,
                    dtype=torch_dtype if torch_dtype in (torch.float16, torch.bfloat16) else None, **kw)
        model.config = cfg
        return model

    # nn.Module-style no-ops the reference scripts call on a loaded model
    def eval(self):
        return self

    def half(self):
        return self

    def to(self, *a, **k):
        return self

    def cuda(self, *a, **k):
        return self

    @property
    def dtype(self):
        return self.engine.dtype

    @classmethod
    def random_init(cls, shape, seed=0, device='cuda:0', decisive=False, **kw):
        return cls(shape, random_weights(shape, seed=seed, device=device, decisive=decisive), device=device,
                   consume_state_dict=True, **kw)
