# -*- coding: utf-8 -*-
"""ORACLE tooling: golden vectors for the GQA / sparse-MoE arithmetic (rows a15c), produced by running the REFERENCE
classes in place (build container only): models/mixtral/modeling_mixtral.py (MixtralForCausalLM: GQA attention :302-380,
MixtralSparseMoeBlock :692-759, rank-4 mask hook :1033-1036) and models/mistral/modeling_mistral.py (MistralForCausalLM).

Those wrappers cannot run their KV-cache path on the installed transformers (legacy DynamicCache calls), so the vectors
come from cache-free forwards (use_cache=False) under a full rank-4 0/1 mask that contains a causal prompt part followed
by a draft tree — numerically the same attention the lookahead step performs with a cache.
Writes tests/golden/moe_tiny_{fp32,bf16}.npz: ids, mask rows, logits [T, V] and router logits per layer.
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.gen_golden_model import OUT, import_reference  # noqa: E402
from oracle.tiny import TINY_GQA, TINY_MOE, moe_weights, random_tree  # noqa: E402  (build-free: no product import)


def build(kind, cfgd, dtype):
    import_reference()
    import transformers.utils as tu
    import transformers.utils.import_utils as iu
    for mod in (tu, iu):
        if not hasattr(mod, 'is_torch_fx_available'):
            setattr(mod, 'is_torch_fx_available', lambda: False)
    common = dict(vocab_size=cfgd['vocab'], hidden_size=cfgd['hidden'], intermediate_size=cfgd['ffn'],
                  num_hidden_layers=cfgd['n_layers'], num_attention_heads=cfgd['n_heads'],
                  num_key_value_heads=cfgd['n_kv_heads'], rms_norm_eps=cfgd['rms_eps'], max_position_embeddings=2048,
                  sliding_window=4096, rope_theta=cfgd['rope_theta'], pad_token_id=0, tie_word_embeddings=False)
    if kind == 'mixtral':
        from lookahead.models.mixtral import modeling_mixtral as mm
        cfg = mm.MixtralConfig(num_local_experts=cfgd['n_experts'], num_experts_per_tok=cfgd['top_k'], **common)
        cfg._attn_implementation = 'eager'
        model = mm.MixtralForCausalLM(cfg)
    else:
        from lookahead.models.mistral import modeling_mistral as mm
        cfg = mm.MistralConfig(**common)
        cfg._attn_implementation = 'eager'
        model = mm.MistralForCausalLM(cfg)
    model = model.eval()
    missing, unexpected = model.load_state_dict(moe_weights(cfgd, 0, torch.float32), strict=False)
    assert not unexpected and all('rotary' in m or 'inv_freq' in m for m in missing), (missing, unexpected)
    return model.to(dtype)


def main():
    for tag, dtype in (('fp32', torch.float32), ('bf16', torch.bfloat16)):
        save = {}
        for kind, cfgd in (('mixtral', TINY_MOE), ('mistral', TINY_GQA)):
            model = build(kind, cfgd, dtype)
            rs = np.random.RandomState(5)
            for case, (P, T) in enumerate([(24, 40), (3, 61), (50, 1)]):
                _, rows = random_tree(rs, T)
                n = P + T
                mask = np.zeros((n, n), dtype=np.int64)
                mask[:P, :P] = np.tril(np.ones((P, P), dtype=np.int64))
                mask[P:, :P] = 1
                for i in range(T):
                    for j in range(T):
                        mask[P + i, P + j] = (int(rows[i]) >> j) & 1
                ids = rs.randint(3, cfgd['vocab'], size=n)
                kw = dict(output_router_logits=True) if kind == 'mixtral' else {}
                with torch.no_grad():
                    out = model(input_ids=torch.from_numpy(ids)[None], attention_mask=torch.from_numpy(mask)[None, None],
                                use_cache=False, **kw)
                save[f'{kind}_{case}_ids'] = ids
                save[f'{kind}_{case}_mask'] = mask.astype(np.int8)
                save[f'{kind}_{case}_logits'] = out.logits[0].float().numpy()
                if kind == 'mixtral':
                    for li, rl in enumerate(out.router_logits):
                        save[f'{kind}_{case}_router{li}'] = rl.float().numpy()
                print(tag, kind, case, 'max|logit|', float(out.logits.abs().max()))
        np.savez_compressed(os.path.join(OUT, f'moe_tiny_{tag}.npz'), **save)


if __name__ == '__main__':
    torch.manual_seed(0)
    torch.set_num_threads(8)
    main()
