# -*- coding: utf-8 -*-
"""Front-door demo: the call sequence of the reference's examples/llama_example.py (from_pretrained -> generate with
decoding_kwargs, lookahead off / on) against the MI355X engine.  The reference script itself runs after changing ONE line,

    - from lookahead.models.llama.modeling_llama import LlamaForCausalLM
    + from painlessinferenceacceleration_amd.modeling_llama import LlamaForCausalLM

(INTEGRATION.md).  This container has no checkpoints and no network, so without --model-dir a small random Llama checkpoint
(config.json + model.safetensors, "permutation LM" weights so that greedy decoding is decisive) is written to a temp dir first.

    python examples/generate_from_checkpoint.py [--model-dir /path/to/llama] [--max-new-tokens 128]
"""
import argparse
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from painlessinferenceacceleration_amd.modeling_llama import LlamaForCausalLM      # noqa: E402


def synthetic_checkpoint(path, layers=4, hidden=1024, heads=8, ffn=2816, vocab=4096):
    """save_pretrained() of a random transformers Llama whose lm_head is a permutation of the embedding"""
    import transformers
    torch.manual_seed(0)
    cfg = transformers.LlamaConfig(vocab_size=vocab, hidden_size=hidden, intermediate_size=ffn, num_hidden_layers=layers,
                                   num_attention_heads=heads, num_key_value_heads=heads, rms_norm_eps=1e-5,
                                   max_position_embeddings=4096, pad_token_id=0, bos_token_id=1, eos_token_id=2,
                                   tie_word_embeddings=False)
    m = transformers.LlamaForCausalLM(cfg)
    sd = m.state_dict()
    for k in sd:
        if k.endswith('o_proj.weight') or k.endswith('down_proj.weight'):
            sd[k].mul_(0.005)
    order = 3 + torch.randperm(vocab - 3)
    perm = torch.arange(vocab)
    perm[order] = torch.roll(order, -1)
    head = torch.empty_like(sd['model.embed_tokens.weight'])
    head[perm] = sd['model.embed_tokens.weight']
    sd['lm_head.weight'].copy_(head)
    m.to(torch.bfloat16).save_pretrained(path, safe_serialization=True)
    return path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model-dir', default=None)
    ap.add_argument('--max-new-tokens', type=int, default=128)
    args = ap.parse_args()
    tmp = None
    model_dir = args.model_dir
    if model_dir is None:
        tmp = tempfile.TemporaryDirectory()
        model_dir = synthetic_checkpoint(tmp.name)
    model = LlamaForCausalLM.from_pretrained(model_dir, cache_dir='../', torch_dtype=torch.float16, low_cpu_mem_usage=True,
                                             device_map={'': 'cuda:0'})
    vocab = model.config.vocab_size
    input_ids = torch.randint(3, vocab, (1, 24), generator=torch.Generator().manual_seed(1)).cuda()
    attention_mask = torch.ones_like(input_ids)
    replies = []
    for use_lookahead in (False, False, True, True):          # the second lookahead request runs on a warm trie
        t0 = time.time()
        out = model.generate(input_ids=input_ids, attention_mask=attention_mask, position_ids=None,
                             pad_token_id=model.generation_config.pad_token_id, eos_token_id=None, use_cache=True,
                             max_new_tokens=args.max_new_tokens, repetition_penalty=1.0, do_sample=False,
                             decoding_kwargs={'use_lookahead': use_lookahead, 'debug_lookahead': False, 'decoding_length': 64,
                                              'branch_length': 12, 'stop_words': set()})
        dt = time.time() - t0
        new = out[0, input_ids.size(-1):].tolist()
        replies.append(new)
        print(f'lookahead:{use_lookahead} time:{dt:.3f}s speed:{len(new) / dt:.1f}token/s first tokens:{new[:8]}')
    assert all(r == replies[0] for r in replies), 'lookahead must not change the greedy output'
    print('lookahead on/off outputs identical')


if __name__ == '__main__':
    main()
