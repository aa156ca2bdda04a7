# -*- coding: utf-8 -*-
"""Secondary measurement (not the BASELINE metric): accepted tokens/s of the cursor-batch path on ONE MI355X as the
number of sequences sharing the 64-row verify block grows.  Same synthetic workload as bench.py per sequence
(Llama-2-7B shape, decisive weights, phrase-bank prompts, noisy-copy trie warm-up); decoding_length is chosen so the
reference's budget rule ((decoding_length // B) // B rows per sample) fills the block: 64*B.

    python scripts/bench_batch.py --batches 1,2,4,8 --steps 48
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import noisy_copies, phrase_prompt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batches', default='1,2,4,8')
    ap.add_argument('--steps', type=int, default=48)
    ap.add_argument('--warmup', type=int, default=6)
    ap.add_argument('--prompt-len', type=int, default=512)
    ap.add_argument('--layers', type=int, default=0)
    args = ap.parse_args()
    from painlessinferenceacceleration_amd.llama_engine import LlamaShape
    from painlessinferenceacceleration_amd.lookahead_cache import LookaheadCache
    from painlessinferenceacceleration_amd.modeling_llama_batch import LlamaForCausalLM

    shape = LlamaShape.llama2_7b()
    if args.layers:
        shape.n_layers = args.layers
    Bs = [int(x) for x in args.batches.split(',')]
    K, W, P, BL = args.steps, args.warmup, args.prompt_len, 12
    n_truth = (K + W) * (BL + 1) + 8
    max_length = P + n_truth + 130
    model = LlamaForCausalLM.random_init(shape, seed=0, max_length=max_length, max_batch=max(Bs), eos_token_id=None, decisive=True)
    eng = model.engine
    prompts = [phrase_prompt(1234 + r, P, shape.vocab) for r in range(max(Bs))]
    ids = torch.tensor(prompts)
    t0 = time.time()
    truth = model.greedy_search(ids, P + n_truth, eos_token_id=None)[:, P:].tolist()
    print(f'[setup] batch greedy of {len(prompts)} x {n_truth} tokens: {time.time() - t0:.1f}s', file=sys.stderr)
    import gc
    gc.collect()
    gc.freeze()          # see bench.py: a generation-2 collection stalls the loop for tens of ms
    rows_out = []
    for B in Bs:
        cache = LookaheadCache(eos_ids=[None])
        for r in range(B):
            for c in noisy_copies(prompts[r][-2:] + truth[r], 12, 0.3, shape.vocab, seed=99 + r):
                cache.put(c, branch_length=BL + 1, mode='output', idx=-1)
            cache.put(prompts[r][1:], branch_length=BL + 1, mode='input', idx=r)
        eng.reset_slot(-1)
        first = eng.bprefill_many({r: prompts[r] for r in range(B)})
        seqs = [list(prompts[r]) + [first[r]] for r in range(B)]
        DL = 64 * B                                      # (DL // B) // B * B = 64 rows per block
        edls, dls = [], []

        def one_step():
            drafts = cache.bat_get_packed([s[-2:] for s in seqs], decoding_length=max(DL // B, 1), branch_length=BL,
                                          mode='mix', indices=list(range(B)), decoding_mode='hier')
            segs = [(r, d[0], d[1], 0, 16) for r, d in enumerate(drafts)]
            out = eng.bstep(segs)
            for r in range(B):
                seqs[r].extend(out[r])
                cache.stream_put(out[r], branch_length=BL + 1, final=False, mode='output', idx=r)
                edls.append(len(out[r])); dls.append(len(drafts[r][0]))

        for _ in range(W):
            one_step()
        n0 = len(edls)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(K):
            one_step()
        torch.cuda.synchronize()
        el = time.time() - t0
        ok = all(seqs[r][P:P + n_truth] == truth[r][:len(seqs[r]) - P] for r in range(B))
        row = {'sequences_per_block': B, 'rows_per_sequence': round(float(np.mean(dls[n0:])), 1),
               'accepted_tokens_per_sec': round(sum(edls[n0:]) / el, 1), 'ms_per_step': round(1e3 * el / K, 3),
               'mean_accept_len_per_sequence': round(float(np.mean(edls[n0:])), 2), 'equals_greedy': ok}
        rows_out.append(row)
        print(json.dumps(row), flush=True)
    print(json.dumps({'workload': 'Llama-2-7B bf16 cursor-batch, one MI355X, 64-row verify block shared by B sequences',
                      'rows': rows_out}))


if __name__ == '__main__':
    main()
