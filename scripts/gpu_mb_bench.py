# -*- coding: utf-8 -*-
"""Step time of the multi-block verify step (la_llama_mstep) on one MI355X at the Llama-2-7B shape: B sequences x 64-row trees
per step for B = 1..8, and prompt prefill (512 tokens) as 64-row steps vs chains of 8 blocks.  Wall clock around
K synchronous steps (graph launch + d2h of the result block), context ~512-600 keys per sequence.

    python scripts/gpu_mb_bench.py [--model 7b|13b|mistral] [--layers N]
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
from painlessinferenceacceleration_amd.llama_engine import LlamaShape, LlamaVerifyEngine, random_weights   # noqa: E402
from tests.gpu_utils import random_tree      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='7b')
    ap.add_argument('--layers', type=int, default=0)
    ap.add_argument('--steps', type=int, default=12)
    args = ap.parse_args()
    shape = {'7b': LlamaShape.llama2_7b, '13b': LlamaShape.llama2_13b, 'mistral': LlamaShape.mistral_7b}[args.model]()
    if args.layers:
        shape.n_layers = args.layers
    sd = random_weights(shape, seed=0, device='cuda:0', decisive=True)
    eng = LlamaVerifyEngine(shape, sd, max_length=1024, n_slots=8, max_blocks=8, consume_state_dict=True)
    rs = np.random.RandomState(0)
    P = 512
    prompts = [rs.randint(3, shape.vocab, size=P).tolist() for _ in range(8)]
    W = 2 * shape.n_params_no_embed()
    out = {'model': args.model, 'layers': shape.n_layers, 'weight_bytes': W}
    # ---- prefill: 64-row steps vs one chain of 8 blocks
    for name, fn in (('prefill_64row_steps', lambda: eng.prefill(prompts[0])), ('prefill_chain_8_blocks', lambda: eng.mprefill(0, prompts[0]))):
        eng.reset(); fn(); eng.reset()
        torch.cuda.synchronize(); t0 = time.time(); fn(); torch.cuda.synchronize()
        out[name + '_ms'] = round(1e3 * (time.time() - t0), 3)
    # ---- verify steps with B blocks
    for B in (1, 2, 4, 8):
        eng.reset()
        eng.mprefill_many({b: prompts[b] for b in range(B)})
        def blocks():
            bl = []
            for b in range(B):
                _, rows = random_tree(rs, 64)
                bl.append((b, rs.randint(3, shape.vocab, size=64).astype(np.int32), rows, 0, 1))
            return bl
        for _ in range(3):
            eng.mstep(blocks())
        bls = [blocks() for _ in range(args.steps)]
        torch.cuda.synchronize(); t0 = time.time()
        for bl in bls:
            eng.mstep(bl)
        torch.cuda.synchronize()
        ms = 1e3 * (time.time() - t0) / args.steps
        flops = 2.0 * shape.n_params_no_embed() * 64 * B
        out[f'mstep_B{B}'] = {'ms': round(ms, 3), 'rows': 64 * B, 'weights_GBps': round(W / ms / 1e6, 1), 'TFLOPs': round(flops / ms / 1e9, 1),
                              'mfma_frac_of_2500': round(flops / ms / 1e9 / 2500.0, 4), 'hbm_frac_of_8000': round(W / ms / 1e6 / 8000.0, 4)}
        print(f'B={B}: {out[f"mstep_B{B}"]}', file=sys.stderr, flush=True)
    # the 64-row single-sequence step for reference
    eng.reset(); eng.prefill(prompts[0])
    _, rows = random_tree(rs, 64)
    ids = rs.randint(3, shape.vocab, size=64).astype(np.int32)
    for _ in range(3):
        eng.step(ids, rows, mode=2)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(args.steps):
        eng.step(ids, rows, mode=2)
    torch.cuda.synchronize()
    out['step64_ms'] = round(1e3 * (time.time() - t0) / args.steps, 3)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
