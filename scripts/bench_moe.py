# -*- coding: utf-8 -*-
"""Secondary measurement (BASELINE config 5): Mixtral-8x7B-shaped sparse-MoE verify step on ONE MI355X (93 GB of bf16
weights resident in HBM), bs=1 (64-row tree) and cursor-batch bs=4 (4 x 16 rows in one block).  Synthetic random
weights; the trie is not involved (fixed random trees): this times the verify step and checks that the batch block
reproduces the bs=1 argmax rows.

    python scripts/bench_moe.py [--layers N] [--steps K]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--layers', type=int, default=32)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--ctx', type=int, default=512)
    args = ap.parse_args()
    from painlessinferenceacceleration_amd.llama_engine import LlamaShape, LlamaVerifyEngine, random_weights
    from tests.gpu_utils import random_tree
    shape = LlamaShape.mixtral_8x7b()
    shape.n_layers = args.layers
    t0 = time.time()
    sd = random_weights(shape, seed=0, device='cuda:0')
    eng = LlamaVerifyEngine(shape, sd, max_length=args.ctx + 64 * (args.steps + 4), n_slots=4, consume_state_dict=True)
    del sd
    torch.cuda.synchronize()
    t_init = time.time() - t0
    wbytes = 2 * shape.n_params_no_embed()
    print(f'[setup] {args.layers} layers, {wbytes / 1e9:.1f} GB of weights packed in {t_init:.0f}s; '
          f'HBM in use {torch.cuda.memory_allocated() / 1e9:.1f} GB', file=sys.stderr, flush=True)
    rs = np.random.RandomState(0)
    out = {'workload': f'Mixtral-8x7B shape ({args.layers} layers), bf16, one MI355X', 'weights_GB': round(wbytes / 1e9, 2)}
    # bs=1: 64-row trees
    prompt = rs.randint(3, shape.vocab, size=args.ctx).tolist()
    eng.prefill(prompt)
    _, rows = random_tree(rs, 64)
    ids = rs.randint(3, shape.vocab, size=64).astype(np.int32)
    for _ in range(3):
        eng.step(ids, rows)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.steps):
        eng.step(ids, rows)
    torch.cuda.synchronize()
    ms1 = 1e3 * (time.time() - t0) / args.steps
    used = [int((eng.route_weights()[l, :64] != 0).any(0).sum()) for l in range(min(args.layers, 4))]
    out['bs1_T64'] = {'ms_per_step': round(ms1, 3), 'weight_stream_GBps': round(wbytes / ms1 / 1e6, 1),
                      'experts_hit_first_layers': used}
    # T=1 greedy steps: two experts per layer are read
    one = np.array([1], dtype=np.uint64)
    for _ in range(3):
        eng.step(ids[:1], one)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.steps):
        eng.step(ids[:1], one)
    torch.cuda.synchronize()
    msg = 1e3 * (time.time() - t0) / args.steps
    out['bs1_T1_greedy'] = {'ms_per_step': round(msg, 3)}
    # cursor batch: 4 sequences x 16 rows
    eng.reset_slot(-1)
    eng.bprefill_many({s: rs.randint(3, shape.vocab, size=args.ctx // 4).tolist() for s in range(4)})
    segs = []
    for s in range(4):
        _, r = random_tree(rs, 16)
        segs.append((s, rs.randint(3, shape.vocab, size=16).astype(np.int32), np.asarray(r, dtype=np.uint64), 0, 16))
    for _ in range(3):
        eng.bstep(segs)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.steps):
        eng.bstep(segs)
    torch.cuda.synchronize()
    ms4 = 1e3 * (time.time() - t0) / args.steps
    out['bs4_T16'] = {'ms_per_step': round(ms4, 3), 'weight_stream_GBps': round(wbytes / ms4 / 1e6, 1)}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
