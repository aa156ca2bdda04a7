import os; os.environ.setdefault('LA_LAB_BUILD', '1')      # A/B script: the lab build (kernel-lab knobs, phase stamps) is the process library
# -*- coding: utf-8 -*-
"""A/B of the idle-window weight prefetch (la_debug_set keys 7 / 8 / 9) on one MI355X: the Llama-2-7B verify step (64-row T64/B8
tree, 512-token context) through the captured graph for every (KiB per workgroup, delay, tail KiB) setting — ms per step, HIP-event time
per kernel class of the eager step, and a bitwise comparison of logits / emitted tokens with the prefetch switched off.

    python scripts/gpu_pf_ab.py [--layers N] [--steps K] [--out gpurun_out/pf_ab.json]

Writes the JSON record and, next to it, `best_kib` / `best_delay` / `best_tail` (one integer each) for follow-up commands."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import fixed_t64b8_tree                                        # noqa: E402
from painlessinferenceacceleration_amd._lib import check, lib            # noqa: E402
from painlessinferenceacceleration_amd.llama_engine import LlamaShape, LlamaVerifyEngine, random_weights   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--layers', type=int, default=32)
    ap.add_argument('--steps', type=int, default=48)
    ap.add_argument('--out', default='gpurun_out/pf_ab.json')
    ap.add_argument('--prompt-len', type=int, default=512)
    ap.add_argument('--settings', default='0:0:0,16:0:0,32:0:0,64:0:0,128:0:0,64:2:0,0:0:32,0:0:64,64:0:32,64:0:64,128:0:64,0:0:0',
                    help='comma list of kib:delay:tail_kib[:attn_staged] (la_debug_set keys 7 / 8 / 9 / 10)')
    args = ap.parse_args()
    torch.cuda.set_device(0)
    shape = LlamaShape.llama2_7b()
    shape.n_layers = args.layers
    sd = random_weights(shape, seed=0, device='cuda:0', decisive=True)
    eng = LlamaVerifyEngine(shape, sd, max_length=max(2048, args.prompt_len + 512), consume_state_dict=True)
    rs = np.random.RandomState(0)
    prompt = rs.randint(3, shape.vocab, size=args.prompt_len).tolist()
    _, _, rows = fixed_t64b8_tree()
    ids = rs.randint(3, shape.vocab, size=64).astype(np.int32)
    results, base = [], None
    for setting in args.settings.split(','):
        kib, dly, tail, staged = ([int(x) for x in setting.split(':')] + [0])[:4]
        check(lib.la_lab_set(7, kib), 'debug_set')
        check(lib.la_lab_set(8, dly), 'debug_set')
        check(lib.la_lab_set(9, tail), 'debug_set')
        check(lib.la_lab_set(10, staged), 'debug_set')
        eng.reset()
        tok = eng.prefill(prompt, fast=False)
        ids[0] = tok
        toks = []
        for _ in range(6):                                               # warm-up (captures the graph with this setting)
            toks.append(eng.step(ids, rows)[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            toks.append(eng.step(ids, rows)[0])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        logits = eng.logits().clone()
        prof = eng.profile(ids, rows, iters=3)
        rec = {'kib': kib, 'delay': dly, 'tail_kib': tail, 'attn_staged': staged, 'ms_per_step': round(ms, 4), 'ms_by_class_events': {k: round(v, 4) for k, v in prof['ms'].items()},
               'ms_eager_step': round(prof['ms_step'], 4)}
        if base is None:
            base = (toks, logits)
            rec['identical_to_off'] = True
        else:
            rec['identical_to_off'] = bool(toks == base[0] and torch.equal(logits, base[1]))
        print(json.dumps(rec), flush=True)
        results.append(rec)
    for k in (7, 8, 9, 10):
        check(lib.la_lab_set(k, 0), 'debug_set')
    ok = [r for r in results if r['identical_to_off']]
    best = min(ok, key=lambda r: r['ms_per_step'])
    off = [r['ms_per_step'] for r in results if r['kib'] == 0 and r['tail_kib'] == 0]
    # only adopt a setting that beats BOTH prefetch-off runs (first and last of the sweep) by more than the run-to-run noise
    if (best['kib'] or best['tail_kib']) and best['ms_per_step'] > min(off) * 0.99:
        best = {'kib': 0, 'delay': 0, 'tail_kib': 0, 'ms_per_step': min(off)}
    out = {'layers': args.layers, 'steps': args.steps, 'results': results,
           'best': {k: best[k] for k in ('kib', 'delay', 'tail_kib', 'ms_per_step')},
           'all_identical': all(r['identical_to_off'] for r in results)}
    os.makedirs(os.path.dirname(args.out) or '.', exist_ok=True)
    with open(args.out, 'w') as f:
        json.dump(out, f, indent=1)
    d = os.path.dirname(args.out) or '.'
    open(os.path.join(d, 'best_kib'), 'w').write(str(best['kib']))
    open(os.path.join(d, 'best_delay'), 'w').write(str(best['delay']))
    open(os.path.join(d, 'best_tail'), 'w').write(str(best['tail_kib']))
    print('BEST', json.dumps(out['best']), 'all_identical', out['all_identical'], flush=True)


if __name__ == '__main__':
    main()
