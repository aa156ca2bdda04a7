import os; os.environ.setdefault('LA_LAB_BUILD', '1')      # A/B script: the lab build (kernel-lab knobs, phase stamps) is the process library
# -*- coding: utf-8 -*-
"""Same-box, same-process A/B of kernel-lab knob settings on the Llama-2-7B single-sequence verify step (round 6).

Every setting is a ';'-separated list of la_lab_set "key=value" pairs applied on top of the library defaults; the settings are
visited round-robin `--reps` times (alternating order, so box drift hits every arm alike), each visit = prefill of the same prompt,
warm-up steps (the step graph is captured again under the new knobs), then `--steps` timed verify steps of the fixed T64/B8 tree
through the captured graph.  Emitted tokens and the logits of the last step are compared bitwise with the first setting's.

    python scripts/gpu_r6_knob_ab.py --settings "base:;o64:26=64;o128:26=128" [--layers 32] [--steps 64] [--reps 3]
"""
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


def parse_settings(spec):
    out = []
    for item in spec.split('|'):
        name, _, kvs = item.partition(':')
        pairs = []
        for kv in kvs.split(';'):
            if kv.strip():
                k, v = kv.split('=')
                pairs.append((int(k), int(v)))
        out.append((name.strip(), pairs))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--layers', type=int, default=32)
    ap.add_argument('--steps', type=int, default=64)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--prompt-len', type=int, default=640)
    ap.add_argument('--model', default='7b', choices=['7b', '13b'])
    ap.add_argument('--out', default='gpurun_out/r6_knob_ab.json')
    ap.add_argument('--settings', default='base:|o64:26=64')
    args = ap.parse_args()
    torch.cuda.set_device(0)
    shape = LlamaShape.llama2_7b() if args.model == '7b' else LlamaShape.llama2_13b()
    shape.n_layers = args.layers
    sd = random_weights(shape, seed=0, device='cuda:0', decisive=True)
    eng = LlamaVerifyEngine(shape, sd, max_length=max(2048, args.prompt_len + 512), consume_state_dict=True)
    rs = np.random.RandomState(0)
    prompt = rs.randint(3, shape.vocab, size=args.prompt_len).tolist()
    _, _, rows = fixed_t64b8_tree()
    ids = rs.randint(3, shape.vocab, size=64).astype(np.int32)
    settings = parse_settings(args.settings)
    keys = sorted({k for _, pairs in settings for k, _ in pairs})
    defaults = {k: lib.la_lab_get(k) for k in keys}
    times = {name: [] for name, _ in settings}
    ident = {name: True for name, _ in settings}
    tok_eq = {name: True for name, _ in settings}
    rel = {name: 0.0 for name, _ in settings}
    base = None
    for rep in range(args.reps):
        order = settings if rep % 2 == 0 else settings[::-1]
        for name, pairs in order:
            for k in keys:
                check(lib.la_lab_set(k, defaults[k]), 'lab_set')
            for k, v in pairs:
                check(lib.la_lab_set(k, v), 'lab_set')
            eng.reset()
            tok = eng.prefill(prompt, fast=False)
            ids[0] = tok
            toks = []
            for _ in range(6):
                toks.append(eng.step(ids, rows)[0])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                toks.append(eng.step(ids, rows)[0])
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / args.steps * 1e3
            logits = eng.logits().clone()
            if base is None:
                base = (toks, logits)
            else:
                if not (toks == base[0] and torch.equal(logits, base[1])):
                    ident[name] = False
                if toks != base[0]:
                    tok_eq[name] = False
                rel[name] = max(rel[name], float((logits.float() - base[1].float()).abs().max() / base[1].float().abs().max()))
            times[name].append(round(ms, 4))
            print(json.dumps({'rep': rep, 'setting': name, 'ms_per_step': round(ms, 4)}), flush=True)
    for k in keys:
        check(lib.la_lab_set(k, defaults[k]), 'lab_set')
    ref = min(times[settings[0][0]])
    summary = []
    for name, pairs in settings:
        t = times[name]
        summary.append({'setting': name, 'knobs': {str(k): v for k, v in pairs}, 'ms_min': min(t), 'ms_median': float(np.median(t)), 'ms_all': t,
                        'vs_first_min_pct': round((min(t) / ref - 1) * 100, 2), 'bitwise_identical_to_first': ident[name], 'tokens_equal_first': tok_eq[name],
                        'max_logit_rel_diff_vs_first': round(rel[name], 6)})
    out = {'model': args.model, 'layers': args.layers, 'steps': args.steps, 'reps': args.reps, 'prompt_len': args.prompt_len, 'summary': summary}
    os.makedirs(os.path.dirname(args.out) or '.', exist_ok=True)
    with open(args.out, 'w') as f:
        json.dump(out, f, indent=1)
    for r in summary:
        print('SUMMARY', json.dumps(r), flush=True)


if __name__ == '__main__':
    main()
