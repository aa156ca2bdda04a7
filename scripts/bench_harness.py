# -*- coding: utf-8 -*-
"""The reference's own benchmark METHODOLOGY (benchmarks/benchmark.py perf_check: tokens/s = generated tokens / wall time of
generate() INCLUDING prefill; trie warmed with the model's own greedy answers; cells (decoding_length, branch_length)) on
one MI355X with the Llama-2-7B-shaped synthetic model of bench.py.  Prompts: phrase-bank text (no datasets here), so the
numbers are not comparable to the README's GSM8K / Dolly rows — they show the lookahead-on / lookahead-off ratio this
implementation reaches under the reference's procedure.

    python scripts/bench_harness.py [--queries 16] [--prompt-len 128] [--new 256]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import phrase_prompt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--queries', type=int, default=16)
    ap.add_argument('--prompt-len', type=int, default=128)
    ap.add_argument('--new', type=int, default=256)
    args = ap.parse_args()
    from painlessinferenceacceleration_amd.benchmark import Benchmark
    from painlessinferenceacceleration_amd.llama_engine import LlamaShape
    from painlessinferenceacceleration_amd.modeling_llama import LlamaForCausalLM
    shape = LlamaShape.llama2_7b()
    model = LlamaForCausalLM.random_init(shape, seed=0, max_length=args.prompt_len + args.new + 80, eos_token_id=None,
                                         decisive=True)
    queries = [phrase_prompt(7000 + i, args.prompt_len, shape.vocab) for i in range(args.queries)]
    b = Benchmark(model=model, eos=None)
    t0 = time.time()
    answers = b.save_answers(queries, max_new_tokens=args.new)            # plain greedy = the warm-up corpus AND the baseline
    t_plain = time.time() - t0
    plain_speed = sum(len(a) for a in answers) / t_plain
    print(f'plain greedy (use_lookahead=False): {plain_speed:.1f} token/s over {len(queries)} queries', flush=True)
    res = b.perf_check(queries, answers=answers, warmup_ids=answers, max_new_tokens=args.new, sizes=(16, 32, 64), lens=(4, 8, 12))
    best = max(res.items(), key=lambda kv: kv[1])
    print(json.dumps({'workload': f'Llama-2-7B shape (synthetic decisive weights), {len(queries)} phrase-bank prompts x '
                                  f'{args.prompt_len} tokens, {args.new} new tokens, trie warmed with the greedy answers '
                                  f'(benchmarks/benchmark.py methodology, prefill included)',
                      'plain_greedy_tokens_per_sec': round(plain_speed, 1),
                      'lookahead_tokens_per_sec': {f'{k[0]}/{k[1]}': round(v, 1) for k, v in res.items()},
                      'best_cell': {'decoding_length': best[0][0], 'branch_length': best[0][1], 'tokens_per_sec': round(best[1], 1),
                                    'speedup_vs_plain': round(best[1] / plain_speed, 2)}}))


if __name__ == '__main__':
    main()
