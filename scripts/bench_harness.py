# -*- coding: utf-8 -*-
"""The reference's own benchmark METHODOLOGY (benchmarks/benchmark.py perf_check: tokens/s = generated tokens / wall time of
generate() INCLUDING prefill; trie warmed with the model's own greedy answers; cells (decoding_length, branch_length)) on
one MI355X with the Llama-2-7B-shaped synthetic model of bench.py.  Prompts: phrase-bank text (no datasets here), so the
numbers are not comparable to the README's GSM8K / Dolly rows — they show the lookahead-on / lookahead-off ratio this
implementation reaches under the reference's procedure.

    python scripts/bench_harness.py [--queries 8] [--prompt-len 128] [--new 256] [--table gpurun_out/harness_table.md]

Round 5: the grid is (decoding_length, branch_length) in {32, 64, 128} x {8, 12, 32} (128-token trees = two chained blocks of one
multi-block pass), the trie is warmed per cell with NOISY copies of the greedy answers (bench.py's recipe: with the synthetic
permutation LM the answers of other prompts never recur, so the reference's "answers of a warm-up set" would give empty drafts,
and exact copies would give all-accepted ones), Benchmark.batch_chat runs the off / on legs, and the cells are written as a
markdown table in the README's column format (model / dataset / GPU / framework / tokens/s / speedup + edl, dl).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import noisy_copies, phrase_prompt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--queries', type=int, default=8)
    ap.add_argument('--table', default='')
    ap.add_argument('--rho', type=float, default=0.3)
    ap.add_argument('--prompt-len', type=int, default=128)
    ap.add_argument('--new', type=int, default=256)
    args = ap.parse_args()
    from painlessinferenceacceleration_amd.benchmark import Benchmark
    from painlessinferenceacceleration_amd.llama_engine import LlamaShape
    from painlessinferenceacceleration_amd.modeling_llama import LlamaForCausalLM
    shape = LlamaShape.llama2_7b()
    model = LlamaForCausalLM.random_init(shape, seed=0, max_length=args.prompt_len + args.new + 140, eos_token_id=None,
                                         decisive=True, max_blocks=2)
    queries = [phrase_prompt(7000 + i, args.prompt_len, shape.vocab) for i in range(args.queries)]
    b = Benchmark(model=model, eos=None)
    t0 = time.time()
    answers = b.save_answers(queries, max_new_tokens=args.new)            # plain greedy = the warm-up corpus AND the baseline
    t_plain = time.time() - t0
    plain_speed = sum(len(a) for a in answers) / t_plain
    print(f'plain greedy (use_lookahead=False): {plain_speed:.1f} token/s over {len(queries)} queries', flush=True)
    warm = []
    for i, (p, a) in enumerate(zip(queries, answers)):
        warm += noisy_copies(p[-2:] + a, 12, args.rho, shape.vocab, seed=99 + i)
    import contextlib
    import io
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res = b.perf_check(queries, answers=answers, warmup_ids=warm, max_new_tokens=args.new, sizes=(32, 64, 128), lens=(8, 12, 32))
    log = buf.getvalue()
    print(log, flush=True)
    cells = {}
    for ln in log.splitlines():
        if ln.startswith('mode:hier bs:1 decoding_length:'):
            dl = int(ln.split('decoding_length:')[1].split()[0]); bl = int(ln.split('branch_length:')[1].split()[0])
            e = ln.split('edl:')[1].split()[0].split('/')
            cells[(dl, bl)] = {'edl': float(e[0]), 'dl': float(e[1]), 'prefill_s': float(e[2]), 'step_s': float(e[3]),
                               'speed': float(ln.split('speed:')[1].split()[0]), 'acc': float(ln.split('acc:')[1].split()[0])}
    best = max(res.items(), key=lambda kv: kv[1])
    model.lookahead_cache.fresh()
    b.warm_up(warm, branch_length=12)
    bc = b.batch_chat(queries, max_new_tokens=args.new, decoding_length=64, branch_length=12, erase=False, batch_size=1, verbose=False)
    summary = {'workload': f'Llama-2-7B shape (synthetic decisive weights), {len(queries)} phrase-bank prompts x '
                           f'{args.prompt_len} tokens, {args.new} new tokens, trie warmed per cell with 12 noisy copies (rho {args.rho}) of each '
                           f'greedy answer (benchmarks/benchmark.py methodology: tokens / wall time of generate(), prefill included)',
               'plain_greedy_tokens_per_sec': round(plain_speed, 1),
               'lookahead_tokens_per_sec': {f'{k[0]}/{k[1]}': round(v, 1) for k, v in res.items()},
               'answers_equal_greedy_in_every_cell': all(c['acc'] == 1.0 for c in cells.values()),
               'best_cell': {'decoding_length': best[0][0], 'branch_length': best[0][1], 'tokens_per_sec': round(best[1], 1),
                             'speedup_vs_plain': round(best[1] / plain_speed, 2)},
               'batch_chat_64_12': {k: (round(v, 2) if isinstance(v, float) else v) for k, v in bc.items()}}
    print(json.dumps(summary))
    if args.table:
        rows = ['| model | dataset | GPU | framework | decoding_length | branch_length | tokens/s | speedup | edl | dl | answers == greedy |',
                '|---|---|---|---|---|---|---|---|---|---|---|',
                f'| Llama-2-7B shape (synthetic) | phrase-bank x {len(queries)} | MI355X | plain greedy (same engine) | - | - | {plain_speed:.1f} | 1.00x | 1.000 | 1.0 | - |']
        for (dl, bl), c in sorted(cells.items()):
            rows.append(f'| Llama-2-7B shape (synthetic) | phrase-bank x {len(queries)} | MI355X | lookahead (this repo) | {dl} | {bl} | {c["speed"]:.1f} | '
                        f'{c["speed"] / plain_speed:.2f}x | {c["edl"]:.3f} | {c["dl"]:.1f} | {"yes" if c["acc"] == 1.0 else "NO"} |')
        with open(args.table, 'w') as f:
            f.write('Benchmark.perf_check on one MI355X (scripts/bench_harness.py; reference README table columns + edl / dl).  '
                    'tokens/s = generated tokens / wall time of generate() INCLUDING prefill and the host loop (benchmarks/benchmark.py:277-328).\n'
                    + summary['workload'] + '\n\n' + '\n'.join(rows) + '\n\n'
                    + f'Benchmark.batch_chat (64/12, trie learning as it goes): speed {bc["speed_off"]:.1f} -> {bc["speed_on"]:.1f} tokens/s, '
                      f'speedup {bc["speedup"]:.3f}, off/on outputs identical: {bc["identical"]}\n')


if __name__ == '__main__':
    main()
