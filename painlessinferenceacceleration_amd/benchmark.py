# -*- coding: utf-8 -*-
"""Benchmark harness with the methodology of lookahead/benchmarks/benchmark.py: the same knobs, the same
metric definitions and the same log lines, driving the MI355X model wrappers.

  speed  = generated tokens / wall time of chat() INCLUDING prefill               (benchmark.py:277-328)
  edl/dl = mean accepted / draft length over decode steps (dls[bs:], edls[bs:])   (:302-316)
  pt/gt  = mean prefill / decode-step wall time (fts[0] / fts[1:])
  warm_up: cache.put([eop] + ids, branch_length + 1, mode='output', idx=-1)      (:159-169)
  perf_check resets the trie per (decoding_length, branch_length) cell and sets max_output_node = max_node_rate *
  decoding_length, max_node = 2 * that                                           (:270-274)

Prompts are token-id lists (no tokenizer or dataset ships with this repo); pass a `tokenizer` with encode/decode to use
text.  The Rouge-L `acc` column needs the optional rouge_score package; without it the column reports exact token-match
rate against the given answers.
"""
import time

import torch


class Benchmark(object):
    def __init__(self, model=None, tokenizer=None, log_dir=None, eos=None, eop=None, device='cuda:0'):
        self.model = model
        self.tokenizer = tokenizer
        self.eos = eos
        self.eop = eop
        self.device = device
        self.logger = open(log_dir, 'a+') if log_dir is not None else None
        self.prompts, self.answers, self.ids = [], [], []
        self.warmup_prompts, self.warmup_answers, self.warmup_ids = [], [], []

    # ------------------------------------------------------------------------------------------------ data
    def tokenize(self, prompt, max_length=256):
        """-> list of token-id lists (benchmark.py:102-113); token-id inputs pass through, truncated."""
        if isinstance(prompt, (list, tuple)) and len(prompt) > 0 and isinstance(prompt[0], int):
            prompt = [prompt]
        out = []
        for p in prompt:
            ids = list(p) if not isinstance(p, str) else self.tokenizer.encode(p)
            out.append(ids[:max_length])
        return out

    def to_words(self, token_ids):
        return self.tokenizer.decode(token_ids) if self.tokenizer is not None else ' '.join(str(t) for t in token_ids)

    # ------------------------------------------------------------------------------------------------ chat
    def chat(self, prompt, max_length=2048, max_new_tokens=256, use_lookahead=False, decoding_length=64, branch_length=8,
             decoding_mode='hier', debug_lookahead=False, max_query_length=2):
        """One generate() call for one prompt or a batch (benchmark.py:115-157).
        -> (prompt, input_id_list, output_id_list, output_texts, kwargs)"""
        ids = self.tokenize(prompt, max_length=max_length)
        bs = len(ids)
        P = max(len(x) for x in ids)
        pad = getattr(self.model.generation_config, 'pad_token_id', 0) or 0
        input_ids = torch.full((bs, P), pad, dtype=torch.long)
        attention_mask = torch.zeros((bs, P), dtype=torch.long)
        for b, x in enumerate(ids):                      # left padding, as the reference tokenizer is configured
            input_ids[b, P - len(x):] = torch.tensor(x)
            attention_mask[b, P - len(x):] = 1
        decoding_kwargs = {'use_lookahead': use_lookahead, 'debug_lookahead': debug_lookahead,
                           'decoding_mode': decoding_mode, 'decoding_length': decoding_length,
                           'branch_length': branch_length, 'max_query_length': max_query_length,
                           'stop_words': {}, 'tokenizer': self.tokenizer}
        outputs = self.model.generate(input_ids=input_ids, attention_mask=attention_mask if bs > 1 else None,
                                      max_new_tokens=max_new_tokens, eos_token_id=self.eos, pad_token_id=pad,
                                      return_dict_in_generate=True, decoding_kwargs=decoding_kwargs)
        seqs = outputs.sequences[:, P:].tolist()
        output_id_list = []
        for row in seqs:
            if self.eos is not None and self.eos in row:
                row = row[:row.index(self.eos)]
            while bs > 1 and row and row[-1] == pad:
                row = row[:-1]
            output_id_list.append(row)
        output_texts = [self.to_words(x) for x in output_id_list]
        return prompt, ids, output_id_list, output_texts, getattr(outputs, 'kwargs', {}) or {}

    def warm_up(self, ids, branch_length=8, eop=None):
        cache = self.model.lookahead_cache
        ts = time.time()
        for i, ids_ in enumerate(ids):
            if ids_ is None:
                continue
            cache.put([eop] + list(ids_) if eop else list(ids_), branch_length=branch_length + 1, mode='output', idx=-1)
            if (i + 1) % 1000 == 0:
                print(f'warmup:{i + 1}, elapse:{round(time.time() - ts, 1)}s')

    def save_answers(self, queries, max_new_tokens=256, batch_size=1):
        """Plain-greedy answers of the model itself: the warm-up corpus of the README tables (benchmark.py:57-77)."""
        out = []
        for k in range(0, len(queries), batch_size):
            _, _, output_id_list, _, _ = self.chat(queries[k:k + batch_size], max_new_tokens=max_new_tokens,
                                                   use_lookahead=False)
            out.extend(output_id_list)
        return out

    # ---------------------------------------------------------------------------------------------- perf_check
    def _score(self, output_ids, output_text, answer):
        try:
            from rouge_score import rouge_scorer
            scorer = rouge_scorer.RougeScorer(['rougeL'], use_stemmer=True)
            return scorer.score(prediction=output_text, target=answer if isinstance(answer, str) else self.to_words(answer))['rougeL'].fmeasure
        except ImportError:
            ans = list(answer) if not isinstance(answer, str) else self.tokenizer.encode(answer)
            n = max(len(ans), len(output_ids), 1)
            return sum(1 for a, b in zip(output_ids, ans) if a == b) / float(n)

    def _run_cell(self, queries, answers, batch_size, gen_kw):
        """One (decoding_length, branch_length) cell: every batch of queries through chat(); returns the accumulated counters."""
        acc = {'tok_in': 0, 'tok_out': 0, 'draft': [], 'accepted': [], 'prefill_s': [], 'step_s': [], 'match': []}
        started = time.time()
        for first in range(0, len(queries) - batch_size + 1, batch_size):
            group = queries[first:first + batch_size]
            _, in_ids, out_ids, out_texts, info = self.chat(group, **gen_kw)
            acc['tok_in'] += sum(map(len, in_ids))
            acc['tok_out'] += sum(map(len, out_ids))
            n = len(group)
            if answers is not None:
                acc['match'] += [self._score(out_ids[i], out_texts[i], answers[first + i]) for i in range(n)]
            # the first `n` entries of dls / edls belong to the prefill step of each sample (benchmark.py:302-305)
            acc['draft'] += list(info.get('dls', []))[n:]
            acc['accepted'] += list(info.get('edls', []))[n:]
            step_times = list(info.get('fts', [0]))
            acc['prefill_s'].append(step_times[0])
            acc['step_s'] += step_times[1:]
        acc['wall_s'] = time.time() - started
        return acc

    def perf_check(self, queries, answers=None, warmup_ids=None, max_new_tokens=256, sizes=(32, 64), lens=(4, 8, 12),
                   decoding_mode='hier', batch_size=1, max_node_rate=16, max_query_length=2):
        """benchmarks/benchmark.py:243-351: one log line per (decoding_length, branch_length) cell; -> {cell: tokens/s}."""
        n_warm = 0 if warmup_ids is None else len(warmup_ids)
        print(f'\nmode:{decoding_mode} bs:{batch_size} queries:{len(queries)} warmup:{n_warm} sizes:{sizes} lens:{lens}')
        if batch_size > 1:                               # batches of similar length (:249-250)
            order = sorted(range(len(queries)), key=lambda i: len(queries[i]))
            queries = [queries[i] for i in order]
            answers = None if answers is None else [answers[i] for i in order]
        trie = self.model.lookahead_cache
        mean = lambda xs: sum(xs) / max(len(xs), 1)      # noqa: E731
        result = {}
        for dl in sizes:
            for bl in lens:
                if dl < bl * batch_size:
                    continue
                lookahead_on = dl > 1 and bl > 0
                if lookahead_on:                         # fresh trie per cell, limits scaled with the draft size (:268-274)
                    trie.fresh()
                    trie.max_output_node = max_node_rate * dl
                    trie.max_node = 2 * max_node_rate * dl
                    if warmup_ids is not None:
                        self.warm_up(warmup_ids, branch_length=bl, eop=self.eop)
                if torch.cuda.is_available():
                    torch.cuda.reset_peak_memory_stats(device=None)
                acc = self._run_cell(queries, answers, batch_size,
                                     dict(max_new_tokens=max_new_tokens, use_lookahead=lookahead_on, decoding_length=dl,
                                          branch_length=bl, decoding_mode=decoding_mode, max_query_length=max_query_length))
                per_query = max(len(queries), 1)
                t = acc['wall_s'] / per_query
                tokens_out = acc['tok_out'] / per_query
                speed = tokens_out / max(t, 1e-9)
                result[(dl, bl)] = speed
                mem = torch.cuda.max_memory_allocated() / 1e9 if torch.cuda.is_available() else 0.0
                log_str = (f'mode:{decoding_mode} bs:{batch_size} decoding_length:{dl} branch_length:{bl} '
                           f'query:{len(queries)} warmup:{n_warm} input:{acc["tok_in"] / per_query:.1f} output:{tokens_out:.1f} '
                           f'edl:{mean(acc["accepted"]):.3f}/{mean(acc["draft"]):.3f}/{mean(acc["prefill_s"]):.3f}/'
                           f'{mean(acc["step_s"]):.3f} time:{t:.3f} speed:{speed:.1f} mem:{mem:.3f} acc:{mean(acc["match"]):.4f}')
                print(log_str)
                if self.logger is not None:
                    self.logger.write(log_str + '\n')
                    self.logger.flush()
        return result

    # ------------------------------------------------------------------------------------------- perf_check_trie
    @staticmethod
    def perf_check_trie(lookahead_cache, warmup_ids, input_ids, output_ids, max_node_rate=16, decoding_length=64,
                        branch_length=24, edl=8, verbose=True):
        """Trie-only timing loop of benchmark.py:353-395: per sample one `put` of the prompt, one `bat_get` every `edl`
        output tokens, then the output streamed in with `stream_put`.  Works on ANY object with the LookaheadCache surface,
        so the native trie and the reference's Python trie can be timed side by side.  -> dict (seconds / counts)."""
        trie = lookahead_cache
        trie.max_output_node = decoding_length * max_node_rate
        trie.fresh()
        for seq in warmup_ids:
            trie.put(list(seq), branch_length=branch_length + 1, mode='output', idx=0, final=False)
        clock = {'put': 0.0, 'get': 0.0}
        n_put = n_get = 0

        def timed(kind, fn, *a, **kw):
            t0 = time.time()
            fn(*a, **kw)
            clock[kind] += time.time() - t0

        for prompt, reply in zip(input_ids, output_ids):
            prompt, reply = list(prompt), list(reply)
            n_put += len(prompt) + len(reply)
            timed('put', trie.put, prompt, branch_length=branch_length + 1, mode='input', idx=0, final=False)
            starts = range(0, len(reply) - 1, edl)
            for j in starts:
                n_get += 1
                timed('get', trie.bat_get, [reply[j:j + 2]], decoding_length=decoding_length, branch_length=branch_length,
                      decoding_cursors=[j], mode='mix', indices=[0], decoding_mode='hier')
            for j in starts:
                timed('put', trie.stream_put, reply[j:j + edl], branch_length=branch_length + 1, mode='output', idx=0,
                      final=False)
            timed('put', trie.stream_put, [], branch_length=branch_length + 1, mode='output', idx=0, final=True)
        count = len(input_ids)
        res = {'samples': count, 'put_tokens': n_put, 'put_s': clock['put'], 'put_us_per_token': 1e6 * clock['put'] / max(n_put, 1),
               'gets': n_get, 'get_s': clock['get'], 'get_ms_per_query': 1e3 * clock['get'] / max(n_get, 1)}
        if verbose:
            print(f'\nparam:{max_node_rate}/{decoding_length}/{branch_length} sample:{count} '
                  f'put:{n_put}/{clock["put"]:.2f}/{res["put_us_per_token"] / 1e3:.2f}/{1e3 * clock["put"] / max(count, 1):.2f} '
                  f'get:{n_get}/{clock["get"]:.2f}/{res["get_ms_per_query"]:.2f}/{1e3 * clock["get"] / max(count, 1):.2f}\n')
        return res

    def grid_search(self, queries, warmup_ids=None, sizes=(16, 32, 64), lens=(4, 8, 12, 16), **kw):
        """benchmark.py:455-468: perf_check over the (decoding_length, branch_length) grid; -> best cell."""
        res = self.perf_check(queries, warmup_ids=warmup_ids, sizes=sizes, lens=lens, **kw)
        best = max(res.items(), key=lambda kv: kv[1]) if res else None
        print('best (decoding_length, branch_length):', best)
        return res
