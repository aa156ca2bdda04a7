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
import cProfile
import io
import json
import pstats
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
    def load_prompts(self, prompt_dir=None, warmup_prompt_dir=None, max_length=1024):
        """benchmark.py:79-100: jsonl files with one {"prompt": ..., "answer": ..., "ids": ...} object per line.  `prompt` is text
        (needs a tokenizer) or, in this repo's synthetic corpora, a token-id list; prompts longer than max_length (characters for
        text as in the reference's first test, tokens otherwise) are dropped.  Fills .prompts / .answers and, from
        warmup_prompt_dir, .warmup_prompts / .warmup_answers / .warmup_ids (the ids of the model's own answers: what warm_up() puts)."""
        def fits(x):
            if isinstance(x, str):
                return len(x) <= max_length or (self.tokenizer is not None and len(self.tokenizer.encode(x)) <= max_length)
            return len(x) <= max_length

        def read(path):
            prompts, answers, ids = [], [], []
            with open(path, 'r') as f:
                for line in f:
                    if not line.strip():
                        continue
                    d = json.loads(line)
                    prompts.append(d['prompt'])
                    answers.append(d.get('answer', None))
                    ids.append(d.get('ids', None))
            return prompts, answers, ids

        prompts, answers, _ = read(prompt_dir)
        self.prompts = [x for x in prompts if fits(x)]
        self.answers = answers                               # (unfiltered, as in the reference, :88)
        if warmup_prompt_dir is not None:
            prompts, answers, ids = read(warmup_prompt_dir)
            self.warmup_prompts = [x for x in prompts if fits(x)]
            self.warmup_answers = answers
            self.warmup_ids = ids

    def save_prompts(self, path, prompts, answers=None, ids=None):
        """Write a jsonl corpus load_prompts() reads (the counterpart of the reference's save_answers file format, :57-77)."""
        with open(path, 'w') as f:
            for i, p in enumerate(prompts):
                d = {'prompt': p if isinstance(p, str) else [int(t) for t in p]}
                if answers is not None:
                    d['answer'] = answers[i] if isinstance(answers[i], str) or answers[i] is None else [int(t) for t in answers[i]]
                if ids is not None:
                    d['ids'] = None if ids[i] is None else [int(t) for t in ids[i]]
                f.write(json.dumps(d) + '\n')

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

    # ---------------------------------------------------------------------------------------------- batch_chat
    def batch_chat(self, qs, max_new_tokens=256, decoding_length=64, branch_length=8, decoding_mode='hier', debug_lookahead=False,
                   erase=True, batch_size=1, max_query_length=2, verbose=True):
        """benchmark.py:188-241: every batch of queries twice — lookahead off, then on — on a trie that learns as it goes (erase =
        fresh trie first); per batch the reference's line (input / output sizes, edl/dl/pt/gt, time, speed, speedup), at the end
        `speed:off->on speedup`.  -> {'speed_off', 'speed_on', 'speedup', 'identical': outputs of the two legs equal on every batch}."""
        total_out, total_t = [0, 0], [0.0, 0.0]
        if erase:
            self.model.lookahead_cache.fresh()
        identical = True
        for i in range(len(qs) // batch_size):
            query = qs[i * batch_size:(i + 1) * batch_size]
            speeds, outs = [], []
            for j, use_lookahead in enumerate([False, True]):
                ts = time.time()
                in_texts, in_ids, out_ids, out_texts, kw = self.chat(query, max_new_tokens=max_new_tokens, use_lookahead=use_lookahead,
                                                                     decoding_length=decoding_length, branch_length=branch_length,
                                                                     decoding_mode=decoding_mode, debug_lookahead=debug_lookahead,
                                                                     max_query_length=max_query_length)
                t = time.time() - ts
                in_char, in_token = sum(len(x) for x in in_texts), sum(len(x) for x in in_ids)
                out_char, out_token = sum(len(x) for x in out_texts), sum(len(x) for x in out_ids)
                speeds.append(out_token / max(t, 1e-9))
                outs.append(out_ids)
                total_out[j] += out_token
                total_t[j] += t
                bs = len(query)
                dls, edls, fts = kw.get('dls', []), kw.get('edls', []), kw.get('fts', [0])
                dl = sum(dls[bs:]) / len(dls[bs:]) if len(dls) > bs else 0.0
                edl = sum(edls[bs:]) / len(edls[bs:]) if len(edls) > bs else 0.0
                pt, gts = fts[0] if fts else 0.0, fts[1:]
                gt = sum(gts) / max(len(gts), 1)
                if verbose:
                    print(f'1/{bs} Robot:{out_texts[0]}')
                    prefix = 'lookahead:' + ('On ' if use_lookahead else 'Off')
                    speedup = speeds[-1] / speeds[0] if use_lookahead else 0.0
                    print(f'{prefix} mode:{decoding_mode} idx:{i} input:{in_char:.1f}/{in_token:.1f} output:{out_char:.1f}/{out_token:.1f} '
                          f'edl:{edl:.3f}/{dl:.3f}/{pt:.3f}/{gt:.3f} time:{t:.3f} speed:{speeds[-1]:.1f} speedup:{speedup:.3f}\n')
            identical = identical and outs[0] == outs[1]
        org, opt = total_out[0] / max(total_t[0], 1e-9), total_out[1] / max(total_t[1], 1e-9)
        print(f'speed:{org:.2f}->{opt:.2f} speedup:{opt / max(org, 1e-9):.3f}')
        return {'speed_off': org, 'speed_on': opt, 'speedup': opt / max(org, 1e-9), 'identical': identical}

    # ---------------------------------------------------------------------------------------------- profile hooks
    def naive_profile(self, qs, use_lookahead=False, count=64, sortby=0, **chat_kw):
        """benchmark.py:397-409: cProfile over chat() of every query; prints (and returns) the top `count` rows by time / cumulative."""
        pr = cProfile.Profile()
        pr.enable()
        for q in qs:
            self.chat(q, use_lookahead=use_lookahead, **chat_kw)
        pr.disable()
        buf = io.StringIO()
        pstats.Stats(pr, stream=buf).sort_stats(pstats.SortKey.TIME if sortby == 0 else pstats.SortKey.CUMULATIVE).print_stats(count)
        print(buf.getvalue())
        return buf.getvalue()

    def naive_profile_trie(self, lookahead_cache, warmup_ids, input_ids, output_ids, max_node_rate=16, decoding_length=64,
                           branch_length=24, edl=8, count=64, sortby=0):
        """benchmark.py:411-426: cProfile over the trie-only loop (perf_check_trie)."""
        pr = cProfile.Profile()
        pr.enable()
        self.perf_check_trie(lookahead_cache, warmup_ids, input_ids, output_ids, max_node_rate=max_node_rate,
                             decoding_length=decoding_length, branch_length=branch_length, edl=edl, verbose=False)
        pr.disable()
        buf = io.StringIO()
        pstats.Stats(pr, stream=buf).sort_stats(pstats.SortKey.TIME if sortby == 0 else pstats.SortKey.CUMULATIVE).print_stats(count)
        print(buf.getvalue())
        return buf.getvalue()

    def torch_profile(self, use_lookahead=False, trace_dir='./prof', prompts=None, **chat_kw):
        """benchmark.py:428-441: torch.profiler (wait 1 / warmup 1 / active 3) over chat() of the loaded prompts, traces for
        tensorboard in trace_dir.  The verify step itself is a captured hipGraph of hand-written kernels — per-kernel numbers come
        from rocprofv3 (scripts/gpu_profile.sh); this hook shows the host side of the loop, as the reference's does."""
        prof = torch.profiler.profile(schedule=torch.profiler.schedule(wait=1, warmup=1, active=3, repeat=1),
                                      on_trace_ready=torch.profiler.tensorboard_trace_handler(trace_dir),
                                      record_shapes=True, with_stack=True)
        prof.start()
        for p in (self.prompts if prompts is None else prompts):
            prof.step()
            self.chat(p, use_lookahead=use_lookahead, **chat_kw)
        prof.stop()
        return prof

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
