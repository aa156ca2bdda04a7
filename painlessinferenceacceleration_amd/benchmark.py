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

    def perf_check(self, queries, answers=None, warmup_ids=None, max_new_tokens=256, sizes=(32, 64), lens=(4, 8, 12),
                   decoding_mode='hier', batch_size=1, max_node_rate=16, max_query_length=2):
        wc = len(warmup_ids) if warmup_ids is not None else 0
        print(f'\nmode:{decoding_mode} bs:{batch_size} queries:{len(queries)} warmup:{wc} sizes:{sizes} lens:{lens}')
        if batch_size > 1:
            order = sorted(range(len(queries)), key=lambda i: len(queries[i]))
            queries = [queries[i] for i in order]
            answers = [answers[i] for i in order] if answers is not None else None
        speeds, outputs = [], {}
        cache = self.model.lookahead_cache
        for decoding_length in sizes:
            for branch_length in lens:
                if decoding_length < branch_length * batch_size:
                    continue
                use_lookahead = decoding_length > 1 and branch_length > 0
                in_token = out_token = 0
                dls, edls, pts, gts, scores, times = [], [], [], [], [], []
                if use_lookahead:
                    cache.fresh()
                    cache.max_output_node = max_node_rate * decoding_length
                    cache.max_node = 2 * max_node_rate * decoding_length
                    if warmup_ids is not None:
                        self.warm_up(warmup_ids, branch_length=branch_length, eop=self.eop)
                if torch.cuda.is_available():
                    torch.cuda.reset_peak_memory_stats(device=None)
                ts = time.time()
                for k in range(len(queries) // batch_size):
                    qs_ = queries[k * batch_size:(k + 1) * batch_size]
                    ts_ = time.time()
                    _, input_id_list, output_id_list, output_texts, kwargs = self.chat(
                        qs_, max_new_tokens=max_new_tokens, use_lookahead=use_lookahead, decoding_length=decoding_length,
                        branch_length=branch_length, decoding_mode=decoding_mode, max_query_length=max_query_length)
                    times.append(time.time() - ts_)
                    in_token += sum(len(x) for x in input_id_list)
                    out_token += sum(len(x) for x in output_id_list)
                    bs = len(qs_)
                    if answers is not None:
                        for i in range(bs):
                            scores.append(self._score(output_id_list[i], output_texts[i], answers[k * batch_size + i]))
                    dls_, edls_ = kwargs.get('dls', []), kwargs.get('edls', [])
                    dls.extend(dls_[bs:] if len(dls_) > bs else [])
                    edls.extend(edls_[bs:] if len(edls_) > bs else [])
                    pts.append(kwargs.get('fts', [0])[0])
                    gts.extend(kwargs.get('fts', [0])[1:])
                n_repeat = max(len(queries), 1)
                t = (time.time() - ts) / n_repeat
                in_token /= n_repeat
                out_token /= n_repeat
                speed = out_token / max(t, 1e-9)
                speeds.append(speed)
                outputs[(decoding_length, branch_length)] = speed
                dl = sum(dls) / max(len(dls), 1)
                edl = sum(edls) / max(len(edls), 1)
                pt = sum(pts) / max(len(pts), 1)
                gt = sum(gts) / max(len(gts), 1)
                mem = torch.cuda.max_memory_allocated() / 1e9 if torch.cuda.is_available() else 0.0
                score = sum(scores) / max(len(scores), 1.0)
                log_str = (f'mode:{decoding_mode} bs:{batch_size} decoding_length:{decoding_length} '
                           f'branch_length:{branch_length} query:{len(queries)} warmup:{wc} input:{in_token:.1f} '
                           f'output:{out_token:.1f} edl:{edl:.3f}/{dl:.3f}/{pt:.3f}/{gt:.3f} time:{t:.3f} '
                           f'speed:{speed:.1f} mem:{mem:.3f} acc:{score:.4f}')
                print(log_str)
                if self.logger is not None:
                    self.logger.write(log_str + '\n')
                    self.logger.flush()
        return outputs

    # ------------------------------------------------------------------------------------------- perf_check_trie
    @staticmethod
    def perf_check_trie(lookahead_cache, warmup_ids, input_ids, output_ids, max_node_rate=16, decoding_length=64,
                        branch_length=24, edl=8, verbose=True):
        """Trie-only timing loop of benchmark.py:353-395 (put per prompt, bat_get every `edl` output tokens, stream_put of
        the output).  Works on ANY object with the LookaheadCache surface, so the native trie and the reference's Python
        trie can be timed side by side.  -> dict of the printed numbers (seconds)."""
        lookahead_cache.max_output_node = decoding_length * max_node_rate
        lookahead_cache.fresh()
        for ids_ in warmup_ids:
            lookahead_cache.put(list(ids_), branch_length=branch_length + 1, mode='output', idx=0, final=False)
        count = len(input_ids)
        put_count = get_count = 0
        put_time = get_time = 0.0
        for i in range(count):
            in_ids, out_ids = list(input_ids[i]), list(output_ids[i])
            put_count += len(in_ids)
            ts = time.time()
            lookahead_cache.put(in_ids, branch_length=branch_length + 1, mode='input', idx=0, final=False)
            put_time += time.time() - ts
            ts = time.time()
            for j in range(0, len(out_ids) - 1, edl):
                get_count += 1
                lookahead_cache.bat_get([out_ids[j:j + 2]], decoding_length=decoding_length, branch_length=branch_length,
                                        decoding_cursors=[j], mode='mix', indices=[0], decoding_mode='hier')
            get_time += time.time() - ts
            put_count += len(out_ids)
            ts = time.time()
            for j in range(0, len(out_ids) - 1, edl):
                lookahead_cache.stream_put(out_ids[j:j + edl], branch_length=branch_length + 1, mode='output', idx=0,
                                           final=False)
            lookahead_cache.stream_put([], branch_length=branch_length + 1, mode='output', idx=0, final=True)
            put_time += time.time() - ts
        res = {'samples': count, 'put_tokens': put_count, 'put_s': put_time, 'put_us_per_token': 1e6 * put_time / max(put_count, 1),
               'gets': get_count, 'get_s': get_time, 'get_ms_per_query': 1e3 * get_time / max(get_count, 1)}
        if verbose:
            print(f'\nparam:{max_node_rate}/{decoding_length}/{branch_length} sample:{count} '
                  f'put:{put_count}/{put_time:.2f}/{res["put_us_per_token"] / 1e3:.2f}/{1e3 * put_time / max(count, 1):.2f} '
                  f'get:{get_count}/{get_time:.2f}/{res["get_ms_per_query"]:.2f}/{1e3 * get_time / max(count, 1):.2f}\n')
        return res

    def grid_search(self, queries, warmup_ids=None, sizes=(16, 32, 64), lens=(4, 8, 12, 16), **kw):
        """benchmark.py:455-468: perf_check over the (decoding_length, branch_length) grid; -> best cell."""
        res = self.perf_check(queries, warmup_ids=warmup_ids, sizes=sizes, lens=lens, **kw)
        best = max(res.items(), key=lambda kv: kv[1]) if res else None
        print('best (decoding_length, branch_length):', best)
        return res
