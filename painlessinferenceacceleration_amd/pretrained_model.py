# -*- coding: utf-8 -*-
"""lookahead_generation() on the MI355X engine — same entry point, arguments and outputs as
LookaheadPreTrainedModel.lookahead_generation (lookahead/lookahead/common/pretrained_model.py:947-1268).

Per verify step the host does exactly three things: one native trie query (la_cache_hier_get), one
la_llama_step (h2d of ids + 64-bit row masks, the captured graph: forward, accept scan, KV commit; d2h of the
accepted tokens) and one native trie update (la_cache_stream_put).  The reference's >= 15 host<->device
synchronisations per step become one.
"""
import time
import warnings
from threading import Thread

import numpy as np
import torch

from . import _lib
from .lookahead_cache import LookaheadCache
from .lookahead_generation_utils import GenerationMode, LookaheadDecoderOnlyOutput, resolve_generate_args

_ONE = np.array([1], dtype=np.uint64)


def _pack_rows(mask, W=None):
    """0/1 mask [T][T] -> packed ancestor rows: uint64[T] (T <= 64) or uint64[T][W] (word w = columns 64 w .. 64 w + 63)."""
    m = np.asarray(mask).astype(np.uint64)
    T = m.shape[1]
    W = (T + 63) // 64 if W is None else W
    out = np.zeros((m.shape[0], W), dtype=np.uint64)
    for w in range(W):
        cols = m[:, 64 * w:64 * w + 64]
        if cols.shape[1]:
            out[:, w] = (cols << np.arange(cols.shape[1], dtype=np.uint64)[None, :]).sum(axis=1).astype(np.uint64)
    return out[:, 0] if W == 1 else out


def _parents_of(rowmask, T):
    """parent row of every tree row = highest set bit below the row over the concatenated ancestor words (DFS order)"""
    rm = np.asarray(rowmask, dtype=np.uint64)
    if rm.ndim == 1:
        rm = rm[:, None]
    parent = [-1] * T
    for j in range(1, T):
        below = 0
        for w in range(rm.shape[1]):
            below |= int(rm[j, w]) << (64 * w)
        below &= (1 << j) - 1
        parent[j] = below.bit_length() - 1
    return parent


def _max_length_of(stopping_criteria, max_length):
    if max_length is not None:
        warnings.warn("`max_length` is deprecated in this function, use "
                      "`stopping_criteria=StoppingCriteriaList([MaxLengthCriteria(max_length=max_length)])` instead.",
                      UserWarning)
        return int(max_length)
    if stopping_criteria is None:
        return None
    ml = getattr(stopping_criteria, 'max_length', None)
    if ml is None and isinstance(stopping_criteria, int):
        ml = stopping_criteria
    return None if ml is None else int(ml)


def _custom_stop(stopping_criteria):
    """The user's criteria besides MaxLengthCriteria as one predicate stop(token_row) -> bool, or None when there are none.
    The reference evaluates `stopping_criteria(input_ids, scores)` after every verify step (pretrained_model.py:1225-1226;
    batch: once per sample on input_ids[i:i+1, :cur+1], pretrained_model_batch.py:1284); scores = the tuple collected so far under
    output_scores, else None.  The length criterion itself is the `len(seq) >= max_length` test of the loops."""
    if stopping_criteria is None or isinstance(stopping_criteria, int) or not callable(stopping_criteria):
        return None
    try:
        crit = list(stopping_criteria)
    except TypeError:
        crit = [stopping_criteria]
    crit = [c for c in crit if type(c).__name__ != 'MaxLengthCriteria']
    if not crit:
        return None

    def stop(token_row, device='cpu', scores=None):
        ids = torch.tensor([list(token_row)], dtype=torch.long, device=device)
        for c in crit:
            r = c(ids, scores)
            if bool(r.any()) if torch.is_tensor(r) else bool(r):
                return True
        return False
    return stop


class LookaheadPreTrainedModel(object):
    """Mixin over an object that owns `self.engine` (LlamaVerifyEngine) and optionally `self.lookahead_cache`."""

    engine = None
    generation_config = None

    # ---------------------------------------------------------------------------------------- draft retrieval
    def lookahead_prepare_inputs_for_generation(self, tail_ids, decoding_kwargs, seq_len):
        """pretrained_model.py:666-756, decode branch: query the trie with the last `max_query_length` tokens.
        Returns (ids int32[T], rowmask uint64[T]) — the rank-4 mask is never materialised (a13/a14)."""
        decoding_length = decoding_kwargs.get('decoding_length', 64)
        branch_length = decoding_kwargs.get('branch_length', 12)
        decoding_mode = decoding_kwargs.get('decoding_mode', 'hier')
        max_length = decoding_kwargs.get('max_length', 2048)
        max_query_length = decoding_kwargs.get('max_query_length', 2)
        update_branch_length = min(branch_length, max_length - seq_len - 1)
        assert update_branch_length >= 0, f'{branch_length=} {max_length=} {seq_len=} {update_branch_length=}'
        qids = tail_ids[-max_query_length:]
        if decoding_mode in ('hier', 'par', 'one'):
            decoding_mode = decoding_mode + '_mix'
        fmt, mode = decoding_mode.split('_')
        tidx = int(decoding_kwargs.get('_trie_idx', 0))      # input-frequency plane: 0, or the global batch index of a sharded job
        ts = time.time()
        if fmt == 'hier' and decoding_kwargs.get('device_trie', False) and len(qids) <= 8 and decoding_length <= _lib.LA_TREE_WIDE_MAX:
            # draft from the workgroup-per-query trie walk over the incremental device mirror (csrc/la_trie_wg.hip; trees wider than one
            # 64-row block come back with uint64[T][4] row masks): bit-identical to the host query; opt-in here because one host query
            # (~20 us) is faster than sync + launch + D2H at bs = 1
            got = self._device_trie().hier_get([list(qids)], decoding_length=decoding_length, branch_length=update_branch_length,
                                               min_input_size=0, min_output_size=max(decoding_length // 2, 1), mode=mode, idxs=[0])[0]
            ids, rowmask, sizes = np.asarray(got[0], dtype=np.int32), np.asarray(got[1], dtype=np.uint64), got[2]
        elif fmt == 'one' and decoding_kwargs.get('device_trie', False) and len(qids) <= 8 and update_branch_length < 64:
            got = self._device_trie(narrow=True).one_get([list(qids)], decoding_length=decoding_length, branch_length=update_branch_length,
                                              mode=mode, idxs=[0])[0]
            ids, rowmask, sizes = np.asarray(got[0], dtype=np.int32), np.asarray(got[1], dtype=np.uint64), got[2]
        elif fmt == 'par' and decoding_kwargs.get('device_trie', False) and len(qids) <= 8 and decoding_length <= 64:
            # par = the hierarchical draft re-laid as independent chains (lookahead_cache.py:441-488): the device retrieves, the
            # re-layout (a handful of set operations on <= 64 rows) stays on the host
            got = self._device_trie(narrow=True).hier_get([list(qids)], decoding_length=decoding_length, branch_length=update_branch_length,
                                                          min_input_size=0, min_output_size=max(decoding_length // 2, 1), mode=mode, idxs=[0])[0]
            T0 = len(got[0])
            dense = np.array([[(int(got[1][i]) >> j) & 1 for j in range(T0)] for i in range(T0)], dtype=np.int64).reshape(T0, T0)
            lst, mask, sizes = self.lookahead_cache.par_layout(got[0], dense)
            ids = np.asarray(lst, dtype=np.int32)
            rowmask = _pack_rows(mask, None)
        elif fmt == 'hier':
            ids, rowmask, _, sizes = self.lookahead_cache.hier_get_packed(
                qids, decoding_length=decoding_length, branch_length=update_branch_length, min_input_size=0,
                min_output_size=max(decoding_length // 2, 1), mode=mode, idx=tidx)
        else:
            lst, mask, sizes = getattr(self.lookahead_cache, fmt + '_get')(
                qids, decoding_length=decoding_length, branch_length=update_branch_length, min_input_size=0,
                min_output_size=max(decoding_length // 2, 1), mode=mode, idx=tidx)
            ids = np.asarray(lst, dtype=np.int32)
            rowmask = _pack_rows(mask, (decoding_length + 63) // 64 if decoding_length > 64 and len(lst) > 1 else None)
        decoding_kwargs['qts'].append(time.time() - ts)
        decoding_kwargs.update({'decoding_qids': qids, 'decoding_ids': ids, 'sizes': sizes})
        return ids, rowmask

    def _device_trie(self, narrow=False):
        """DeviceTrie over self.lookahead_cache, input-frequency plane of idx 0 (rebuilt when the cache object changes; narrow: also when
        its result block grew to 256 rows per query — the one-branch walk writes 64-row blocks)."""
        from .device_trie import DeviceTrie
        dt = getattr(self, '_dev_trie', None)
        if dt is None or dt.cache is not self.lookahead_cache or dt._revoked or (narrow and dt.rows != 64):        # another DeviceTrie took the cache's mirror
            dt = self._dev_trie = DeviceTrie(self.lookahead_cache, idxs=[0], device=self.engine.device)
        return dt

    # ---------------------------------------------------------------------------------------------- the loop
    @torch.no_grad()
    def lookahead_generation(self, *args, **kwargs):
        """Entry point (signature of the reference's lookahead_generation, see _lookahead_generation).  The step loop
        runs with the cyclic garbage collector paused: a generation-2 pass over torch's object graph stalls the host
        thread for tens of ms (measured 37 ms = 9 verify steps) and the loop creates no reference cycles."""
        import gc
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            return self._lookahead_generation(*args, **kwargs)
        finally:
            if was_enabled:
                gc.enable()

    def _lookahead_generation(self, input_ids, logits_processor=None, stopping_criteria=None, max_length=None,
                             pad_token_id=None, eos_token_id=None, output_attentions=None,
                             output_hidden_states=None, output_scores=None, return_dict_in_generate=None,
                             synced_gpus=False, streamer=None, **model_kwargs):
        # processors are applied sequentially along the accepted path (pretrained_model.py:834), so a non-empty
        # list (or sampling) takes the host-walked path: device forward only (mode 2), one logits row per accepted token,
        # host-decided commit.  The empty-list greedy default stays entirely on the device.
        if isinstance(logits_processor, (list, tuple)) and not callable(logits_processor):
            from transformers import LogitsProcessorList      # a plain list of processors: the reference's generate() wraps it the same way
            logits_processor = LogitsProcessorList(list(logits_processor))
        sequential = (logits_processor is not None and len(logits_processor) > 0) or \
            bool(model_kwargs.get('decoding_kwargs', {}).get('do_sample', False))
        if output_attentions or output_hidden_states:
            raise NotImplementedError('attentions / hidden_states are intermediates the device path never materialises '
                                      '(fused attention, activations in MFMA fragment order); scores are returned (output_scores)')
        gc = self.generation_config
        pad_token_id = pad_token_id if pad_token_id is not None else getattr(gc, 'pad_token_id', None)
        eos_token_id = eos_token_id if eos_token_id is not None else getattr(gc, 'eos_token_id', None)
        if isinstance(eos_token_id, int):
            eos_token_id = [eos_token_id]
        return_dict_in_generate = bool(return_dict_in_generate) if return_dict_in_generate is not None \
            else bool(getattr(gc, 'return_dict_in_generate', False))
        output_scores = bool(output_scores) if output_scores is not None else bool(getattr(gc, 'output_scores', False))
        # `scores` as the reference returns them (pretrained_model.py:1092, 1195, 1208-1209): ONE entry per verify step, and that entry
        # is model_kwargs['next_tokens_scores'] — which only the no-draft branch writes (:795: the prefill and steps whose retrieval
        # came back empty).  A step WITH drafts therefore appends the previous no-draft step's tensor again (pinned on
        # tests/golden/llama_tiny_scores_fp32.npz).  decoding_kwargs['fresh_scores'] = True (an extension) appends, on draft steps,
        # the processed logits row the step's LAST emitted token was picked from instead.
        scores = () if (return_dict_in_generate and output_scores) else None

        if not hasattr(self, 'lookahead_cache') or self.lookahead_cache is None:
            self.lookahead_cache = LookaheadCache()
        decoding_kwargs = model_kwargs['decoding_kwargs']
        self.lookahead_cache.eos_ids = eos_token_id
        self.lookahead_cache.stop_words = decoding_kwargs.get('stop_words', {})
        decoding_kwargs.update({'eos': eos_token_id[0] if eos_token_id is not None else 2,
                                'edls': [], 'dls': [], 'fts': [], 'qts': []})
        stop_max_length = _max_length_of(stopping_criteria, max_length)
        if stop_max_length is None:
            raise ValueError('lookahead_generation needs a MaxLengthCriteria (stopping_criteria.max_length)')
        decoding_length = decoding_kwargs.get('decoding_length', 64)
        # trees wider than one 64-row block run as consecutive blocks of one multi-block pass (eng.tstep; the reference's best
        # published setting is decoding_length=128, branch_length=32, lookahead/README.md:100)
        wide = int(decoding_length) > _lib.LA_TREE_MAX
        if not 1 <= int(decoding_length) <= _lib.LA_TREE_WIDE_MAX:
            raise ValueError(f'decoding_length={decoding_length}: a draft tree holds at most {_lib.LA_TREE_WIDE_MAX} tokens '
                             f'(4 blocks of 64 rows; the reference grid-searches up to 256, benchmarks/benchmark.py:358)')
        if wide and 64 * int(getattr(self.engine, 'max_blocks', 0) or 0) < int(decoding_length):
            raise ValueError(f'decoding_length={decoding_length} needs an engine created with max_blocks >= '
                             f'{(int(decoding_length) + 63) // 64} (LlamaForCausalLM(..., max_blocks=...))')
        if int(decoding_kwargs.get('branch_length', 12)) + 1 > (_lib.LA_MOUT_TOKS if wide else 64):
            raise ValueError(f'branch_length={decoding_kwargs.get("branch_length")}: a step emits at most {_lib.LA_MOUT_TOKS} tokens')
        if int(decoding_kwargs.get('max_query_length', 2)) < 1:
            raise ValueError('max_query_length must be >= 1')
        decoding_kwargs['max_length'] = stop_max_length
        decoding_kwargs['decoding_max_length'] = stop_max_length + decoding_length + 1
        attention_mask = model_kwargs.get('attention_mask', None)
        if attention_mask is not None and attention_mask.dim() == 2 and not bool(attention_mask.bool().all()):
            raise NotImplementedError('left-padded prompts need the batch path (bs=1 engine assumes no padding)')

        assert input_ids.size(0) == 1
        out_device = input_ids.device
        seq = input_ids[0].tolist()
        branch_length = decoding_kwargs.get('branch_length', 12)
        eng = self.engine
        cap = eng._capacity() if hasattr(eng, '_capacity') else eng.max_keys
        assert stop_max_length + decoding_length + 1 <= cap, f'engine KV capacity {cap} < max_length + decoding_length + 1'
        # Sharded job (distributed.py): decoding_kwargs['gather'] = AcceptedTokenGather(..., b_loc=1, mode='strict' | 'split-phase').
        # This rank decodes ITS sequence; every trie call is keyed by the sequence's GLOBAL batch index, the per-step stream_put
        # becomes the all-gather of all ranks' accepted tokens applied in batch-index order (pretrained_model_batch.py:1254-1259),
        # and the rank keeps serving the collective after its own sequence has finished.
        gather = decoding_kwargs.get('gather', None)
        if gather is not None:
            assert gather.b_loc == 1, 'one sequence per rank on this loop (pretrained_model_batch.py takes B_loc > 1)'
            assert not decoding_kwargs.get('device_trie', False), 'sharded decoding keeps the trie replicas on the host'
        tidx = gather.global_index(0) if gather is not None else 0
        decoding_kwargs['_trie_idx'] = tidx
        if gather is not None:
            gather.begin_request()         # no state of an earlier request (un-collected gather, all_done) leaks into this one
        if gather is not None:            # every replica holds every sequence's input frequencies, put in batch-index order
            for b_, p_ in enumerate(gather.exchange_prompts([seq[1:]])):
                self.lookahead_cache.put(p_, branch_length=branch_length + 1, mode='input', idx=b_)
        else:
            self.lookahead_cache.put(seq[1:], branch_length=branch_length + 1, mode='input', idx=0)
        ts = time.time()
        eng.reset()
        first = True
        eos_set = set(eos_token_id) if eos_token_id is not None else set()
        do_sample = bool(decoding_kwargs.get('do_sample', False))
        dm = decoding_kwargs.get('decoding_mode', 'hier')
        dm = dm + '_mix' if dm in ('hier', 'par', 'one') else dm
        native_mode = {'input': 0, 'output': 1, 'mix': 2}.get(dm.split('_')[1], 2)
        max_query_length = int(decoding_kwargs.get('max_query_length', 2))
        custom_stop = _custom_stop(stopping_criteria)      # user StoppingCriteria: evaluated per step, interpreter loop only
        fresh_scores = scores is not None and bool(decoding_kwargs.get('fresh_scores', False))
        if fresh_scores:
            sequential = True            # the host walk sees every logits row it picks from
        native_loop = (not sequential and streamer is None and dm.split('_')[0] == 'hier' and not wide
                       and custom_stop is None and gather is None and scores is None
                       and not decoding_kwargs.get('device_trie', False)
                       and not decoding_kwargs.get('debug_lookahead', False) and decoding_kwargs.get('native_loop', True)
                       and 1 <= max_query_length <= 8          # la_lookahead_decode's query buffer; longer queries use this loop
                       and len([e for e in (eos_token_id or []) if e is not None]) <= 8      # ... and its eos list (la_decode_params.eos[8])
                       and hasattr(eng, 'decode_native'))

        def pick(scores_ids, row):
            """next token from one logits row through the processor list (pretrained_model.py:833-839)"""
            ctx = torch.tensor([scores_ids], dtype=torch.long, device=eng.device)
            lg = eng.mlogits() if wide else eng.logits()
            sc = logits_processor(ctx, lg[row][None].clone()) if logits_processor is not None and \
                len(logits_processor) > 0 else lg[row][None]
            if scores is not None:
                picked[0] = sc.clone()
            if do_sample:
                return int(torch.multinomial(torch.softmax(sc.float(), dim=-1), num_samples=1)[0, 0])
            return int(torch.argmax(sc, dim=-1)[0])

        picked = [None]            # scores of the last pick() call
        last_scores = None         # what the reference's model_kwargs['next_tokens_scores'] holds
        flushed = False
        try:
            while True:
                if first:
                    host_row = sequential or scores is not None      # the last prompt row's logits must stay readable
                    if wide:
                        tok = eng.mprefill(0, seq)
                        row = (len(seq) - 1) % (64 * eng.max_blocks)
                    else:
                        tok = eng.prefill(seq, fast=False) if host_row else eng.prefill(seq)
                        row = (len(seq) - 1) % 64
                    if host_row:
                        t = pick(seq, row)                           # sequential: THE pick; otherwise the same argmax, taken for its scores
                        next_tokens = [t] if sequential else [tok]
                    else:
                        next_tokens = [tok]
                    last_scores = picked[0]
                    decoding_kwargs['dls'].append(1)
                    decoding_kwargs['edls'].append(1)
                    first = False
                else:
                    ids, rowmask = self.lookahead_prepare_inputs_for_generation(seq, decoding_kwargs, len(seq))
                    if len(ids) == 0:
                        ids, rowmask = np.asarray(seq[-1:], dtype=np.int32), _ONE
                    if sequential:
                        T = len(ids)
                        if wide:
                            eng.tstep(ids, rowmask, mode=2)
                        else:
                            eng.verify_only(ids, rowmask)
                        parent = _parents_of(rowmask, T)
                        # every row whose parent is live and whose token was picked stays live; the first one supplies the
                        # next logits row (the reference's surviving leaf branches, pretrained_model.py:831, 850-860: in a
                        # par layout a shared prefix is duplicated across chains and the walk may move to a later chain)
                        cur, live, rows, next_tokens = 0, {0}, [0], []
                        while True:
                            t = pick(seq + next_tokens, cur)
                            next_tokens.append(t)
                            nxt = [j for j in range(1, T) if parent[j] in live and int(ids[j]) == t]
                            if not nxt:
                                break
                            cur, live = nxt[0], set(nxt)
                            rows.append(cur)
                        if wide:
                            eng.tcommit(rows, T)
                        else:
                            eng.commit(rows)
                    elif wide:
                        next_tokens, _ = eng.tstep(ids, rowmask, mode=0)
                    elif gather is not None and hasattr(eng, 'step_async'):
                        eng.step_async(ids, rowmask, mode=0)
                        gather.overlap(self.lookahead_cache, branch_length)      # split-phase: the previous step's gather + puts, under this pass
                        next_tokens, _ = eng.step_finish()
                    else:
                        next_tokens, _ = eng.step(ids, rowmask, mode=0)
                    if scores is not None:
                        if sequential:
                            if len(ids) <= 1 or fresh_scores:        # no-draft step (:783-795) — or every step (extension)
                                last_scores = picked[0]
                        elif len(ids) <= 1:
                            # device-accepted step without drafts: the block's only logits row, through the (empty) processor list
                            lg = eng.mlogits() if wide else eng.logits()
                            last_scores = lg[0][None].clone()
                    decoding_kwargs['dls'].append(len(ids))
                    decoding_kwargs['edls'].append(len(next_tokens))
                    if decoding_kwargs.get('debug_lookahead', False):
                        tok = decoding_kwargs.get('tokenizer', None)
                        words = '' if tok is None else tok.decode(next_tokens)
                        print(f'decoding_length:{len(ids)} accept_length:{len(next_tokens)} '
                              f'query:{decoding_kwargs["decoding_qids"]} hits:{decoding_kwargs["sizes"]} '
                              f'accept_token:{next_tokens} accept_word:{words}')
                seq.extend(next_tokens)
                if scores is not None:
                    scores += (last_scores,)
                if streamer is not None:
                    streamer.put(np.array([next_tokens]))
                finished = len(seq) >= stop_max_length or any(t in eos_set for t in next_tokens) or \
                    (custom_stop is not None and custom_stop(seq, out_device, scores))           # :1225-1231
                if gather is not None:
                    gather.step_update(self.lookahead_cache, next_tokens, branch_length, done=finished)
                else:
                    self.lookahead_cache.stream_put(next_tokens, branch_length=branch_length + 1, final=False,
                                                    mode='output', idx=0)
                te = time.time()
                decoding_kwargs['fts'].append(te - ts)
                ts = te
                if finished:
                    if gather is not None:
                        gather.drain(self.lookahead_cache, branch_length)          # until every rank's sequence has finished
                        gather.flush(self.lookahead_cache, branch_length)
                    else:
                        self.lookahead_cache.stream_put([], branch_length=branch_length + 1, final=True, mode='output', idx=0)
                    flushed = True
                    break
                if native_loop:
                    # every remaining step runs in la_lookahead_decode: same calls in the same order (hier_get -> la_llama_step
                    # -> stream_put -> stop checks), without the interpreter between them
                    new, dls_, edls_, fts_, qts_, fin = eng.decode_native(
                        self.lookahead_cache, seq, stop_max_length, eos_ids=eos_set, decoding_length=decoding_length,
                        branch_length=branch_length, max_query_length=decoding_kwargs.get('max_query_length', 2),
                        mode=native_mode, idx=0)
                    seq.extend(new)
                    decoding_kwargs['dls'].extend(dls_)
                    decoding_kwargs['edls'].extend(edls_)
                    decoding_kwargs['fts'].extend(fts_)
                    decoding_kwargs['qts'].extend(qts_)
                    assert fin
                    self.lookahead_cache.stream_put([], branch_length=branch_length + 1, final=True, mode='output', idx=0)
                    flushed = True
                    break
        finally:
            # an exception mid-generation must not leave this request's stream buffer / input frequencies behind (they would
            # mix into the next request); the reference flushes on the normal path only (pretrained_model.py:1236-1239)
            if not flushed:
                if gather is not None:
                    # sharded request: this rank failed (or was interrupted) mid-request — it must still serve the per-step collective
                    # until every rank has finished, and flush ALL B sequences, or the other ranks would block in drain() / finish() and
                    # this replica would keep un-flushed stream buffers of the other sequences (AcceptedTokenGather.abort_request)
                    gather.abort_request(self.lookahead_cache, branch_length)
                else:
                    self.lookahead_cache.stream_put([], branch_length=branch_length + 1, final=True, mode='output', idx=tidx)
        if streamer is not None:
            streamer.end()
        sequences = torch.tensor([seq], dtype=torch.long, device=out_device)
        if return_dict_in_generate:
            kwargs = {k: decoding_kwargs[k] for k in ('dls', 'edls', 'fts', 'qts')}
            if scores is not None:
                scores = tuple(s_.to(out_device) for s_ in scores)
            return LookaheadDecoderOnlyOutput(sequences=sequences, scores=scores, attentions=None, hidden_states=None,
                                              kwargs=kwargs)
        return sequences

    # ---------------------------------------------------------------------------------- plain greedy (mode off)
    @torch.no_grad()
    def greedy_search(self, input_ids, max_length, eos_token_id=None, logits_processor=None, do_sample=False,
                      stopping_criteria=None):
        """Plain decoding through the same engine (T=1 blocks): the `use_lookahead=False` leg of the reference's
        examples (examples/llama_example.py:39-69).  With a processor list or sampling each token is picked on the host
        from the block's logits row (forward-only step + commit)."""
        eos = set([eos_token_id] if isinstance(eos_token_id, int) else (eos_token_id or []))
        seq = input_ids[0].tolist()
        eng = self.engine
        eng.reset()
        host_pick = do_sample or (logits_processor is not None and len(logits_processor) > 0)
        custom_stop = _custom_stop(stopping_criteria)        # the caller's criteria besides the length bound, once per token

        def pick(row):
            ctx = torch.tensor([seq], dtype=torch.long, device=eng.device)
            scores = eng.logits()[row][None].clone()
            if logits_processor is not None and len(logits_processor) > 0:
                scores = logits_processor(ctx, scores)
            if do_sample:
                return int(torch.multinomial(torch.softmax(scores.float(), dim=-1), num_samples=1)[0, 0])
            return int(torch.argmax(scores, dim=-1)[0])

        tok = eng.prefill(seq, fast=False) if host_pick else eng.prefill(seq)
        if host_pick:
            tok = pick((len(seq) - 1) % 64)
        seq.append(tok)
        while len(seq) < max_length and tok not in eos and not (custom_stop is not None and custom_stop(seq, input_ids.device)):
            if host_pick:
                eng.verify_only(np.asarray([tok], dtype=np.int32), _ONE)
                tok = pick(0)
                eng.commit([0])
            else:
                toks, _ = eng.step(np.asarray([tok], dtype=np.int32), _ONE, mode=0)
                tok = toks[0]
            seq.append(tok)
        return torch.tensor([seq], dtype=torch.long, device=input_ids.device)

    # ------------------------------------------------------------------------------------------- front door
    def _get_generation_mode(self, decoding_kwargs):
        """pretrained_model.py:55-106 reduced to the two modes this path serves."""
        dk = decoding_kwargs or {}
        if dk.get('use_lookahead', False) and dk.get('decoding_length', 64) > 1 and dk.get('branch_length', 12) > 0:
            return GenerationMode.LOOKAHEAD_GENERATION
        return GenerationMode.GREEDY_SEARCH

    def generate(self, inputs=None, generation_config=None, logits_processor=None, stopping_criteria=None,
                 prefix_allowed_tokens_fn=None, synced_gpus=None, assistant_model=None, streamer=None, **kwargs):
        """generate() with the reference's signature (common/pretrained_model.py:110-121) for the two modes this package serves —
        lookahead and plain greedy / sampling through the same engine.  Everything the reference derives before it dispatches is
        derived the same way (lookahead_generation_utils.resolve_generate_args): keyword > generation_config > model defaults,
        config-derived logits processors (repetition_penalty, no_repeat_ngram_size, bad_words_ids, min_length, min_new_tokens)
        with the caller's `logits_processor` merged behind them, MaxLengthCriteria / MaxTimeCriteria with the caller's
        `stopping_criteria` merged behind them; the lookahead branch receives processors + criteria and no warper, exactly as
        :428-441; temperature / top_k / top_p act in the sampling mode (`do_sample` without lookahead, :465-479).
        Call shapes of the reference's own callers work unchanged: examples/llama_example.py:51-60,
        benchmarks/benchmark.py:282-300."""
        if prefix_allowed_tokens_fn is not None or assistant_model is not None:
            raise NotImplementedError('prefix_allowed_tokens_fn / assistant_model: outside the lookahead path (SURVEY section 2, out of scope)')
        input_ids = inputs if inputs is not None else kwargs.pop('input_ids', None)
        kwargs.pop('input_ids', None)
        if input_ids is None:
            raise ValueError('generate() needs input_ids')
        ga, model_kwargs = resolve_generate_args(self.generation_config, input_ids.size(1), generation_config=generation_config,
                                                 logits_processor=logits_processor, stopping_criteria=stopping_criteria,
                                                 device=getattr(getattr(self, 'engine', None), 'device', None), **kwargs)
        attention_mask = model_kwargs.pop('attention_mask', None)
        dk = ga.decoding_kwargs
        if streamer is not None:
            pass                                             # the loops feed the streamer themselves (prompt first, :318-319)
        if self._get_generation_mode(dk) == GenerationMode.LOOKAHEAD_GENERATION:
            dk['generation_mode'] = GenerationMode.LOOKAHEAD_GENERATION
            dk['do_sample'] = ga.do_sample
            return self.lookahead_generation(input_ids, logits_processor=ga.logits_processor if len(ga.logits_processor) else None,
                                             stopping_criteria=ga.stopping_criteria, pad_token_id=ga.pad_token_id,
                                             eos_token_id=ga.eos_token_id, output_scores=ga.output_scores or None,
                                             return_dict_in_generate=ga.return_dict_in_generate,
                                             streamer=streamer, attention_mask=attention_mask, decoding_kwargs=dk)
        from transformers import LogitsProcessorList
        procs = LogitsProcessorList(list(ga.logits_processor) + list(ga.logits_warper))
        out = self.greedy_search(input_ids, ga.max_length, ga.eos_token_id if ga.eos_token_id is not None
                                 else getattr(self.generation_config, 'eos_token_id', None),
                                 logits_processor=procs if len(procs) else None, do_sample=ga.do_sample,
                                 stopping_criteria=ga.stopping_criteria)
        return LookaheadDecoderOnlyOutput(sequences=out, kwargs={}) if ga.return_dict_in_generate else out

    def stream_generate(self, *args, **kwargs):
        """pretrained_model.py:1323-1350: run generate() on a worker thread, yield from the streamer."""
        streamer = kwargs.get('streamer')
        assert streamer is not None, 'stream_generate needs a streamer with an iterator interface'
        Thread(target=self.generate, args=args, kwargs=kwargs).start()
        for item in streamer:
            yield item
