# -*- coding: utf-8 -*-
"""Batch (bs>1) lookahead_generation() on the MI355X engine — the twin of
lookahead/lookahead/common/pretrained_model_batch.py:1002-1330 with the same entry point, arguments and outputs.

What changes on the device side (see include/lookahead_hip.h, "cursor batch"): every sample owns a slot of the KV
cache; the drafts of all active samples are packed, unpadded, into ONE 64-row verify block (the reference's own budget
rule — decoding_length // bs per sample, halved again inside bat_get, SURVEY H2 — never asks for more), so the weights
stream through the CUs once per step for the whole batch.  Per step the host does: one native trie query per sample,
one la_llama_bstep (forward + per-sample accept walk + KV row moves on device), one native trie update per sample.

Budget modes.  Default = the reference's rule: (decoding_length // bs) // bs rows per sample.  When the batch asks
for more than 64 rows in total — or decoding_kwargs['per_sample_budget'] is set, which gives EVERY sample its own
decoding_length-token tree as BASELINE configs 3-5 state (an extension: the reference halves the budget twice) — each sample
gets a 64-row block of its own and the blocks of a step run in ONE pass over the weights (la_llama_mstep, M = 64 x samples
rows through the LDS-staged GEMM family).  Tokens are unaffected by the budget (lookahead is lossless); dls/edls differ.
"""
import time
import warnings

import numpy as np
import torch

from . import _lib
from .lookahead_cache import LookaheadCache
from .lookahead_generation_utils import LookaheadDecoderOnlyOutput, resolve_generate_args
from .pretrained_model import _custom_stop, _max_length_of

_ONE = np.array([1], dtype=np.uint64)


class LookaheadPreTrainedModel(object):
    """Mixin over an object that owns `self.engine` (LlamaVerifyEngine with n_slots >= batch size)."""

    engine = None
    generation_config = None

    def lookahead_prepare_inputs_for_generation(self, rows, batch_indices, decoding_kwargs):
        """pretrained_model_batch.py:706-743: per-sample drafts for the last two tokens at each cursor.
        -> list of (ids int32[T_b], rowmask uint64[T_b]) in batch order."""
        decoding_length = decoding_kwargs.get('decoding_length', 64)
        branch_length = decoding_kwargs.get('branch_length', 12)
        decoding_mode = decoding_kwargs.get('decoding_mode', 'hier')
        qids = [r[-2:] for r in rows]
        if decoding_mode in ('hier', 'par', 'one'):
            decoding_mode = decoding_mode + '_mix'
        fmt, mode = decoding_mode.split('_')
        # trees wider than a 64-row block (<= LA_TREE_WIDE_MAX rows = up to 4 blocks of the pass) come from the hier walk — the host trie's
        # or, round 6, the device trie's workgroup kernel (multi-word row masks); the one-branch walk keeps the 64-row cap
        wide_ok = fmt == 'hier' and bool(getattr(self.engine, 'max_blocks', 0))
        cap_rows = min(_lib.LA_TREE_WIDE_MAX, 64 * int(self.engine.max_blocks)) if wide_ok else _lib.LA_TREE_MAX      # a tree fits one pass
        if decoding_kwargs.get('per_sample_budget', False):
            # every sample gets a decoding_length-token tree (bat_get divides its argument by the batch size once)
            sub = min(decoding_length, cap_rows) * len(qids)
        else:
            sub = max(decoding_length // len(qids), 1)                 # pretrained_model_batch.py:713
            sub = min(sub, cap_rows * len(qids))                       # a sample's tree never exceeds the rows one pass can give it
        ts = time.time()
        per_wide = min(sub // len(qids), _lib.LA_TREE_WIDE_MAX)
        if per_wide > _lib.LA_TREE_MAX and fmt == 'hier' and not decoding_kwargs.get('device_trie', False):
            # a per-sample budget wider than a 64-row block: bat_get's own rule sample by sample (lookahead_cache.py:534-541:
            # hier_get(decoding_length = budget, min_input_size = 0, min_output_size = max(budget // 2, 1))), multi-word row masks
            drafts = []
            for q_, i_ in zip(qids, batch_indices):
                ids_, rm_, _, sizes_ = self.lookahead_cache.hier_get_packed(q_, decoding_length=per_wide, branch_length=branch_length,
                                                                          min_input_size=0, min_output_size=max(per_wide // 2, 1),
                                                                          mode=mode, idx=i_)
                drafts.append((np.array(ids_, dtype=np.int32), np.array(rm_, dtype=np.uint64), list(sizes_)))
        elif decoding_kwargs.get('device_trie', False) and fmt == 'one':
            # one greedy chain per sample from one launch (la_trie_one_get_dev2); budget rule of bat_get (:534-541)
            per = sub // len(qids)
            got = self._device_trie(decoding_kwargs['_n_samples'], narrow=True).one_get(
                qids, idxs=batch_indices, decoding_length=per, branch_length=branch_length, mode=mode)
            drafts = [(np.asarray(g[0], dtype=np.int32), np.asarray(g[1], dtype=np.uint64), g[2]) for g in got]
        elif decoding_kwargs.get('device_trie', False) and fmt == 'hier':
            # the drafts of ALL active samples from one launch over the incremental device mirror of the trie (one workgroup per
            # sample, its own input-frequency plane; per-sample budgets above 64 rows come back with multi-word row masks): no host trie query on the step's critical path.  Same budget rule as
            # bat_get (lookahead_cache.py:534-541): per sample sub // bs rows, min_output_size = max(per // 2, 1).
            per = min(sub // len(qids), _lib.LA_TREE_WIDE_MAX)
            got = self._device_trie(decoding_kwargs['_n_samples']).hier_get(
                qids, idxs=batch_indices, decoding_length=per, branch_length=branch_length, min_input_size=0,
                min_output_size=max(per // 2, 1), mode=mode)
            drafts = [(np.asarray(g[0], dtype=np.int32), np.asarray(g[1], dtype=np.uint64), g[2]) for g in got]
        else:
            drafts = self.lookahead_cache.bat_get_packed(qids, decoding_length=sub, branch_length=branch_length, mode=mode,
                                                         indices=batch_indices, decoding_mode=fmt)
        decoding_kwargs['qts'].append(time.time() - ts)
        decoding_kwargs.update({'decoding_qids': qids, 'decoding_ids': [d[0] for d in drafts],
                                'hit_sizes': [d[2] for d in drafts], 'batch_indices': batch_indices})
        return [(d[0], d[1]) for d in drafts]

    def _device_trie(self, n_samples, updates=False, narrow=False):
        """DeviceTrie over self.lookahead_cache with one input-frequency plane per batch index (rebuilt when the cache object or
        the batch size changes).  updates=True (the chained loop with device-side stream_put) adds the token -> root table and the
        block capacities; without it a sync() patch carries no root-index pass (vocab-sized memset + kernel + meta copy).
        narrow=True (the chained step, the one-branch walk): a mirror whose result block grew to 256 rows per query (it served trees
        wider than a block) is replaced — those consumers read 64-row blocks."""
        from .device_trie import DeviceTrie
        dt = getattr(self, '_dev_trie', None)
        if dt is None or dt.cache is not self.lookahead_cache or len(dt.idxs) < n_samples or dt._revoked or \
                (updates and not dt.put_vocab) or (narrow and dt.rows != 64):
            dt = self._dev_trie = DeviceTrie(self.lookahead_cache, idxs=list(range(n_samples)), device=self.engine.device,
                                             put_vocab=self.engine.shape.vocab if updates else None)
        return dt

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
        except BaseException:
            # an abnormal exit (KeyboardInterrupt, a raising StoppingCriteria, a failed assert) may leave device-side trie inserts
            # the host never replayed: drop the mirror owner, the next generation builds a fresh image from the host trie
            self._dev_trie = None
            raise
        finally:
            if was_enabled:
                gc.enable()

    def _lookahead_generation(self, input_ids, logits_processor=None, stopping_criteria=None, max_length=None,
                             pad_token_id=None, eos_token_id=None, output_attentions=None,
                             output_hidden_states=None, output_scores=None, return_dict_in_generate=None,
                             synced_gpus=False, streamer=None, **model_kwargs):
        # the reference's batch loop always walks every sample's tree token by token with the processors applied to
        # input_ids[b, :cur+i+2] (pretrained_model_batch.py:814-875).  With an empty list and greedy decoding that walk equals the
        # device accept scan; a non-empty list or sampling takes the sequential path: forward-only step (mode 2), host walk over
        # the logits rows, host-decided commit (la_llama_bcommit / la_llama_mcommit).
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
        # `scores` as the reference's batch loop returns them (pretrained_model_batch.py:1145, 1247, 1263-1264): one [bs, vocab] entry per
        # loop iteration, and every entry is model_kwargs['next_tokens_scores'] — written by the PREFILL branch only (:789, :807), so the
        # tuple repeats the processed logits of the last prompt rows (pinned on tests/golden/llama_tiny_scores_fp32.npz)
        scores = () if (return_dict_in_generate and output_scores) else None
        prefill_scores = None
        if not hasattr(self, 'lookahead_cache') or self.lookahead_cache is None:
            self.lookahead_cache = LookaheadCache()
        decoding_kwargs = model_kwargs['decoding_kwargs']
        self.lookahead_cache.eos_ids = eos_token_id
        self.lookahead_cache.stop_words = decoding_kwargs.get('stop_words', {})
        pad = pad_token_id if pad_token_id is not None else 2
        decoding_kwargs.update({'pad': pad, 'edls': [], 'dls': [], 'fts': [], 'qts': []})
        decoding_kwargs['_n_samples'] = int(input_ids.shape[0])
        stop_max_length = _max_length_of(stopping_criteria, max_length)
        if stop_max_length is None:
            raise ValueError('lookahead_generation needs a MaxLengthCriteria (stopping_criteria.max_length)')
        custom_stop = _custom_stop(stopping_criteria)
        decoding_length = decoding_kwargs.get('decoding_length', 63)
        decoding_kwargs['max_length'] = stop_max_length
        decoding_kwargs['decoding_max_length'] = stop_max_length + decoding_length + 1
        branch_length = decoding_kwargs.get('branch_length', 8)

        out_device = input_ids.device
        ids0 = input_ids.cpu().numpy().astype(np.int64)
        bs, P = ids0.shape
        attention_mask = model_kwargs.get('attention_mask', None)
        if attention_mask is None:
            am = np.ones_like(ids0)
        elif attention_mask.dim() == 2:
            am = attention_mask.cpu().numpy().astype(np.int64)
        else:
            raise ValueError(f'unsupport attention_mask.shape:{attention_mask.shape}')
        eng = self.engine
        assert bs <= eng.n_slots, f'batch of {bs} needs an engine with n_slots >= {bs} (has {eng.n_slots})'
        cap = eng._capacity() if hasattr(eng, '_capacity') else eng.max_keys
        # rows one sample's tree can take in a step: at most decoding_length (the whole budget once the other samples have retired,
        # :713), never more than one pass can give it (lookahead_prepare_inputs_for_generation: 64 rows unless the host trie's hier
        # walk feeds a multi-block engine); a verify block is reserved whole
        _fmt = str(decoding_kwargs.get('decoding_mode', 'hier')).split('_')[0]
        _wide_ok = _fmt == 'hier' and bool(getattr(eng, 'max_blocks', 0))
        _cap_rows = min(_lib.LA_TREE_WIDE_MAX, 64 * int(eng.max_blocks)) if _wide_ok else _lib.LA_TREE_MAX
        _per = max(min(int(decoding_length), _cap_rows), _lib.LA_TREE_MAX)
        assert stop_max_length + _per + 1 <= cap, f'engine KV capacity {cap} per slot is too small'
        # Sharded job (distributed.py): decoding_kwargs['gather'] = AcceptedTokenGather(..., b_loc=bs, mode=...).  This rank decodes ITS
        # bs sequences; local row i is global batch index gather.global_index(i) in every trie call, the per-step stream_put_many
        # becomes the all-gather of all ranks' accepted tokens applied in batch-index order (:1254-1259), and the rank keeps serving
        # the collective until every rank has finished.
        gather = decoding_kwargs.get('gather', None)
        if gather is not None:
            assert gather.b_loc == bs, f'AcceptedTokenGather(b_loc={gather.b_loc}) vs a local batch of {bs}'
            assert not decoding_kwargs.get('device_trie', False), 'sharded decoding keeps the trie replicas on the host'
        gi = (lambda b_: gather.global_index(b_)) if gather is not None else (lambda b_: b_)
        # decoding_kwargs['overlap_trie_update'] (extension, off by default: the reference updates before it queries): the trie update
        # of step k runs on the host AFTER the verify pass of step k + 1 has been queued — drafts see a step's tokens one step later
        # (the split-phase order of the sharded job), emitted tokens are unaffected
        overlap_put = bool(decoding_kwargs.get('overlap_trie_update', False)) and gather is None
        deferred_put = [None]

        def run_deferred_put():
            if deferred_put[0] is not None:
                self.lookahead_cache.stream_put_many(deferred_put[0], branch_length=branch_length + 1, final=False)
                deferred_put[0] = None

        if gather is not None:            # every replica holds every sequence's input frequencies, put in batch-index order
            gather.begin_request()        # no state of an earlier request (un-collected gather, all_done, failed ranks) leaks into this one
            for b_, p_ in enumerate(gather.exchange_prompts([ids0[i, 1:-1].tolist() for i in range(bs)])):
                self.lookahead_cache.put(p_, branch_length=branch_length + 1, mode='input', idx=b_)
        else:
            for i in range(bs):                                                 # :1204-1207 (pads included, as there)
                self.lookahead_cache.put(ids0[i, 1:-1].tolist(), branch_length=branch_length + 1, mode='input', idx=i)
        rows = [ids0[i].tolist() for i in range(bs)]      # padded-coordinate token rows; cursor = len(row) - 1
        finished_rows = [None] * bs
        batch_indices = list(range(bs))
        eos_set = set(eos_token_id) if eos_token_id is not None else set()
        try:
            ts = time.time()
            eng.reset_slot(-1)
            # prefill: valid prompt tokens of every sample, packed into shared 64-row chain blocks
            prompts = {i: ids0[i][am[i] == 1].tolist() for i in range(bs)}
            multi = bool(getattr(eng, 'max_blocks', 0))
            do_sample = bool(decoding_kwargs.get('do_sample', False))

            picked = [None]

            def pick(ctx_ids, row):
                """next token from one logits row through the processor list (pretrained_model_batch.py:840-846)"""
                sc = row[None]
                if logits_processor is not None and len(logits_processor) > 0:
                    ctx = torch.tensor([ctx_ids], dtype=torch.long, device=eng.device)
                    sc = logits_processor(ctx, sc.clone())
                picked[0] = sc
                if do_sample:
                    return int(torch.multinomial(torch.softmax(sc.float(), dim=-1), num_samples=1)[0, 0])
                return int(torch.argmax(sc, dim=-1)[0])

            if sequential or scores is not None:
                # prompts one slot after the other: the last prompt row's logits stay readable for the processor call of :783
                # (batch-wise there; the processors are row-wise, so one padded row at a time is the same call)
                first, rows_sc = {}, []
                for i in range(bs):
                    n = len(prompts[i])
                    if multi:
                        tok = eng.mprefill(i, prompts[i])
                        last = (n - 1) % (64 * eng.max_blocks)
                        t = pick(ids0[i].tolist(), eng.mlogits()[last])
                    else:
                        tok = eng.bprefill(i, prompts[i])
                        t = pick(ids0[i].tolist(), eng.logits()[(n - 1) % 64])
                    first[i] = t if sequential else tok
                    rows_sc.append(picked[0].clone())
                if scores is not None:
                    prefill_scores = torch.cat(rows_sc, 0).to(out_device)
            else:
                first = eng.mprefill_many(prompts) if multi else eng.bprefill_many(prompts)
            next_token_list = [[first[i]] for i in range(bs)]
            dmode = decoding_kwargs.get('decoding_mode', 'hier')
            # decoding_kwargs['device_trie']: True / False, or absent = AUTO (round 6): the on-GPU trie chained in front of the verify pass
            # from 8 sequences per GPU on, where one launch of the workgroup-per-query kernel (~120 us whatever the batch) beats the
            # host's per-sample queries + input upload (Mistral-7B bs=8: 9.58 vs 9.62 ms per step, profiles/r06_device_trie_wg.txt)
            chain_ok = multi and not sequential and streamer is None and \
                bool(decoding_kwargs.get('per_sample_budget', False)) and dmode.split('_')[0] in ('hier', 'one') and \
                int(decoding_length) <= _lib.LA_TREE_MAX and \
                not decoding_kwargs.get('debug_lookahead', False)
            if decoding_kwargs.get('device_trie', None) is None:
                decoding_kwargs['device_trie'] = bool(chain_ok and bs >= 8 and gather is None and dmode.split('_')[0] == 'hier')
            chained = bool(decoding_kwargs['device_trie']) and chain_ok
            # device_trie_update (default on with the chained device trie): the trie UPDATE of every step runs on the device as well
            dev_put = chained and bool(decoding_kwargs.get('device_trie_update', True)) and branch_length + 1 <= 64
            put_on_device, buffers_loaded = False, False
            replay_due, full_image_due, replay_calls = None, False, 1      # puts of the last chained step the host trie has not repeated yet
            decoding_kwargs['dls'].extend([1] * bs)
            decoding_kwargs['edls'].extend([1] * bs)
            max_cur = 0
            while True:
                for k, b in enumerate(batch_indices):
                    rows[b].extend(next_token_list[k])
                if streamer is not None:
                    streamer.put(np.array(next_token_list[0]))
                if put_on_device:
                    # the device inserted these tokens into its trie image itself, straight from the step's output block
                    # (la_trie_stream_put_dev behind the verify pass): the host trie repeats the same puts in the same order and
                    # drops the words it logged for them — nothing of the update crosses PCIe.  Round 4: the replay is deferred until
                    # the NEXT step's kernels are queued (the device image needs nothing from the host), so it overlaps the GPU step
                    replay_due = [(b, next_token_list[k]) for k, b in enumerate(batch_indices)]
                    put_on_device = False
                elif gather is None:                                                # :1254-1259, one native call for the batch
                    puts = [(b, [x for x in next_token_list[k] if x != -1]) for k, b in enumerate(batch_indices)]
                    if overlap_put:
                        run_deferred_put()                                          # (a step that found no overlap point)
                        deferred_put[0] = puts
                    else:
                        self.lookahead_cache.stream_put_many(puts, branch_length=branch_length + 1, final=False)
                max_cur = max(max_cur, max(len(rows[b]) - 1 for b in batch_indices))
                keep = []
                for k, b in enumerate(batch_indices):                               # :1269-1276 + _early_stop :937-980
                    if len(rows[b]) >= stop_max_length or any(t in eos_set for t in next_token_list[k]) or \
                            (custom_stop is not None and custom_stop(rows[b], out_device, scores)):        # :1284
                        finished_rows[b] = list(rows[b])
                    else:
                        keep.append(b)
                if gather is not None:
                    # one collective per loop iteration on every rank: this rank's lists in local row order ([] for retired rows), DONE once
                    # no row is left; strict = gathered and applied now, split-phase = applied under the next verify pass (overlap below)
                    mine = [[] for _ in range(bs)]
                    for k, b in enumerate(batch_indices):
                        mine[b] = [x for x in next_token_list[k] if x != -1]
                    gather.step_update(self.lookahead_cache, mine, branch_length, done=not keep)
                batch_indices = keep
                if scores is not None:
                    scores += (prefill_scores,)
                te = time.time()
                decoding_kwargs['fts'].append(te - ts)
                ts = te
                if not batch_indices:
                    break
                if chained:
                    dt0 = self._device_trie(decoding_kwargs['_n_samples'], dev_put, narrow=True)
                    if replay_due is not None and (len(batch_indices) > eng.max_blocks or full_image_due):
                        # several engine passes per step, or the host image outgrew the device's: replay first, then a synced query
                        dt0.replay(replay_due, branch_length + 1, calls=replay_calls)
                        replay_due, full_image_due = None, False
                    # device trie chained in front of the verify pass (la_llama_mstep_trie): ONE query launch for all active samples on the
                    # engine's stream, the step input assembled on the device from its outputs, one 64-row block per sample — no draft
                    # crosses PCIe in either direction; the host reads back the accepted tokens and the draft lengths only.  Same budget
                    # rule as the host path with per_sample_budget (lookahead_cache.py:534-541).
                    ts_q = time.time()
                    qids = [rows[b][-2:] for b in batch_indices]
                    per = min(decoding_length, _lib.LA_TREE_MAX)
                    dm = decoding_kwargs.get('decoding_mode', 'hier')
                    mode_q = (dm if '_' in dm else dm + '_mix').split('_')[1]
                    dt = self._device_trie(decoding_kwargs['_n_samples'], dev_put, narrow=True)
                    with torch.cuda.stream(eng.stream):
                        if dm.split('_')[0] == 'one':
                            dt.one_get_dev(qids, idxs=batch_indices, decoding_length=per, branch_length=branch_length, mode=mode_q,
                                           sync=replay_due is None)
                        else:
                            dt.hier_get_dev(qids, idxs=batch_indices, decoding_length=per, branch_length=branch_length, min_input_size=0,
                                            min_output_size=max(per // 2, 1), mode=mode_q, sync=replay_due is None)
                        if dev_put and not buffers_loaded:
                            dt.load_stream_buffers()        # the hold-back buffers as the host's stream_put calls left them
                            buffers_loaded = True
                        emitted, widths = {}, []
                        for g0 in range(0, len(batch_indices), eng.max_blocks):
                            grp = batch_indices[g0:g0 + eng.max_blocks]
                            eng.mstep_trie_async(dt, g0, grp, [stop_max_length - (len(rows[b]) - 1) - 1 for b in grp],
                                                 [rows[b][-1] for b in grp], put_idxs=grp if dev_put else None,
                                                 put_branch_length=branch_length + 1)
                            if replay_due is not None:         # the previous step's update, on the host trie, while the GPU verifies
                                full_image_due = not dt.replay(replay_due, branch_length + 1, calls=replay_calls)
                                replay_due = None
                            toks, Ts = eng.mstep_trie_finish()
                            for b, tk in zip(grp, toks):
                                emitted[b] = tk
                            widths.extend(Ts)
                        put_on_device = dev_put
                        replay_calls = (len(batch_indices) + eng.max_blocks - 1) // eng.max_blocks       # stream_put_dev calls of this step
                    decoding_kwargs['qts'].append(time.time() - ts_q)
                    decoding_kwargs.update({'decoding_qids': qids, 'decoding_ids': None, 'hit_sizes': None, 'batch_indices': batch_indices})
                    width = max(widths)
                    next_token_list = [emitted[b] for b in batch_indices]
                    for k in range(len(batch_indices)):
                        decoding_kwargs['dls'].append(width)
                        decoding_kwargs['edls'].append(len(next_token_list[k]))
                    continue
                drafts = self.lookahead_prepare_inputs_for_generation([rows[b] for b in batch_indices], [gi(b) for b in batch_indices],
                                                                      decoding_kwargs)
                segments = []
                for b, (d_ids, d_rm) in zip(batch_indices, drafts):
                    if len(d_ids) == 0:
                        d_ids, d_rm = np.asarray(rows[b][-1:], dtype=np.int32), _ONE
                    cur = len(rows[b]) - 1
                    segments.append((b, d_ids, d_rm, 2 if sequential else 0, stop_max_length - cur - 1))
                if sum(len(sg[1]) for sg in segments) <= _lib.LA_TREE_MAX and all(np.ndim(sg[2]) == 1 for sg in segments):
                    run_deferred_put()                                     # (bstep has no asynchronous form: same order, no overlap)
                    if gather is not None:
                        gather.overlap(self.lookahead_cache, branch_length)
                    emitted = eng.bstep(segments)                          # the whole batch shares one 64-row block
                    if sequential:
                        base, logits = eng.bstep_rows(), eng.logits()
                        kept = {}
                        for sg in segments:
                            emitted[sg[0]], kept[sg[0]] = self._sequential_walk(rows[sg[0]], sg, logits, base[sg[0]], pick)
                        eng.bcommit(kept)
                else:
                    assert multi, 'more than 64 draft rows per step need an engine created with max_blocks > 1'
                    emitted = {}
                    # ceil(T / 64) blocks per sample (1 for the usual 64-row trees), as many samples per pass as max_blocks holds
                    groups, cur_g, cur_b = [], [], 0
                    for sg in segments:
                        nb_ = (len(sg[1]) + 63) // 64
                        assert nb_ <= eng.max_blocks, f'a {len(sg[1])}-row tree needs an engine created with max_blocks >= {nb_}'
                        if cur_b + nb_ > eng.max_blocks:
                            groups.append(cur_g)
                            cur_g, cur_b = [], 0
                        cur_g.append(sg)
                        cur_b += nb_
                    groups.append(cur_g)
                    for group in groups:
                        wide_pass = any(len(sg[1]) > 64 or np.ndim(sg[2]) == 2 for sg in group)
                        if wide_pass or not hasattr(eng, 'mstep_async'):
                            run_deferred_put()
                            if gather is not None:
                                gather.overlap(self.lookahead_cache, branch_length)
                            out = eng.mstep_trees(group) if wide_pass else eng.mstep(group)
                        else:
                            # queue the pass, then do the host work nothing on the device waits for — the deferred trie update / the
                            # previous step's gather and its puts — while the GPU verifies
                            eng.mstep_async(group)
                            run_deferred_put()
                            if gather is not None:
                                gather.overlap(self.lookahead_cache, branch_length)
                            out = eng.mstep_finish()
                        if sequential:
                            logits, kept, base = eng.mlogits(), [], 0
                            for sg in group:
                                toks, keep_rows = self._sequential_walk(rows[sg[0]], sg, logits, base, pick)
                                emitted[sg[0]] = toks
                                kept.append(keep_rows)
                                base += 64 * ((len(sg[1]) + 63) // 64)
                            if wide_pass:
                                eng.mcommit_trees(kept, [len(sg[1]) for sg in group])
                            else:
                                eng.mcommit(kept)
                            continue
                        for sg, toks in zip(group, out):
                            emitted[sg[0]] = toks
                width = max(len(sg[1]) for sg in segments)
                next_token_list = [emitted[b] for b in batch_indices]
                for k in range(len(batch_indices)):
                    decoding_kwargs['dls'].append(width)
                    decoding_kwargs['edls'].append(len(next_token_list[k]))
            if replay_due is not None:                                              # the last step's update (the loop ended before another launch)
                self._device_trie(decoding_kwargs['_n_samples'], dev_put, narrow=True).replay(replay_due, branch_length + 1, calls=replay_calls)
                replay_due = None
            run_deferred_put()
            if gather is not None:
                gather.drain(self.lookahead_cache, branch_length)                   # until every rank has finished its sequences
                gather.flush(self.lookahead_cache, branch_length)                   # :1288-1290 for all B sequences, batch-index order
            else:
                for i in range(bs):                                                 # :1288-1290
                    self.lookahead_cache.stream_put([], branch_length=branch_length + 1, final=True, mode='output', idx=i)
        except BaseException:
            # A failure (or an interrupt) mid-request must not leave the trie with this request's stream buffers / input frequencies, and in a
            # sharded job must not strand the other ranks: they are (or will be) waiting in the per-step collective.  The failing rank keeps
            # its side of the protocol — DONE | FAILED contributions until every rank has finished, then the flush of all B sequences
            # (AcceptedTokenGather.abort_request) — and only then re-raises.
            try:
                run_deferred_put()
            except Exception:       # noqa: BLE001 — the original exception is the one to report
                pass
            if gather is not None:
                gather.abort_request(self.lookahead_cache, branch_length)
            else:
                for i in range(bs):
                    self.lookahead_cache.stream_put([], branch_length=branch_length + 1, final=True, mode='output', idx=i)
            raise
        if streamer is not None:
            streamer.end()
        seqs = np.full((bs, max_cur + 1), pad, dtype=np.int64)
        for b in range(bs):
            r = finished_rows[b][:max_cur + 1]
            seqs[b, :len(r)] = r
        sequences = torch.from_numpy(seqs).to(out_device)
        if return_dict_in_generate:
            kwargs = {k: decoding_kwargs[k] for k in ('dls', 'edls', 'fts', 'qts')}
            return LookaheadDecoderOnlyOutput(sequences=sequences, scores=scores, attentions=None, hidden_states=None,
                                              kwargs=kwargs)
        return sequences

    @staticmethod
    def _sequential_walk(row_ids, segment, logits, base, pick):
        """pretrained_model_batch.py:829-886 for one sample: starting at the root, pick the next token from the current tree row's
        logits with the processors seeing the padded row plus the tokens accepted so far, then follow the child carrying that
        token (first such row in DFS order).  -> (emitted tokens, kept tree rows, root first)."""
        _, d_ids, d_rm, _, limit = segment
        T = len(d_ids)
        parent = [-1] * T
        wide = np.ndim(d_rm) == 2
        for j in range(1, T):
            if wide:                                   # multi-word row mask: word w = tree columns 64 w .. 64 w + 63
                full = 0
                for w in range(d_rm.shape[1]):
                    full |= int(d_rm[j][w]) << (64 * w)
            else:
                full = int(d_rm[j])
            below = full & ((1 << j) - 1)
            parent[j] = below.bit_length() - 1
        limit = max(1, min(int(limit), 32))
        cur, kept, toks = 0, [0], []
        while True:
            t = pick(row_ids + toks, logits[base + cur])
            toks.append(t)
            if len(toks) >= limit:
                break
            nxt = next((j for j in range(1, T) if parent[j] == cur and int(d_ids[j]) == t), None)
            if nxt is None:
                break
            cur = nxt
            kept.append(cur)
        return toks, kept

    @torch.no_grad()
    def greedy_search(self, input_ids, max_length, attention_mask=None, eos_token_id=None, pad_token_id=0,
                      logits_processor=None, do_sample=False, stopping_criteria=None):
        """Plain decoding of the whole batch through the same engine (one row per sample per block).  With a processor list or
        sampling every token is picked on the host from the sample's logits row (forward-only step + la_llama_bcommit).
        `stopping_criteria`: the caller's criteria besides the length bound (MaxTimeCriteria, custom ones) are evaluated per
        row after every token, on the row's own tokens — a row they stop is retired like one that met eos."""
        ids0 = input_ids.cpu().numpy().astype(np.int64)
        bs, P = ids0.shape
        am = np.ones_like(ids0) if attention_mask is None else attention_mask.cpu().numpy().astype(np.int64)
        eos = set([eos_token_id] if isinstance(eos_token_id, int) else (eos_token_id or []))
        eng = self.engine
        eng.reset_slot(-1)
        prompts = {i: ids0[i][am[i] == 1].tolist() for i in range(bs)}
        host_pick = do_sample or (logits_processor is not None and len(logits_processor) > 0)
        multi = bool(getattr(eng, 'max_blocks', 0))

        def pick(ctx_ids, row):
            scores = row[None]
            if logits_processor is not None and len(logits_processor) > 0:
                scores = logits_processor(torch.tensor([ctx_ids], dtype=torch.long, device=eng.device), scores.clone())
            if do_sample:
                return int(torch.multinomial(torch.softmax(scores.float(), dim=-1), num_samples=1)[0, 0])
            return int(torch.argmax(scores, dim=-1)[0])

        if host_pick:
            first = {}
            for i in range(bs):
                n = len(prompts[i])
                if multi:
                    eng.mprefill(i, prompts[i])
                    first[i] = pick(ids0[i].tolist(), eng.mlogits()[(n - 1) % (64 * eng.max_blocks)])
                else:
                    eng.bprefill(i, prompts[i])
                    first[i] = pick(ids0[i].tolist(), eng.logits()[(n - 1) % 64])
        else:
            first = eng.mprefill_many(prompts) if multi else eng.bprefill_many(prompts)
        rows = [ids0[i].tolist() + [first[i]] for i in range(bs)]
        custom_stop = _custom_stop(stopping_criteria)

        def alive(b):
            return len(rows[b]) < max_length and rows[b][-1] not in eos and \
                not (custom_stop is not None and custom_stop(rows[b], input_ids.device))

        live = [b for b in range(bs) if alive(b)]
        while live:
            out = eng.bstep([(b, np.asarray(rows[b][-1:], dtype=np.int32), _ONE, 2 if host_pick else 0, 1) for b in live])
            if host_pick:
                base, logits = eng.bstep_rows(), eng.logits()
                for b in live:
                    rows[b].append(pick(rows[b], logits[base[b]]))
                eng.bcommit({b: [0] for b in live})
            else:
                for b in live:
                    rows[b].append(out[b][0])
            live = [b for b in live if alive(b)]
        L = max(len(r) for r in rows)
        seqs = np.full((bs, L), pad_token_id, dtype=np.int64)
        for b in range(bs):
            seqs[b, :len(rows[b])] = rows[b]
        return torch.from_numpy(seqs).to(input_ids.device)

    def generate(self, inputs=None, generation_config=None, logits_processor=None, stopping_criteria=None,
                 prefix_allowed_tokens_fn=None, synced_gpus=None, assistant_model=None, streamer=None, **kwargs):
        """generate() with the reference's signature (common/pretrained_model_batch.py generate, same front half as
        pretrained_model.py:110-372): see LookaheadPreTrainedModel.generate of pretrained_model.py here — one resolver
        (lookahead_generation_utils.resolve_generate_args) serves both.  A non-empty processor list / do_sample take the
        sequential accept path (host-side token pick), everything else stays on the device."""
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
        if dk.get('use_lookahead', False) and dk.get('decoding_length', 64) > 1 and dk.get('branch_length', 12) > 0:
            dk['do_sample'] = ga.do_sample
            return self.lookahead_generation(input_ids, logits_processor=ga.logits_processor if len(ga.logits_processor) else None,
                                             stopping_criteria=ga.stopping_criteria, pad_token_id=ga.pad_token_id,
                                             eos_token_id=ga.eos_token_id, return_dict_in_generate=ga.return_dict_in_generate,
                                             streamer=streamer, attention_mask=attention_mask, decoding_kwargs=dk)
        from transformers import LogitsProcessorList
        procs = LogitsProcessorList(list(ga.logits_processor) + list(ga.logits_warper))
        out = self.greedy_search(input_ids, ga.max_length, attention_mask=attention_mask,
                                 eos_token_id=ga.eos_token_id if ga.eos_token_id is not None
                                 else getattr(self.generation_config, 'eos_token_id', None),
                                 pad_token_id=ga.pad_token_id if ga.pad_token_id is not None else 0,
                                 logits_processor=procs if len(procs) else None, do_sample=ga.do_sample,
                                 stopping_criteria=ga.stopping_criteria)
        return LookaheadDecoderOnlyOutput(sequences=out, kwargs={}) if ga.return_dict_in_generate else out
