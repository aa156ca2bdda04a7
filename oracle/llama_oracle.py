# -*- coding: utf-8 -*-
"""ORACLE (test infrastructure, not product code): CPU restatement of the verify forward, the accept
scan and the bs=1 lookahead loop of alipay/PainlessInferenceAcceleration.

  forward      lookahead/lookahead/models/llama/modeling_llama.py:76-90 (RMSNorm), 93-169 (RoPE),
               172-186 (MLP), 189-308 (attention), 311-376 (layer), 544-677 (model + rank-4 mask hook),
               769 (lm_head)
  accept scan  lookahead/lookahead/common/pretrained_model.py:764-892
  KV keep set  pretrained_model.py:865-875, 894-907
  loop         pretrained_model.py:947-1268 (+ 666-756 draft retrieval)
  GQA / MoE    models/mistral/modeling_mistral.py:236-318 (repeat_kv attention); models/mixtral/modeling_mixtral.py:
               668-759 (expert MLP + router)
  batch twin   models/llama/modeling_llama_batch.py:121-138, 190-201, 297-323, 340-420, 913-915 (forward);
               common/pretrained_model_batch.py:664-759, 767-935, 937-999, 1002-1330 (loop)

Plain torch on CPU in the dtype of the weights (fp32 or bf16), same operation order and the same
rounding points as the reference (every nn.Linear / elementwise result is materialised in the
weight dtype).  Pinning: oracle/gen_golden_model.py runs the reference classes (imported from
/root/reference) on a tiny seeded Llama and commits tokens / dls / edls / per-step argmax rows /
logits samples to tests/golden/llama_tiny_*.npz + accept_scan.json; tests/test_oracle_llama.py
checks this file against them.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math
import time

import numpy as np
import torch


# ------------------------------------------------------------------------------------------ forward
def _rms(x, w, eps, cast_first=False):
    """LlamaRMSNorm (models/llama/modeling_llama.py:86-90): one rounding, (w * (x * rsqrt)).to(dtype).  cast_first:
    Mistral/MixtralRMSNorm (models/mixtral/modeling_mixtral.py:160-165): w * (x_fp32 * rsqrt).to(dtype), two roundings."""
    var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
    if cast_first:
        return w * (x.to(torch.float32) * torch.rsqrt(var + eps)).to(x.dtype)
    return (w * (x * torch.rsqrt(var + eps))).to(x.dtype)


def _rope_cos_sin(pos, head_dim, theta, dtype):
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    ang = (inv[:, None].float() @ pos[None, :].float()).transpose(0, 1)
    emb = torch.cat((ang, ang), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def _rot_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


class OracleLlama(object):
    """Functional Llama over a HF-named state dict.  KV cache: list of (k, v), each [n_kv, C, hd]."""

    def __init__(self, shape, state_dict):
        self.s = shape
        self.w = state_dict
        self.dtype = state_dict['lm_head.weight'].dtype

    @torch.no_grad()
    def forward(self, ids, mask, past, forced_routing=None):
        """ids: LongTensor [T]; mask: 0/1 LongTensor [T, C+T] (the rank-4 mask without its two unit dims);
        past: None or list of (k, v).  -> logits [T, V], new past.  forced_routing (MoE, tests only): per layer a [T, E]
        tensor of routing weights (0 = not routed) used INSTEAD of the router's own top-k, so that the continuous part of
        a device run can be compared row by row although top-k selection is discontinuous."""
        s, w, dt = self.s, self.w, self.dtype
        T, hd, nh, nkv = ids.shape[0], s.head_dim, s.n_heads, s.n_kv_heads
        lin = torch.nn.functional.linear
        # batch dim of 1 kept so that the CPU kernels see the reference's tensor shapes ([1,T,*], [1,H,T,hd])
        h = w['model.embed_tokens.weight'][ids][None]
        mask4 = mask[None, None]
        pos = torch.sum(mask4, dim=-1).squeeze(1) - 1                          # model hook, :586
        win = int(getattr(s, 'sliding_window', 0) or 0)
        if win > 0:
            # EXTENSION (not on the reference's lookahead path): the transformers sliding-window rule
            # (modeling_attn_mask_utils._make_causal_mask: masked iff pos_row - pos_key > window) on the committed keys
            C = mask.shape[1] - T
            keypos = torch.arange(C)
            mask4 = mask4.clone()
            mask4[0, 0, :, :C] &= ((pos[0][:, None] - keypos[None, :]) <= win).long()
        bias = (1.0 - mask4.to(dt)) * torch.finfo(dt).min                      # :587
        cos, sin = _rope_cos_sin(pos[0], hd, s.rope_theta, dt)
        cos, sin = cos[None, None], sin[None, None]
        new_past = []
        self.router_trace = []
        self.hidden_trace = []
        cf = bool(getattr(s, 'norm_cast_first', False))
        for i in range(s.n_layers):
            p = f'model.layers.{i}.'
            x = _rms(h, w[p + 'input_layernorm.weight'], s.rms_eps, cf)
            q = lin(x, w[p + 'self_attn.q_proj.weight']).view(1, T, nh, hd).transpose(1, 2)
            k = lin(x, w[p + 'self_attn.k_proj.weight']).view(1, T, nkv, hd).transpose(1, 2)
            v = lin(x, w[p + 'self_attn.v_proj.weight']).view(1, T, nkv, hd).transpose(1, 2)
            q = (q * cos) + (_rot_half(q) * sin)
            k = (k * cos) + (_rot_half(k) * sin)
            if past is not None:
                k = torch.cat([past[i][0][None], k], dim=2)
                v = torch.cat([past[i][1][None], v], dim=2)
            new_past.append((k[0], v[0]))
            kk, vv = k, v
            if nkv != nh:                                                      # repeat_kv (mistral/modeling_mistral.py:236-318)
                rep = nh // nkv
                kk = k[:, :, None].expand(1, nkv, rep, k.shape[2], hd).reshape(1, nh, -1, hd)
                vv = v[:, :, None].expand(1, nkv, rep, v.shape[2], hd).reshape(1, nh, -1, hd)
            att = torch.matmul(q, kk.transpose(2, 3)) / math.sqrt(hd)
            att = att + bias
            att = torch.max(att, torch.tensor(torch.finfo(att.dtype).min))
            att = torch.softmax(att, dim=-1, dtype=torch.float32).to(dt)
            o = torch.matmul(att, vv).transpose(1, 2).reshape(1, T, nh * hd)
            h = h + lin(o, w[p + 'self_attn.o_proj.weight'])
            x = _rms(h, w[p + 'post_attention_layernorm.weight'], s.rms_eps, cf)
            if getattr(s, 'n_experts', 0) > 0:
                h = h + self._moe(x, p, None if forced_routing is None else forced_routing[i])
            else:
                g = torch.nn.functional.silu(lin(x, w[p + 'mlp.gate_proj.weight']))
                u = lin(x, w[p + 'mlp.up_proj.weight'])
                h = h + lin(g * u, w[p + 'mlp.down_proj.weight'])
            if getattr(self, 'trace_hidden', False):                           # tests: residual stream after every layer
                self.hidden_trace.append(h[0].clone())
        h = _rms(h, w['model.norm.weight'], s.rms_eps, cf)
        return lin(h, w['lm_head.weight'])[0], new_past

    def _moe(self, x, p, forced=None):
        """MixtralSparseMoeBlock.forward (mixtral/modeling_mixtral.py:717-759): router logits in the activation dtype,
        softmax in fp32, top-k, renormalise, cast back; experts visited in index order, each adding
        w2(silu(w1 x) * w3 x) * routing_weight for its rows into a zero buffer of the activation dtype (index_add_)."""
        s, w = self.s, self.w
        lin = torch.nn.functional.linear
        xs = x.reshape(-1, x.shape[-1])
        logits = lin(xs, w[p + 'block_sparse_moe.gate.weight'])
        rw = torch.softmax(logits, dim=1, dtype=torch.float)
        rw, sel = torch.topk(rw, s.top_k, dim=-1)
        rw = (rw / rw.sum(dim=-1, keepdim=True)).to(xs.dtype)
        if forced is not None:
            order = torch.argsort((forced != 0).to(torch.int8), dim=-1, descending=True, stable=True)[:, :s.top_k]
            sel = order
            rw = torch.gather(forced, 1, order).to(xs.dtype)
        out = torch.zeros_like(xs)
        self.last_router_logits = logits
        self.router_trace.append(logits)
        for e in range(s.n_experts):
            slot, rows = torch.where((sel == e).t())
            if rows.shape[0] == 0:
                continue
            q = p + f'block_sparse_moe.experts.{e}.'
            cur = xs[rows]
            y = lin(torch.nn.functional.silu(lin(cur, w[q + 'w1.weight'])) * lin(cur, w[q + 'w3.weight']), w[q + 'w2.weight'])
            out.index_add_(0, rows, (rw[rows, slot, None] * y).to(xs.dtype))
        return out.reshape(x.shape)


# -------------------------------------------------------------------------------------- accept scan
def parents_from_mask(mask):
    """Row i's parent = the highest set column below i (rows are DFS-ordered, lookahead_cache.py:278-283)."""
    T = mask.shape[0]
    par = [-1] * T
    for i in range(1, T):
        cols = np.nonzero(mask[i, :i])[0]
        par[i] = int(cols[-1]) if len(cols) else -1
    return par


def _live_children(ids, par, live, want):
    """Rows whose parent is a live row and whose draft token is `want`, ascending.  The reference keeps EVERY leaf branch whose
    token at the current depth equals the picked token (pretrained_model.py:850-860) and reads the next logits row from the
    first of them (mask_indices[0], :831).  In a hier tree the survivors share one node per depth, so this is "the child of
    the current row"; in a par layout (lookahead_cache.py:441-488) a shared prefix is duplicated across chains, all copies
    stay live, and the walk may continue on a later chain when the first one stops matching."""
    return [j for j in range(1, len(ids)) if par[j] in live and int(ids[j]) == want]


def accept_scan(ids, mask, argmax_rows):
    """ids: list[T] (ids[0] = root), mask: [T,T] 0/1, argmax_rows[t] = greedy token after tree row t.
    -> (next_token_list, logit_indices).  Restates pretrained_model.py:806-864: starting at the root, follow
    the rows whose draft token equals the current row's argmax (first such row in DFS order supplies the next logits) until
    none does; the emitted tokens are the argmax of every visited row (matches + 1 bonus)."""
    par = parents_from_mask(np.asarray(mask))
    cur, live, toks, rows = 0, {0}, [], [0]
    while True:
        want = int(argmax_rows[cur])
        toks.append(want)
        nxt = _live_children(ids, par, live, want)
        if not nxt:
            break
        cur, live = nxt[0], set(nxt)
        rows.append(cur)
    return toks, rows


def kv_keep_positions(context_length, n_draft, logit_indices):
    """Positions of the KV rows the reference keeps (pretrained_model.py:865-875, 894-907): the whole prefix
    including the root row, then one row per accepted draft token; nothing is dropped when every draft token was
    accepted."""
    m = len(logit_indices) - 1
    if n_draft == m:
        return list(range(context_length + n_draft))
    return list(range(context_length)) + [context_length - 1 + i for i in logit_indices[1:]]


# --------------------------------------------------------------------------------------------- loop
@torch.no_grad()
def accept_scan_sequential(ids, mask, logits, seq, logits_processor, limit=None):
    """The accept walk with a logits-processor list (pretrained_model.py:825-864): the processors see the sequence
    INCLUDING the tokens accepted so far in this step, so rows are evaluated one after another along the path.
    limit (batch twin, pretrained_model_batch.py:862): at most this many tokens are emitted."""
    par = parents_from_mask(np.asarray(mask))
    cur, live, toks, rows = 0, {0}, [], [0]
    while True:
        ctx = torch.tensor([list(seq) + toks], dtype=torch.long)
        want = int(torch.argmax(logits_processor(ctx, logits[cur][None].clone()), dim=-1)[0])
        toks.append(want)
        if limit is not None and len(toks) >= limit:
            break
        nxt = _live_children(ids, par, live, want)
        if not nxt:
            break
        cur, live = nxt[0], set(nxt)
        rows.append(cur)
    return toks, rows


@torch.no_grad()
def lookahead_generate(model, cache, prompt, max_length, eos_token_id=2, decoding_length=64, branch_length=12,
                       decoding_mode='hier', max_query_length=2, stop_words=None, max_steps=None, record=None,
                       logits_processor=None):
    """bs=1 lookahead_generation, greedy decoding; logits_processor: None / empty (row-parallel argmax) or a
    LogitsProcessorList (sequential walk).  -> dict(sequences, dls, edls, fts, qts).  `cache` is any object with the
    LookaheadCache surface."""
    eos = [eos_token_id] if isinstance(eos_token_id, int) else list(eos_token_id if eos_token_id is not None else [None])
    cache.eos_ids = eos
    cache.stop_words = stop_words if stop_words is not None else {}
    seq = [int(x) for x in prompt]
    dls, edls, fts, qts = [], [], [], []
    cache.put(seq[1:], branch_length=branch_length + 1, mode='input', idx=0)          # :1153-1156
    past = None
    ts = time.time()
    steps = 0
    while True:
        if past is None:
            P = len(seq)
            mask = torch.tril(torch.ones((P, P), dtype=torch.long))
            logits, past = model.forward(torch.tensor(seq, dtype=torch.long), mask, None)
            if logits_processor:
                toks = [int(torch.argmax(logits_processor(torch.tensor([seq]), logits[-1][None].clone()), dim=-1)[0])]
            else:
                toks = [int(torch.argmax(logits[-1]))]
            dls.append(1); edls.append(1)
            if record is not None:
                record.append({'ids': list(seq), 'argmax': [int(x) for x in torch.argmax(logits, -1)], 'next': toks})
        else:
            ubl = min(branch_length, max_length - len(seq) - 1)                      # :680
            assert ubl >= 0
            fmt, mode = (decoding_mode if '_' in decoding_mode else decoding_mode + '_mix').split('_')   # :712-714
            tq = time.time()
            d_ids, d_mask, sizes = getattr(cache, fmt + '_get')(seq[-max_query_length:], decoding_length=decoding_length,
                                                               branch_length=ubl, min_input_size=0,
                                                               min_output_size=max(decoding_length // 2, 1),
                                                               mode=mode, idx=0)
            qts.append(time.time() - tq)
            T, C = len(d_ids), len(seq) - 1
            d_mask = np.asarray(d_mask).astype(np.int64)
            full = torch.cat([torch.ones((T, C), dtype=torch.long), torch.from_numpy(d_mask)], dim=1)
            logits, past = model.forward(torch.tensor(d_ids, dtype=torch.long), full, past)
            am = [int(x) for x in torch.argmax(logits, -1)]
            if logits_processor:
                toks, rows = accept_scan_sequential(d_ids, d_mask, logits, seq, logits_processor)
            elif T == 1:
                toks, rows = [am[0]], [0]
            else:
                toks, rows = accept_scan(d_ids, d_mask, am)
            keep = kv_keep_positions(len(seq), T - 1, rows)
            if len(keep) != past[0][0].shape[1]:
                idx = torch.tensor(keep, dtype=torch.long)
                past = [(k[:, idx], v[:, idx]) for k, v in past]
            dls.append(T); edls.append(len(toks))
            if record is not None:
                record.append({'ids': list(d_ids), 'rows': [int(sum(int(b) << j for j, b in enumerate(r))) for r in d_mask],
                               'argmax': am, 'next': toks, 'accepted_rows': rows, 'sizes': list(sizes)})
        seq.extend(toks)
        cache.stream_put(toks, branch_length=branch_length + 1, final=False, mode='output', idx=0)   # :1203
        finished = len(seq) >= max_length or any(e in toks for e in eos)                         # :1225-1231
        steps += 1
        if max_steps is not None and steps >= max_steps:
            finished = True
        te = time.time(); fts.append(te - ts); ts = te
        if finished:
            cache.stream_put([], branch_length=branch_length + 1, final=True, mode='output', idx=0)
            break
    return {'sequences': seq, 'dls': dls, 'edls': edls, 'fts': fts, 'qts': qts}


@torch.no_grad()
def greedy_generate(model, prompt, n_new):
    """Plain greedy decoding with the same forward (lookahead output must equal this in fp32)."""
    seq = [int(x) for x in prompt]
    P = len(seq)
    logits, past = model.forward(torch.tensor(seq), torch.tril(torch.ones((P, P), dtype=torch.long)), None)
    for _ in range(n_new):
        t = int(torch.argmax(logits[-1]))
        seq.append(t)
        logits, past = model.forward(torch.tensor([t]), torch.ones((1, len(seq)), dtype=torch.long), past)
    return seq


# ================================================================================ batch (cursor) twin
class OracleLlamaBatch(OracleLlama):
    """The "fused" batch forward (models/llama/modeling_llama_batch.py): same network, different schedule —
    one QKV projection whose Q rows are pre-divided by sqrt(head_dim) (:340-346), RoPE as a gather from a table of
    2x2 rotations (:121-131, 190-201), scores = K.(Q)^T transposed back plus the additive mask (:311-316), softmax in
    the storage dtype (:318, NOT fp32), a pre-allocated KV buffer [B, H, decoding_max_length, hd] written in place at
    each sample's cursor (:384-400), lm_head on the last position only at prefill (:913-915)."""

    def __init__(self, shape, state_dict, max_pos=2048):
        super().__init__(shape, state_dict)
        assert shape.n_kv_heads == shape.n_heads, 'the reference batch model assumes MHA (modeling_llama_batch.py:342)'
        s, dt = shape, self.dtype
        inv = 1.0 / (s.rope_theta ** (torch.arange(0, s.head_dim, 2).float() / s.head_dim))
        fr = torch.einsum('i,j->ij', torch.arange(max_pos, dtype=inv.dtype), inv)
        c, sn = fr.cos(), fr.sin()
        self.rot = torch.stack([torch.stack([c, -sn], dim=1), torch.stack([sn, c], dim=1)], dim=1).to(dt)   # [pos,2,2,hd/2]
        coef = math.sqrt(s.head_dim)
        self.wqkv = [torch.cat([self.w[f'model.layers.{i}.self_attn.q_proj.weight'] / coef,
                                self.w[f'model.layers.{i}.self_attn.k_proj.weight'],
                                self.w[f'model.layers.{i}.self_attn.v_proj.weight']], dim=0) for i in range(s.n_layers)]

    def _rope(self, x, pos):
        c = self.rot[pos.unsqueeze(1)]                                   # [B,1,T,2,2,hd/2]
        shp = x.shape
        xv = x.reshape(*shp[:-1], 1, 2, shp[-1] // 2)
        return (c * xv).sum(-2).view(shp)

    @torch.no_grad()
    def forward_batch(self, ids, mask, past, cursors=None, decoding_max_length=None):
        """ids LongTensor [B,T]; mask 0/1 LongTensor [B,1,T,S]; past None (prefill) or list of [k,v] buffers
        [B,H,Lmax,hd] (updated in place); cursors: write position per sample.  -> logits [B,T|1,V], past."""
        s, w, dt = self.s, self.w, self.dtype
        B, T = ids.shape
        hd, nh = s.head_dim, s.n_heads
        lin = torch.nn.functional.linear
        pos = torch.sum(mask, dim=-1).squeeze(1) - 1                         # :731
        bias = (1.0 - mask.to(dt)) * torch.finfo(dt).min                      # :733
        h = w['model.embed_tokens.weight'][ids]
        prefill = past is None
        new_past = [] if prefill else past
        for i in range(s.n_layers):
            p = f'model.layers.{i}.'
            x = _rms(h, w[p + 'input_layernorm.weight'], s.rms_eps)
            mat = lin(x, self.wqkv[i]).view(B, T, 3, nh, hd).permute(0, 3, 2, 1, 4)
            q, k, v = mat.unbind(2)
            q, k = self._rope(q, pos), self._rope(k, pos)
            if prefill:
                keys, vals = k, v
                zeros = torch.zeros((B, nh, decoding_max_length - T, hd), dtype=dt)
                new_past.append([torch.cat([k, zeros], 2), torch.cat([v, zeros], 2)])
            else:
                pk, pv = past[i]
                for b, cur in enumerate(cursors):
                    pk[b, :, cur:cur + T] = k[b]
                    pv[b, :, cur:cur + T] = v[b]
                top = max(cursors) + T
                keys, vals = pk[:, :, :top], pv[:, :, :top]
            if B == 1:
                att = torch.baddbmm(bias.squeeze(0), q.squeeze(0), keys.squeeze(0).transpose(-1, -2))[None]
            else:
                att = torch.matmul(keys, q.transpose(-1, -2)).transpose(-1, -2)
                att = att.add_(bias)
            att = torch.softmax(att, dim=-1)
            o = torch.matmul(att, vals).permute(0, 2, 1, 3).contiguous().view(B, T, nh * hd)
            h = h + lin(o, w[p + 'self_attn.o_proj.weight'])
            x = _rms(h, w[p + 'post_attention_layernorm.weight'], s.rms_eps)
            g = torch.nn.functional.silu(lin(x, w[p + 'mlp.gate_proj.weight']))
            h = h + lin(g * lin(x, w[p + 'mlp.up_proj.weight']), w[p + 'mlp.down_proj.weight'])
        h = _rms(h, w['model.norm.weight'], s.rms_eps)
        if prefill:
            h = h[:, -1:]
        return lin(h, w['lm_head.weight']), new_past


def accept_scan_limited(ids, mask, argmax_rows, limit):
    """Batch variant of the accept scan (pretrained_model_batch.py:829-886): identical walk, but at most `limit`
    tokens are emitted (the loop bound min(max_branch_length, input_length - cur - 2) + 1, :862)."""
    toks, rows = accept_scan(ids, mask, argmax_rows)
    return toks[:limit], rows[:limit]


@torch.no_grad()
def lookahead_generate_batch(model, cache, input_ids, attention_mask, max_length, eos_token_id=2, pad_token_id=0,
                             decoding_length=64, branch_length=12, decoding_mode='hier', stop_words=None, record=None,
                             logits_processor=None):
    """bs>1 lookahead_generation (pretrained_model_batch.py:1002-1330), greedy decoding; logits_processor: None / empty
    (row-parallel argmax) or a LogitsProcessorList — then the prefill calls it batch-wise on the padded prompts (:783) and
    every accepted token is picked along the path with the processors seeing rows[b, :cur+i+2], pads included (:814-875).  input_ids / attention_mask: int arrays [B,P] (left padding allowed).  -> dict(sequences [B,L] array,
    dls, edls).  Per step: drafts from cache.bat_get with budget decoding_length // active (:713), one batched
    forward, a per-sample accept walk, in-place KV moves, per-sample stream_put, finished samples leave the batch."""
    ids0 = np.asarray(input_ids, dtype=np.int64)
    am = np.asarray(attention_mask, dtype=np.int64)
    B, P = ids0.shape
    eos = [eos_token_id] if isinstance(eos_token_id, int) else list(eos_token_id)
    cache.eos_ids = eos
    cache.stop_words = stop_words if stop_words is not None else {}
    L = max_length + decoding_length + 1                                         # decoding_max_length (:1168)
    col = np.concatenate([am, np.ones((B, L - P), dtype=np.int64)], axis=1)
    full = torch.tril(torch.from_numpy(col)[:, None, None].expand(-1, -1, L, -1), 0).contiguous()   # [B,1,L,L] (:1183)
    for i in range(B):
        cache.put(ids0[i, 1:-1].tolist(), branch_length=branch_length + 1, mode='input', idx=i)      # :1207-1209
    rows = np.concatenate([ids0, np.full((B, max_length - P), pad_token_id, dtype=np.int64)], axis=1)   # padded (:791-793)
    out_rows = rows.copy()
    dls, edls = [], []
    # prefill (:783-812)
    logits, past = model.forward_batch(torch.from_numpy(ids0), full[:, :, :P, :P], None, decoding_max_length=L)
    sequential = logits_processor is not None and len(logits_processor) > 0
    last = logits_processor(torch.from_numpy(ids0), logits[:, -1].clone()) if sequential else logits[:, -1]
    first = torch.argmax(last, dim=-1).tolist()
    active = list(range(B))                     # batch_indices
    cursors = [P] * B
    for b in range(B):
        rows[b, P] = first[b]
    dls += [1] * B
    edls += [1] * B
    emitted = [[t] for t in first]
    max_cur = 0
    fmt, mode = (decoding_mode if '_' in decoding_mode else decoding_mode + '_mix').split('_')
    while True:
        # trie update, stop checks, retire finished samples (:1247-1283, 937-980)
        for k, b in enumerate(active):
            cache.stream_put([t for t in emitted[k] if t != -1], branch_length=branch_length + 1, final=False,
                             mode='output', idx=b)
        max_cur = max(max_cur, max(cursors))
        keep = []
        for k, b in enumerate(active):
            done = cursors[k] + 1 >= max_length or any(e in emitted[k] for e in eos)
            if done:
                out_rows[b, :rows.shape[1]] = rows[k]
            else:
                keep.append(k)
        if len(keep) != len(active) and len(keep) > 0:
            rows = rows[keep]
            full = full[keep]
            past = [[kk[keep], vv[keep]] for kk, vv in past]
        cursors = [cursors[k] for k in keep]
        active = [active[k] for k in keep]
        if not active:
            break
        # drafts (:706-721) and the cursor-aligned mask (:727-731)
        n = len(active)
        qs = [[int(rows[k, c - 1]), int(rows[k, c])] for k, c in enumerate(cursors)]
        d_ids, d_masks, sizes = cache.bat_get(qs, decoding_length=max(decoding_length // n, 1), branch_length=branch_length,
                                              decoding_cursors=list(cursors), mode=mode, indices=list(active),
                                              decoding_mode=fmt)
        W = len(d_ids[0])
        lo = min(cursors)
        step_mask = torch.cat([full[:, :, lo:lo + W, :lo], torch.from_numpy(d_masks[:, None])], dim=-1)
        logits, past = model.forward_batch(torch.tensor(d_ids, dtype=torch.long), step_mask, past, cursors=cursors)
        am_rows = torch.argmax(logits, dim=-1).tolist()
        emitted = []
        for k in range(n):
            cur, off = cursors[k], cursors[k] - lo
            own = d_masks[k][:, off:off + W]
            T = int(sum(int(own[j, j]) for j in range(W)))      # real rows carry their own diagonal bit; pad rows do not
            limit = max_length - cur - 1                         # emitted tokens <= min(depth, input_length-cur-2)+1 (:862)
            if sequential:
                toks, acc = accept_scan_sequential(d_ids[k][:T], own[:T, :T], logits[k], rows[k, :cur + 1].tolist(),
                                                   logits_processor, limit)
            else:
                toks, acc = accept_scan_limited(d_ids[k][:T], own[:T, :T], am_rows[k], limit)
            m = len(toks) - 1
            rows[k, cur + 1:cur + 1 + len(toks)] = toks
            if acc[-1] != m:                                      # accepted rows are not already contiguous (:893-904)
                src = torch.tensor([cur + r for r in acc[1:]], dtype=torch.long)
                for kk, vv in past:
                    kk[k, :, cur + 1:cur + 1 + m] = kk[k][:, src]
                    vv[k, :, cur + 1:cur + 1 + m] = vv[k][:, src]
            dls.append(W)
            edls.append(len(toks))
            cursors[k] = cur + len(toks)
            emitted.append(toks)
        if record is not None:
            record.append({'cursors': [c - len(e) for c, e in zip(cursors, emitted)], 'bidx': list(active),
                           'ids': [list(x) for x in d_ids], 'next': [list(e) for e in emitted],
                           'logits': logits.float().numpy().copy()})
    for i in range(B):
        cache.stream_put([], branch_length=branch_length + 1, final=True, mode='output', idx=i)      # :1288-1290
    return {'sequences': out_rows[:, :max_cur + 1], 'dls': dls, 'edls': edls}
