# -*- coding: utf-8 -*-
"""ORACLE tooling (test infrastructure): the tiny seeded models, warm-up recipes and random trees the golden-vector generators and the
tests share.  Build-free on purpose — numpy and torch only, nothing from the product package — so that `make golden` runs in a fresh
checkout BEFORE csrc/build.sh (the product package refuses to import without its HIP library).  Weights are pure functions of a numpy
seed (MT19937: stable across platforms), so no weights are stored.  tests/tiny_model.py re-exports everything here."""
import numpy as np
import torch

TINY = dict(n_layers=2, hidden=256, n_heads=2, n_kv_heads=2, ffn=512, vocab=512, rms_eps=1e-5)


def tiny_weights(seed=0, dtype=torch.float32, std=0.08, cfg=None):
    rs = np.random.RandomState(seed)
    c = dict(TINY)
    if cfg:
        c.update(cfg)
    hd = 128 if cfg and cfg.get('head_dim') else c['hidden'] // c['n_heads']

    def w(n, k):
        return torch.from_numpy((rs.standard_normal((n, k)) * std).astype(np.float32)).to(dtype)

    def nw():
        return torch.from_numpy((1.0 + 0.1 * rs.standard_normal(c['hidden'])).astype(np.float32)).to(dtype)

    sd = {'model.embed_tokens.weight': w(c['vocab'], c['hidden'])}
    for i in range(c['n_layers']):
        p = f'model.layers.{i}.'
        sd[p + 'self_attn.q_proj.weight'] = w(c['n_heads'] * hd, c['hidden'])
        sd[p + 'self_attn.k_proj.weight'] = w(c['n_kv_heads'] * hd, c['hidden'])
        sd[p + 'self_attn.v_proj.weight'] = w(c['n_kv_heads'] * hd, c['hidden'])
        sd[p + 'self_attn.o_proj.weight'] = w(c['hidden'], c['n_heads'] * hd)
        sd[p + 'mlp.gate_proj.weight'] = w(c['ffn'], c['hidden'])
        sd[p + 'mlp.up_proj.weight'] = w(c['ffn'], c['hidden'])
        sd[p + 'mlp.down_proj.weight'] = w(c['hidden'], c['ffn'])
        sd[p + 'input_layernorm.weight'] = nw()
        sd[p + 'post_attention_layernorm.weight'] = nw()
    sd['model.norm.weight'] = nw()
    sd['lm_head.weight'] = w(c['vocab'], c['hidden'])
    return sd


# tiny Mixtral / Mistral (GQA, head_dim 128; Mixtral: 8 experts top-2, rope_theta 1e6) shared by oracle/gen_golden_moe.py
TINY_MOE = dict(n_layers=2, hidden=256, n_heads=2, n_kv_heads=1, ffn=512, vocab=512, rms_eps=1e-5, n_experts=8, top_k=2,
                rope_theta=1e6)
TINY_GQA = dict(n_layers=2, hidden=512, n_heads=4, n_kv_heads=2, ffn=512, vocab=512, rms_eps=1e-5, n_experts=0, top_k=2,
                rope_theta=10000.0)




def moe_weights(cfg, seed=0, dtype=torch.float32, std=0.06, router_std=0.2):
    """HF Mixtral / Mistral-named state dict from numpy's MT19937."""
    rs = np.random.RandomState(seed)
    hd = cfg['hidden'] // cfg['n_heads']

    def w(n, k, s=std):
        return torch.from_numpy((rs.standard_normal((n, k)) * s).astype(np.float32)).to(dtype)

    def nw():
        return torch.from_numpy((1.0 + 0.1 * rs.standard_normal(cfg['hidden'])).astype(np.float32)).to(dtype)

    sd = {'model.embed_tokens.weight': w(cfg['vocab'], cfg['hidden'])}
    for i in range(cfg['n_layers']):
        p = f'model.layers.{i}.'
        sd[p + 'self_attn.q_proj.weight'] = w(cfg['n_heads'] * hd, cfg['hidden'])
        sd[p + 'self_attn.k_proj.weight'] = w(cfg['n_kv_heads'] * hd, cfg['hidden'])
        sd[p + 'self_attn.v_proj.weight'] = w(cfg['n_kv_heads'] * hd, cfg['hidden'])
        sd[p + 'self_attn.o_proj.weight'] = w(cfg['hidden'], cfg['n_heads'] * hd)
        if cfg['n_experts'] > 0:
            sd[p + 'block_sparse_moe.gate.weight'] = w(cfg['n_experts'], cfg['hidden'], router_std)
            for e in range(cfg['n_experts']):
                q = p + f'block_sparse_moe.experts.{e}.'
                sd[q + 'w1.weight'] = w(cfg['ffn'], cfg['hidden'])
                sd[q + 'w2.weight'] = w(cfg['hidden'], cfg['ffn'])
                sd[q + 'w3.weight'] = w(cfg['ffn'], cfg['hidden'])
        else:
            sd[p + 'mlp.gate_proj.weight'] = w(cfg['ffn'], cfg['hidden'])
            sd[p + 'mlp.up_proj.weight'] = w(cfg['ffn'], cfg['hidden'])
            sd[p + 'mlp.down_proj.weight'] = w(cfg['hidden'], cfg['ffn'])
        sd[p + 'input_layernorm.weight'] = nw()
        sd[p + 'post_attention_layernorm.weight'] = nw()
    sd['model.norm.weight'] = nw()
    sd['lm_head.weight'] = w(cfg['vocab'], cfg['hidden'])
    return sd


def tiny_decisive_weights(seed=0, dtype=torch.float32, cfg=None):
    """The tiny Llama as a "permutation LM" (same recipe as llama_engine.random_weights(decisive=True), but a pure function of a
    numpy seed): o_proj / down_proj std 1e-4 keep the residual stream on the token embedding and lm_head[pi(t)] = embed[t] for one
    cycle pi over [3, V), so the greedy continuation has margins of ~15 sigma — identical tokens in fp32 and bf16, on the
    reference's CPU path and on the MI355X engine.  Used by oracle/gen_golden_noisy.py and the partial-accept parity tests."""
    sd = tiny_weights(seed, torch.float32, cfg=cfg)
    rs = np.random.RandomState(seed + 1000)
    c = dict(TINY)
    if cfg:
        c.update(cfg)
    for i in range(c['n_layers']):
        p = f'model.layers.{i}.'
        for name in ('self_attn.o_proj.weight', 'mlp.down_proj.weight'):
            sd[p + name] = sd[p + name] * (1e-4 / 0.08)
    V = c['vocab']
    order = 3 + rs.permutation(V - 3)
    perm = np.arange(V)
    perm[order] = np.roll(order, -1)
    head = torch.empty_like(sd['model.embed_tokens.weight'])
    head[torch.from_numpy(perm)] = sd['model.embed_tokens.weight']
    sd['lm_head.weight'] = head
    return {k: v.to(dtype) for k, v in sd.items()}


def noisy_copies(truth, n_copies, rho, vocab, seed):
    """bench.py's trie warm-up: n_copies of the continuation with each token replaced with probability rho."""
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(n_copies):
        t = np.array(truth)
        hit = rs.rand(len(t)) < rho
        t[hit] = rs.randint(3, vocab, size=int(hit.sum()))
        out.append(t.tolist())
    return out


def random_tree(rs, T, branch=0.35):
    """Random DFS-ordered tree of T rows -> (parent list, uint64 row masks)."""
    parent = [-1]
    rows = [1]
    stack = [0]
    for i in range(1, T):
        while len(stack) > 1 and rs.rand() < branch:
            stack.pop()
        p = stack[-1]
        parent.append(p)
        rows.append(rows[p] | (1 << i))
        stack.append(i)
    return parent, np.array(rows, dtype=np.uint64)
