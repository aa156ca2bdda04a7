# -*- coding: utf-8 -*-
"""-m gpu: Mistral (GQA) and Mixtral (sparse MoE) arithmetic through the engine against the REFERENCE's own logits
(tests/golden/moe_tiny_bf16.npz: cache-free reference forwards under a prompt+tree mask) and against the oracle.
Tolerance stated below (TOL_MOE, GAP)."""
import os

import numpy as np
import pytest
import torch

from oracle import llama_oracle as lo
from painlessinferenceacceleration_amd.llama_engine import LlamaVerifyEngine
from tests.tiny_model import GOLDEN, TINY_GQA, TINY_MOE, moe_shape, moe_weights

pytestmark = pytest.mark.gpu


TOL_MOE = 3e-2      # per row: max|logit_hip - logit_ref| <= 3e-2 * max|logit_ref| (tiny 2-layer models in the Mistral RMSNorm
                    # flavour sit at 2.0-2.2e-2 on a few rows; the 7B-shape tests keep 2e-2)
GAP = 0.05          # a row is "decisively routed" when p_k - p_(k+1) > GAP in every MoE layer (oracle probabilities)


def _check(got, ref, rows, what, tol=TOL_MOE):
    got, ref = got.float().cpu(), ref.float().cpu()
    bad = []
    for t in rows:
        bound = tol * float(ref[t].abs().max())
        err = float((got[t] - ref[t]).abs().max())
        if err > bound:
            bad.append((t, round(err, 3), round(bound, 3)))
    assert not bad, f'{what}: rows outside the tolerance: {bad}'


def _decisive_rows(oracle, n_rows, top_k):
    keep = []
    for t in range(n_rows):
        ok = True
        for rl in oracle.router_trace:
            p = torch.sort(torch.softmax(rl[t].float(), -1), descending=True).values
            ok = ok and float(p[top_k - 1] - p[top_k]) > GAP
        if ok:
            keep.append(t)
    return keep


def _rows_of(mask, P, T):
    return np.array([sum(int(mask[P + i, P + j]) << j for j in range(T)) for i in range(T)], dtype=np.uint64)


@pytest.mark.parametrize('balanced', [True, False])
def test_mistral_engine_matches_reference_logits(balanced):
    """Dense GQA model with the Mistral RMSNorm flavour against the reference MistralForCausalLM's logits."""
    g = np.load(os.path.join(GOLDEN, 'moe_tiny_bf16.npz'))
    shape = moe_shape(TINY_GQA)
    sd = {k: v.to(torch.bfloat16) for k, v in moe_weights(TINY_GQA, 0, torch.float32).items()}
    eng = LlamaVerifyEngine(shape, dict(sd), max_length=256, balanced=balanced)
    for case, (P, T) in enumerate([(24, 40), (3, 61), (50, 1)]):
        ids, mask, ref = g[f'mistral_{case}_ids'], g[f'mistral_{case}_mask'].astype(np.int64), g[f'mistral_{case}_logits']
        eng.reset()
        eng.prefill(ids[:P].tolist())
        _check(eng.logits()[:P], torch.from_numpy(ref[:P]), range(P), f'mistral case {case} prefill vs reference', 3e-2)
        eng.step(ids[P:].astype(np.int32), _rows_of(mask, P, T))
        _check(eng.logits()[:T], torch.from_numpy(ref[P:]), range(T), f'mistral case {case} tree vs reference', 3e-2)


def test_mixtral_engine_matches_reference_and_oracle():
    """Sparse MoE.  Top-k routing is discontinuous (measured: probabilities 0.264 vs 0.252 swap the 2nd expert under
    bf16-level differences of the router input), so parity is stated in three parts:
      (1) routing: on every decisively routed row (p_k - p_(k+1) > GAP in the oracle) the engine picks the oracle's
          experts in every layer, with weights within 5e-2 (softmax slope x bf16 noise of the router logits);
      (2) continuous part: ALL rows match the oracle re-run with the engine's own routing forced (TOL_MOE);
      (3) reference: rows whose attention span holds no row with a swapped expert match the REFERENCE
          MixtralForCausalLM logits and the oracle's own-routing logits (5e-2: routing WEIGHTS still differ at the 1e-2
          level, which the expert outputs amplify)."""
    g = np.load(os.path.join(GOLDEN, 'moe_tiny_bf16.npz'))
    shape = moe_shape(TINY_MOE)
    sd = {k: v.to(torch.bfloat16) for k, v in moe_weights(TINY_MOE, 0, torch.float32).items()}
    eng = LlamaVerifyEngine(shape, dict(sd), max_length=256)
    oracle = lo.OracleLlama(shape, sd)
    n_dec = n_all = n_clean = 0
    for case, (P, T) in enumerate([(24, 40), (3, 61), (50, 1)]):
        ids, mask, ref = g[f'mixtral_{case}_ids'], g[f'mixtral_{case}_mask'].astype(np.int64), g[f'mixtral_{case}_logits']
        eng.reset()
        eng.prefill(ids[:P].tolist())
        rw_pre = eng.route_weights()[:, :P].clone().cpu()
        lg_pre = eng.logits()[:P].clone()
        eng.step(ids[P:].astype(np.int32), _rows_of(mask, P, T))
        rw = torch.cat([rw_pre, eng.route_weights()[:, :T].cpu()], 1)            # [L, P+T, 8]
        got = torch.cat([lg_pre, eng.logits()[:T]], 0)
        lg_own, _ = oracle.forward(torch.from_numpy(ids), torch.from_numpy(mask), None)
        rows = _decisive_rows(oracle, P + T, shape.top_k)
        n_dec += len(rows); n_all += P + T
        for li, rl in enumerate(oracle.router_trace):                             # (1)
            p = torch.softmax(rl.float(), -1)
            v, sel = torch.topk(p, shape.top_k, -1)
            v = (v / v.sum(-1, keepdim=True)).to(torch.bfloat16).float()
            for t in rows:
                mine = sorted(int(e) for e in torch.nonzero(rw[li, t]).flatten())
                assert mine == sorted(sel[t].tolist()), (case, li, t, mine, sel[t].tolist())
                for k in range(shape.top_k):
                    assert abs(float(rw[li, t, sel[t, k]]) - float(v[t, k])) <= 5e-2, (case, li, t)
        flipped = set()
        for li, rl in enumerate(oracle.router_trace):
            sel = torch.topk(torch.softmax(rl.float(), -1), shape.top_k, -1).indices
            for t in range(P + T):
                if sorted(int(e) for e in torch.nonzero(rw[li, t]).flatten()) != sorted(sel[t].tolist()):
                    flipped.add(t)
        assert not (flipped & set(rows))
        clean = [t for t in range(P + T) if not any(mask[t, j] and j in flipped for j in range(P + T))]
        n_clean += len(clean)
        lg_forced, _ = oracle.forward(torch.from_numpy(ids), torch.from_numpy(mask), None,
                                      forced_routing=[rw[li, :, :shape.n_experts] for li in range(shape.n_layers)])
        _check(got, lg_forced, range(P + T), f'mixtral case {case} vs oracle with the engine routing')      # (2)
        _check(got, torch.from_numpy(ref), clean, f'mixtral case {case} vs reference (clean rows)', 5e-2)   # (3)
        _check(got, lg_own, clean, f'mixtral case {case} vs oracle (clean rows)', 5e-2)
    print(f'mixtral: {n_dec} of {n_all} rows decisively routed, {n_clean} rows with no expert flip in their attention span')
    assert n_dec >= 0.25 * n_all and n_clean >= 0.5 * n_all


def test_moe_greedy_steps_touch_two_experts_only():
    """T=1 steps route to top-2 experts: the other six expert GEMM pairs return at once; results must not depend on the
    stale contents of their scratch (run the same sequence after a wide step and after a reset)."""
    shape = moe_shape(TINY_MOE)
    sd = {k: v.to(torch.bfloat16) for k, v in moe_weights(TINY_MOE, 3, torch.float32).items()}
    eng = LlamaVerifyEngine(shape, dict(sd), max_length=256)
    rs = np.random.RandomState(0)
    prompt = rs.randint(3, 512, size=30).tolist()
    outs = []
    for poison in (False, True):
        eng.reset()
        if poison:                       # a 64-row step first, so every expert's scratch holds unrelated data
            eng.prefill(rs.randint(3, 512, size=64).tolist())
            eng.reset()
        tok = eng.prefill(prompt)
        seq = [tok]
        for _ in range(12):
            t, _ = eng.step(np.asarray([seq[-1]], dtype=np.int32), np.array([1], dtype=np.uint64))
            seq.append(t[0])
        outs.append((seq, eng.logits()[:1].clone()))
    assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1])
