# -*- coding: utf-8 -*-
"""-m gpu: Mistral (GQA) and Mixtral (sparse MoE) arithmetic through the engine against the REFERENCE's own logits
(tests/golden/moe_tiny_bf16.npz: cache-free reference forwards under a prompt+tree mask) and against the oracle.
Tolerance as in test_gpu_e2e.py."""
import os

import numpy as np
import pytest
import torch

from oracle import llama_oracle as lo
from painlessinferenceacceleration_amd.llama_engine import LlamaVerifyEngine
from tests.test_gpu_e2e import _check_rows
from tests.tiny_model import GOLDEN, TINY_GQA, TINY_MOE, moe_shape, moe_weights

pytestmark = pytest.mark.gpu


def _rows_of(mask, P, T):
    return np.array([sum(int(mask[P + i, P + j]) << j for j in range(T)) for i in range(T)], dtype=np.uint64)


@pytest.mark.parametrize('balanced', [True, False])
@pytest.mark.parametrize('kind', ['mixtral', 'mistral'])
def test_engine_matches_reference_mixtral_mistral_logits(kind, balanced):
    g = np.load(os.path.join(GOLDEN, 'moe_tiny_bf16.npz'))
    cfg = TINY_MOE if kind == 'mixtral' else TINY_GQA
    shape = moe_shape(cfg)
    sd = {k: v.to(torch.bfloat16) for k, v in moe_weights(cfg, 0, torch.float32).items()}
    eng = LlamaVerifyEngine(shape, dict(sd), max_length=256, balanced=balanced)
    oracle = lo.OracleLlama(shape, sd)
    for case, (P, T) in enumerate([(24, 40), (3, 61), (50, 1)]):
        ids, mask, ref = g[f'{kind}_{case}_ids'], g[f'{kind}_{case}_mask'].astype(np.int64), g[f'{kind}_{case}_logits']
        eng.reset()
        eng.prefill(ids[:P].tolist())
        _check_rows(eng.logits()[:P], torch.from_numpy(ref[:P]), range(P), f'{kind} case {case} prefill vs reference')
        eng.step(ids[P:].astype(np.int32), _rows_of(mask, P, T))
        _check_rows(eng.logits()[:T], torch.from_numpy(ref[P:]), range(T), f'{kind} case {case} tree vs reference')
        lg, _ = oracle.forward(torch.from_numpy(ids), torch.from_numpy(mask), None)
        _check_rows(eng.logits()[:T], lg[P:], range(T), f'{kind} case {case} tree vs oracle')


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
