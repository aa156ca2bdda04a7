# -*- coding: utf-8 -*-
"""Diagnostic: relative logits error (max|d| / max|ref| per row) of the multi-block prefill chain vs the 64-row prefill,
both against the oracle, on the tiny seeded Llama."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import llama_oracle as lo
from painlessinferenceacceleration_amd.llama_engine import LlamaVerifyEngine
from tests.test_gpu_e2e import _bf16_sd
from tests.tiny_model import tiny_shape

shape = tiny_shape()
for seed in (1, 2, 3):
    sd = _bf16_sd(seed)
    oracle = lo.OracleLlama(shape, sd)
    eng = LlamaVerifyEngine(shape, sd, max_length=1024, n_slots=1, max_blocks=8)
    rs = np.random.RandomState(seed)
    for P in (40, 100, 150, 300, 512):
        prompt = rs.randint(3, shape.vocab, size=P).tolist()
        lg, _ = oracle.forward(torch.tensor(prompt), torch.tril(torch.ones((P, P), dtype=torch.long)), None)
        lg = lg.float()
        last = (P - 1) // 64
        rows = (P - 1) % 64 + 1
        ref = lg[last * 64:]
        eng.reset()
        eng.mprefill(0, prompt)
        a = eng.mlogits()[(last % 8) * 64:(last % 8) * 64 + rows].float().cpu()
        eng.reset()
        eng.prefill(prompt)
        b = eng.logits()[:rows].float().cpu()
        ea = ((a - ref).abs().max(1).values / ref.abs().max(1).values)
        eb = ((b - ref).abs().max(1).values / ref.abs().max(1).values)
        eab = ((a - b).abs().max(1).values / ref.abs().max(1).values)
        print(f'seed {seed} P {P}: mb vs oracle max {float(ea.max()):.4f} mean {float(ea.mean()):.4f} | 64-row vs oracle max {float(eb.max()):.4f} '
              f'mean {float(eb.mean()):.4f} | mb vs 64-row max {float(eab.max()):.4f}', flush=True)
