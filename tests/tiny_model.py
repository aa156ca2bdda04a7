# -*- coding: utf-8 -*-
"""The tiny seeded models shared by the oracle / golden / GPU parity tests.  The recipes themselves (weights as pure functions of a
numpy seed, warm-up copies) live build-free in oracle/tiny.py, which the golden generators import; this module adds the pieces that need
the product package (LlamaShape)."""
import os

import numpy as np

from oracle.tiny import (TINY, TINY_GQA, TINY_MOE, moe_weights, noisy_copies, tiny_decisive_weights,  # noqa: F401
                         tiny_weights)
from painlessinferenceacceleration_amd.llama_engine import LlamaShape

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def tiny_shape(**over):
    c = dict(TINY); c.update(over)
    return LlamaShape(c['n_layers'], c['hidden'], c['n_heads'], c['n_kv_heads'], c['ffn'], c['vocab'], c['rms_eps'])


def load_golden(tag):
    return np.load(os.path.join(GOLDEN, f'llama_tiny_{tag}.npz'))


def moe_shape(cfg):
    return LlamaShape(cfg['n_layers'], cfg['hidden'], cfg['n_heads'], cfg['n_kv_heads'], cfg['ffn'], cfg['vocab'],
                      cfg['rms_eps'], rope_theta=cfg['rope_theta'], n_experts=cfg['n_experts'], top_k=cfg['top_k'],
                      norm_cast_first=True)


