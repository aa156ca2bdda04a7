# -*- coding: utf-8 -*-
"""CPU: the parts of bench.py that do not need a GPU — so that an import error or a broken helper is caught by the
`-m "not gpu"` suite, not at round end on the GPU box."""
import json
import os

import numpy as np

import bench
from painlessinferenceacceleration_amd.llama_engine import LlamaShape

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_algorithmic_bytes_match_survey_8d():
    """Llama-2-7B W = 2 * (N_params - N_embed) = 13 214 687 232 B; KV 524 288 B per context token."""
    s7 = LlamaShape.llama2_7b()
    assert 2 * s7.n_params_no_embed() == 13214687232
    s13 = LlamaShape.llama2_13b()
    assert 2 * s13.n_params_no_embed() == 25704048640
    b0 = bench.algorithmic_bytes(s7, 64, 0, 0)
    b1 = bench.algorithmic_bytes(s7, 64, 640, 0)
    assert b1 - b0 == 640 * 524288
    assert b0 == 13214687232 + 64 * 524288 + 64 * (4096 * 2 + 8)
    # the SURVEY's worked example: bs=1, T=64, C=640 -> ~13.58 GB
    assert abs(b1 - 13.58e9) < 0.02e9


def test_synthetic_workload_generators_are_deterministic():
    a = bench.phrase_prompt(1234, 512, 32000)
    assert a == bench.phrase_prompt(1234, 512, 32000) and len(a) == 512 and min(a) >= 3 and max(a) < 32000
    truth = list(range(100, 400))
    c1 = bench.noisy_copies(truth, 12, 0.3, 32000, seed=99)
    c2 = bench.noisy_copies(truth, 12, 0.3, 32000, seed=99)
    assert c1 == c2 and len(c1) == 12
    kept = np.mean([np.mean(np.array(c) == np.array(truth)) for c in c1])
    assert 0.6 < kept < 0.8                      # rho = 0.3 of the tokens are replaced


def test_cpu_baseline_leg_runs_on_a_small_shape():
    s = LlamaShape(2, 256, 2, 2, 512, 1024, 1e-5)
    r = bench.cpu_baseline(s, 16, 64, 5.0, budget_s=3.0)
    assert r['kind'] == 'port' and r['unit'] == 'tokens/s' and r['value'] > 0 and r['cores'] >= 1
    assert r['ms_per_step'] > 0 and 'sample' in r


def test_committed_pmc_summary_is_what_bench_reads():
    pmc = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_latest.json')))
    keys = [k for k in pmc['kernels'] if k.startswith('void k_gemm64r<4, 1, 4, 8')]
    assert len(keys) == 1
    e = pmc['kernels'][keys[0]]
    s7 = LlamaShape.llama2_7b()
    algorithmic = 2 * s7.ffn * s7.hidden * 2 + 64 * s7.hidden * 2 + 64 * s7.ffn * 2
    assert algorithmic <= e['hbm_bytes_per_launch'] <= 1.10 * algorithmic      # nothing is re-read from HBM
