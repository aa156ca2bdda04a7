# -*- coding: utf-8 -*-
"""GPU: `python bench.py --gpus N` called the way the driver calls the N = 1 command (no rank environment) spawns its own N
ranks and prints ONE JSON line.  On the 1-GPU box the ranks share the device (BENCH_SHARE_GPU=1) and the accepted-token
gather runs over gloo on host tensors (BENCH_DIST_BACKEND=gloo): the control flow of the N-rank job, not its transport."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize('strict', [False, True])
def test_bench_self_launches_two_ranks(strict):
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR'):
        env.pop(k, None)
    env.update({'BENCH_SHARE_GPU': '1', 'BENCH_DIST_BACKEND': 'gloo'})
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '6', '--warmup', '2', '--layers', '2',
           '--no-cpu-baseline', '--secondary', ''] + (['--strict-gather'] if strict else [])
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-1000:], r.stderr[-3000:])
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['scaling'] == 'weak' and j['steps'] == 6
    c = j['config']
    assert c['gather_mode'] == ('strict' if strict else 'split-phase')
    assert c['gather_transport'] == 'torch.distributed(gloo)' and c['rccl_ranks'] == 0      # the transport is named, never implied
    assert c['sequences'] == 2 and c['lookahead_equals_greedy'] is True


@pytest.mark.gpu
def test_bench_self_launches_eight_ranks_and_runs_config4_as_its_own_eight_rank_job():
    """`python bench.py --gpus 8` as the driver's scaling run calls it (no rank environment): 8 ranks of the headline workload, then
    BASELINE config 4 — Llama-2-13B, 4 sequences per GPU = bs 32 batch-sharded over 8 ranks — as its own 8-rank job embedded under
    `secondary`.  The box has one GPU: the ranks share it and the gather runs over gloo (control flow, not transport), and both
    models are cut to 2 layers (--layers / --secondary-layers: launch-path test, not a metric)."""
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR'):
        env.pop(k, None)
    env.update({'BENCH_SHARE_GPU': '1', 'BENCH_DIST_BACKEND': 'gloo'})
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '4', '--warmup', '1', '--layers', '2',
           '--secondary-layers', '2', '--no-cpu-baseline', '--strict-gather']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-1000:], r.stderr[-3000:])
    j = json.loads(lines[0])
    assert j['n_gpus'] == 8 and j['scaling'] == 'weak' and j['config']['sequences'] == 8
    assert j['config']['gather_mode'] == 'strict' and j['config']['rccl_ranks'] == 0
    sec = j['secondary']
    assert len(sec) == 1 and 'error' not in sec[0], sec
    leg = sec[0]
    assert leg['n_gpus'] == 8 and leg['sequences'] == 32 and 'Llama-2-13B' in leg['workload'] and leg['n_layers'] == 2
    assert leg['lookahead_equals_greedy'] is True and leg['gather_transport'] == 'torch.distributed(gloo)'
