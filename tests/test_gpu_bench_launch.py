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
    assert c['gather_us_per_step'] >= 0 and c['slowest_rank_wait_us'] >= 0 and 'notes' in j
    assert any(ln.startswith('BENCH_DETAIL {') for ln in r.stdout.splitlines())      # the full record precedes the compact line


@pytest.mark.gpu
def test_bench_two_ranks_on_rccl_sharing_the_one_gpu_or_the_documented_refusal():
    """Multi-GPU readiness without a multi-GPU node (round-5 review item 8): the N = 2 job with the REAL transport — torch.distributed 'nccl'
    (= RCCL) for the control collectives and la_gather_accepted (ncclAllGather) for the per-step gather — both ranks on the box's single
    device.  RCCL may refuse two ranks on one device (ncclInvalidUsage / 'Duplicate GPU detected'): then the refusal is what this test
    documents (skip with RCCL's own message) and the gloo test above remains the N > 1 control-flow evidence."""
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR', 'BENCH_DIST_BACKEND'):
        env.pop(k, None)
    env.update({'BENCH_SHARE_GPU': '1', 'HSA_ENABLE_IPC_MODE_LEGACY': '0'})
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '6', '--warmup', '2', '--layers', '2',
           '--no-cpu-baseline', '--secondary', '']
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420, cwd=ROOT)
    except subprocess.TimeoutExpired:
        pytest.skip('RCCL with two ranks on one device did not complete within 420 s (treated as a refusal); gloo covers the control flow')
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    if r.returncode != 0 or len(lines) != 1:
        tail = (r.stderr or r.stdout)[-3000:]
        refusal = [w for w in ('Duplicate GPU', 'invalid usage', 'ncclInvalidUsage', 'NCCL error', 'ncclUnhandledCudaError', 'ncclSystemError', 'DistBackendError') if w in tail]
        assert refusal, tail                                    # anything else is a bug of ours, not RCCL's refusal
        pytest.skip('RCCL refuses two ranks on one device (%s): documented; gloo covers the N > 1 control flow' % ', '.join(refusal))
    j = json.loads(lines[0])
    c = j['config']
    assert j['n_gpus'] == 2 and c['sequences'] == 2 and c['lookahead_equals_greedy'] is True
    assert c['rccl_ranks'] == 2 and ('rccl' in c['gather_transport'] or 'nccl' in c['gather_transport'])
    assert c['gather_us_per_step'] is not None


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
    assert leg['n_gpus'] == 8 and leg['sequences'] == 32 and leg['model'] == 'Llama-2-13B' and leg['n_layers'] == 2      # compact leg (bench.compact_leg)
    assert leg['equals_greedy'] is True and leg['gather_transport'] == 'torch.distributed(gloo)'
    assert leg['gather_us_per_step'] is not None and leg['slowest_rank_wait_us'] is not None
    assert len(lines[0]) < 8000                                  # the whole record fits the driver's captured 8 KB tail
