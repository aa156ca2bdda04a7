# -*- coding: utf-8 -*-
"""bench.py — accepted tokens/s of the LOOKAHEAD verify loop on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 64 --warmup 8
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (configs[1]): Llama-2-7B (synthetic weights, real shape, bf16; see random_weights(decisive=True)), bs=1 per GPU, 64-token draft tree /
~8 branches per verify step, hier mode, decoding_length=64, branch_length=12.  One "step" = trie query ->
captured verify graph (embed, 32 layers, lm_head+argmax, accept scan, KV commit) -> trie update.
Synthetic data: the prompt is 512 phrase-bank tokens; the trie is warmed, as the reference's
Benchmark.warm_up does (benchmarks/benchmark.py:159-169), with 12 noisy copies of the model's own greedy
continuation (each token replaced with probability rho=0.3), so drafts are multi-branch and partially accepted.
Multi-GPU: independent sequences per rank (batch sharding, weak scaling); the only exchange is the per-step all-gather of
accepted tokens over RCCL (la_gather_accepted) so that every rank's trie replica sees every sequence.  Default = split-phase
mode: started after step k, collected while the GPU runs step k+1, then applied for all sequences in global batch-index order
(replicas stay identical; a step's tokens reach the drafts one step later; emitted tokens are unaffected).  --strict-gather
selects the blocking mode in which the trie state at query time equals the reference's single-process order; the JSON line
says which mode a number comes from.

Other workloads (secondary lines, same JSON shape; BASELINE configs 3-4): --batch B runs B sequences per GPU, each with its OWN
64-token tree per step, in one pass over the weights (la_llama_mstep, M = 64 B rows); --model 13b|mistral picks the shape.
The line also carries: the 8(d) fixed "T64/B8" tree with the accept sweep a in {0,3,6,12} (`fixed_tree_sweep`), the
reference-style speed including prefill (`speed_incl_prefill`), and `cpu_baseline`: the oracle's own lookahead loop (a port of
the reference's transformers CPU path) run on this box's host cores on the full 32-layer model for a few real verify steps.
"""
import argparse
import json
import os
if any(os.environ.get(_k) for _k in ('LA_PF_KIB', 'LA_DEBUG', 'LA_MB_KS2')):
    os.environ.setdefault('LA_LAB_BUILD', '1')      # kernel-lab knobs exist in the lab build only (csrc/la_knobs.h): the A/B runs take it as the process library
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def phrase_prompt(seed, n, vocab):
    rs = np.random.RandomState(seed)
    bank = [rs.randint(3, vocab, size=rs.randint(4, 16)).tolist() for _ in range(400)]
    w = 1.0 / np.arange(1, 401) ** 1.3
    w /= w.sum()
    out = []
    while len(out) < n:
        out.extend(bank[rs.choice(400, p=w)])
    return out[:n]


def noisy_copies(truth, n_copies, rho, vocab, seed):
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(n_copies):
        t = np.array(truth)
        hit = rs.rand(len(t)) < rho
        t[hit] = rs.randint(3, vocab, size=int(hit.sum()))
        out.append(t.tolist())
    return out


def algorithmic_bytes(shape, T, ctx, logits_bytes):
    """W + KVr + KVw + A per verify step (bf16)."""
    W = 2 * (shape.n_params_no_embed())
    kv_tok = 2 * shape.n_layers * shape.n_kv_heads * shape.head_dim * 2
    return W + kv_tok * ctx + kv_tok * T + T * (shape.hidden * 2 + 8) + logits_bytes


HBM_ACHIEVABLE_GBS = 6290.0   # MI355X_MICROARCH.md: 6.29 TB/s measured (float4 copy), the rate a streaming kernel can reach


def floor_model(shape, n_cus=256):
    """DESIGN.md section 4 (round 4): what the launch-per-GEMM design of the 64-row step can reach.  Every GEMM launch obeys
    T = fixed + W_bytes / stream + x_bytes_per_CU / l2_rate  (fixed ~3.5 us of pipeline fill + reduction / epilogue; weights stream from
    HBM at ~6.4 TB/s chip-wide = 25 GB/s per CU; the activation operand every workgroup re-reads comes from its XCD's L2 at ~130 GB/s
    per CU, and a CU's vector-memory path returns in order, so the two times ADD).  -> the terms, the GEMM floor of one step and the
    fraction of the 8 TB/s roof a step made of these launches alone would reach (non-GEMM kernels at zero)."""
    fixed_us, stream, l2 = 3.5, 6.4e12, 130e9
    H, F, V, hd = shape.hidden, shape.ffn, shape.vocab, shape.head_dim
    qkv_n = (shape.n_heads + 2 * shape.n_kv_heads) * hd
    ks = max(1, n_cus // max(H // 64, 1))                 # K splits of the slab GEMMs: row-blocks x splits fit one wave of workgroups
    launches = [('qkv', qkv_n * H * 2, 64 * H * 2), ('o_proj', H * shape.n_heads * hd * 2, 64 * shape.n_heads * hd * 2 // ks),
                ('gate_up', 2 * F * H * 2, 64 * H * 2), ('down', H * F * 2, 64 * F * 2 // ks)]
    per_layer = {n: fixed_us + 1e6 * w / stream + 1e6 * x / l2 for n, w, x in launches}
    lm = fixed_us + 1e6 * V * H * 2 / stream + 1e6 * 64 * H * 2 / l2
    n_exp = max(shape.n_experts, 1)
    mlp = (per_layer['gate_up'] + per_layer['down']) * n_exp
    gemm_us = shape.n_layers * (per_layer['qkv'] + per_layer['o_proj'] + mlp) + lm
    return {'form': 'T_launch = fixed + W / stream + x_per_CU / l2', 'fixed_us': fixed_us, 'stream_TBps': stream / 1e12, 'l2_per_cu_GBps': l2 / 1e9,
            'per_layer_us': {k: round(v, 2) for k, v in per_layer.items()}, 'lm_head_us': round(lm, 2),
            'gemm_floor_ms_per_step': round(gemm_us / 1e3, 4)}


def fixed_t64b8_tree():
    """SURVEY 8(d) "T64/B8": 63 draft nodes + root, 8 leaves all at depth 12 — a main chain of 12 plus 7 side branches forking
    after depth 9,8,6,4,3,2,1 with lengths 3,4,6,8,9,10,11 (DFS order: main chain, then the deepest fork first).
    -> (parent[64], depth[64], uint64 row masks[64])"""
    parent, depth = [-1], [0]
    for d in range(1, 13):
        parent.append(d - 1); depth.append(d)
    for fork, length in ((9, 3), (8, 4), (6, 6), (4, 8), (3, 9), (2, 10), (1, 11)):
        p = fork                                   # main-chain node of that depth has index == depth
        for k in range(length):
            parent.append(p); depth.append(fork + 1 + k)
            p = len(parent) - 1
    assert len(parent) == 64 and sum(1 for i in range(64) if i not in parent) == 8 and max(depth) == 12
    rows = []
    for i, p in enumerate(parent):
        rows.append((rows[p] if p >= 0 else 0) | (1 << i))
    return parent, depth, np.array(rows, dtype=np.uint64)


def host_batch_drafts(cache, tails, idxs, DL, BL, ubls):
    """Drafts of all sequences of a batch step from the HOST trie -> [(ids int32[T], rowmask uint64[T])].  One native call for the
    whole batch (la_cache_bat_get_packed, what pretrained_model_batch.lookahead_generation uses: per-sample budget DL, min_output_size
    DL // 2) when every sequence still has the full branch length; per-sample calls otherwise (a sequence near max_length clamps its
    branch length, pretrained_model.py:680).  8 queries: 103 us in one call vs 257 us as 8 Python-level calls (build container)."""
    one = lambda t: (np.asarray(t[-1:], dtype=np.int32), np.array([1], dtype=np.uint64))
    if len(tails) > 1 and all(u == BL for u in ubls) and DL <= 64:
        got = cache.bat_get_packed(tails, decoding_length=DL * len(tails), branch_length=BL, mode='mix', indices=list(idxs),
                                   decoding_mode='hier')
        return [(g[0], g[1]) if len(g[0]) else one(t) for g, t in zip(got, tails)]
    out = []
    for t, ix, u in zip(tails, idxs, ubls):
        ids, rowmask, _, _ = cache.hier_get_packed(t, decoding_length=DL, branch_length=u, min_input_size=0, min_output_size=DL // 2,
                                                   mode='mix', idx=ix)
        out.append((ids.copy(), rowmask.copy()) if len(ids) else one(t))
    return out


def _pf_setting():
    """(KiB per consumer workgroup, start delay, gate/up tail KiB) of the weight prefetch in effect (library default or LA_PF_KIB)."""
    from painlessinferenceacceleration_amd import _lib
    if not _lib.LAB_BUILD:               # the product library: the constexpr defaults of csrc/la_knobs.h (prefetch off)
        return 0, 0, 0
    return _lib.lab_get(7), _lib.lab_get(8), _lib.lab_get(9)


_RDZV_KEYS = ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'GROUP_RANK', 'GROUP_WORLD_SIZE', 'ROLE_RANK', 'ROLE_WORLD_SIZE',
              'ROLE_NAME', 'MASTER_ADDR', 'MASTER_PORT', 'TORCHELASTIC_RESTART_COUNT', 'TORCHELASTIC_MAX_RESTARTS',
              'TORCHELASTIC_RUN_ID', 'TORCHELASTIC_USE_AGENT_STORE', 'TORCHELASTIC_ERROR_FILE', 'TORCH_NCCL_ASYNC_ERROR_HANDLING')


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def self_launch_cmd(argv, n, port):
    """The command `python bench.py --gpus N ...` turns itself into when no rank environment is present: one process per GPU
    under torch.distributed.run on this node (the shape of the driver's own N > 1 launch), rendezvous on 127.0.0.1."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={int(n)}', '--master-addr', '127.0.0.1',
            '--master-port', str(int(port)), os.path.abspath(__file__)] + [str(a) for a in argv]


def clean_rank_env(env=None):
    """environment for a fresh N-rank job started from inside (or outside) another one: no inherited rank / rendezvous variables"""
    env = dict(os.environ if env is None else env)
    for k in _RDZV_KEYS:
        env.pop(k, None)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return env


def self_launch(args_gpus, argv):
    """`python bench.py --gpus N` with no WORLD_SIZE: spawn the N ranks ourselves; rank 0's single JSON line is the child's stdout."""
    import subprocess
    r = subprocess.run(self_launch_cmd(argv, args_gpus, _free_port()), env=clean_rank_env())
    sys.exit(r.returncode)


def secondary_legs(spec, gpus=1, layers=0):
    """The batch configurations (BASELINE configs 3-5) as secondary lines of the default run: each `model:batch` leg is this script
    run again in its own process (`--model M --batch B`, 24 timed steps, no CPU leg) after the headline's timed region; the
    fields a reader needs are kept.  A leg that fails is reported as such — it never touches the headline line."""
    import subprocess
    legs = []
    for item in spec.split(','):
        item = item.strip()
        if not item:
            continue
        parts = item.split(':')
        model, batch = parts[0], parts[1]
        dev_trie_leg = len(parts) > 2 and parts[2] == 'dev'      # model:batch:dev = the same leg with the drafts from the ON-GPU trie (the default from 8 sequences on)
        leg_args = ['--gpus', str(gpus), '--model', model, '--batch', batch, '--steps', '24', '--warmup', '4', '--no-cpu-baseline']
        if dev_trie_leg:
            leg_args.append('--device-trie')
        if len(parts) > 2 and parts[2] == 'host':                # model:batch:host = the drafts from the host trie where the default is the on-GPU trie
            leg_args.append('--host-trie')
        if len(parts) > 2 and parts[2] == 'deferred':            # model:batch:deferred = the host-trie leg with the trie update under the next pass
            leg_args += ['--deferred-trie-update', '--host-trie']
        if layers:                       # launch-path tests only (tests/test_gpu_bench_launch.py): a truncated model, flagged in the leg
            leg_args += ['--layers', str(int(layers)), '--steps', '6', '--warmup', '2']
        cmd = [sys.executable, os.path.abspath(__file__)] + leg_args if gpus == 1 else self_launch_cmd(leg_args, gpus, _free_port())
        env = clean_rank_env()
        env['BENCH_IS_SECONDARY'] = '1'
        t0 = time.time()
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420 if gpus == 1 else 900)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
            if r.returncode != 0 or not line:
                legs.append({'model': model, 'batch': int(batch), 'error': (r.stderr or r.stdout)[-300:], 'wall_s': round(time.time() - t0, 1)})
                continue
            legs.append(compact_leg(json.loads(line[-1]), item, round(time.time() - t0, 1)))
        except Exception as e:           # noqa: BLE001 — a secondary leg must never take the headline line down
            legs.append({'model': model, 'batch': int(batch), 'error': repr(e)[:300], 'wall_s': round(time.time() - t0, 1)})
    return legs


def compact_leg(j, name, wall_s=None):
    """One secondary leg as ~300 bytes: what a reader of the driver's captured tail needs to judge the leg; the prose that is the same for
    every leg lives once in the headline's `notes` table, the leg's full record in its own BENCH_DETAIL line / detail file."""
    c, r = j['config'], (j.get('roofline') or {})
    alg = (r.get('hbm') or {}).get('algorithmic_bytes') or (r.get('verify_step') or {}).get('algorithmic_bytes')
    leg = {'name': name, 'model': c.get('model'), 'sequences': c.get('sequences'), 'n_gpus': j.get('n_gpus'), 'ms_per_step': j['ms_per_step'], 'value': j['value'],
           'accept_len': c.get('mean_accept_len'), 'draft_len': c.get('mean_draft_len'), 'bound': r.get('bound'), 'frac': r.get('frac'),
           'hbm_frac': (r.get('hbm') or r.get('verify_step') or {}).get('frac'), 'mfma_frac': (r.get('mfma') or {}).get('frac'),
           'traffic_ratio': round(r['traffic'] / alg, 3) if (r.get('traffic') and alg) else None,
           'draft_retrieval': c.get('draft_retrieval'), 'trie_update': c.get('trie_update'), 'equals_greedy': c.get('lookahead_equals_greedy'),
           'steps': j.get('steps')}
    if c.get('n_layers_truncated'):
        leg['n_layers'] = c.get('n_layers')
    if j.get('n_gpus', 1) > 1:
        leg.update({'gather_mode': c.get('gather_mode'), 'gather_transport': c.get('gather_transport'), 'rccl_ranks': c.get('rccl_ranks'),
                    'gather_us_per_step': c.get('gather_us_per_step'), 'slowest_rank_wait_us': c.get('slowest_rank_wait_us')})
    if wall_s is not None:
        leg['wall_s'] = wall_s
    return leg


NOTES = {
    'workload': 'lookahead verify loop (hier drafts): per sequence a 64-token draft tree over 8-12 noisy branches of its greedy continuation, '
                'synthetic permutation-LM weights (N(0,0.02); o/down std 1e-4; lm_head[pi(t)] = embed[t]) for decisive greedy margins, 512-token phrase-bank prompts',
    'value': 'accepted tokens / s over the timed verify steps, inputs resident in HBM; prefill excluded (speed_incl_prefill = the reference headline definition)',
    'roofline.timing': 'dominant kernel from live HIP events on the engine stream (all layers\' launches back to back per event pair); rocprofv3 averages of the same command: newest profiles/r*_profile_raw.txt',
    'roofline.traffic': 'HBM bytes per launch (bs=1) / per verify step (batch legs) from committed FETCH_SIZE / WRITE_SIZE passes (profiles/pmc_latest.json, pmc_secondary.json); traffic_ratio = traffic / algorithmic bytes',
    'draft_retrieval': 'host = native C++ trie (la_cache_*); device = on-GPU trie (la_trie_wg.hip: one workgroup per query), drafts chained in front of the verify pass, trie update on the device',
    'trie_update': 'ref-order = before the next query (reference order); deferred = under the next verify pass (drafts see a step one step later); device = la_trie_stream_put_dev',
    'secondary': 'each leg = this script in its own process after the headline\'s timed region (24 steps, 4 warm-up): BASELINE config 3 (mistral:8 = on-GPU trie, the default from 8 sequences per GPU on; :host = host trie in the reference update order, :deferred = host trie, update under the next pass), config 4 per-GPU share (13b:4), config 5 (mixtral:4), and 13b:1',
    'detail': 'the full record (every field of earlier rounds) is the BENCH_DETAIL line above this one and gpurun_out/bench_detail_*.json',
}


def compact_record(out):
    """The driver keeps the last 8 KB of stdout: the final JSON line carries the contract fields and one compact object per secondary leg;
    prose and bulky diagnostics stay in the BENCH_DETAIL line printed before it."""
    c, r, cpu = out['config'], out.get('roofline') or {}, out.get('cpu_baseline')
    sip = c.get('speed_incl_prefill') or {}
    cfg = {k: c.get(k) for k in ('workload', 'model', 'n_layers', 'n_layers_truncated', 'prompt_len', 'parallelism', 'sequences', 'kv_cache', 'gather_mode', 'gather_transport',
                                 'rccl_ranks', 'gather_us_per_step', 'slowest_rank_wait_us', 'trie_update', 'draft_retrieval', 'mean_accept_len', 'mean_draft_len',
                                 'verify_steps_per_sec', 'context_mean_timed', 'context_at_end', 'trie_query_ms_mean', 'lookahead_equals_greedy',
                                 'plain_greedy_tokens_per_sec', 'device_trie_stats') if c.get(k) is not None}
    cfg['speed_incl_prefill'] = {k: sip.get(k) for k in ('prefill_ms', 'tokens_per_sec', 'at_256_new_tokens')}
    if c.get('native_loop'):
        cfg['native_loop'] = {k: c['native_loop'].get(k) for k in ('ms_per_step', 'equals_greedy')}
    roof = {k: r.get(k) for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'frac_of_achievable', 'bytes_per_launch', 'ms_per_launch') if k in r}
    if 'verify_step' in r:
        v = r['verify_step']
        fm = v.get('floor_model') or {}
        roof['verify_step'] = {'algorithmic_bytes': v.get('algorithmic_bytes'), 'ms_graph_step': v.get('ms_graph_step'), 'achieved_GBps': v.get('achieved_GBps'),
                               'frac': v.get('frac'), 'frac_of_achievable': v.get('frac_of_achievable'), 'ms_by_class_events': v.get('ms_by_class_events'),
                               'floor_model': {k: fm.get(k) for k in ('gemm_floor_ms_per_step', 'nongemm_ms_per_step', 'frac_of_peak_if_nongemm_were_free', 'frac_of_peak_at_floor') if k in fm}}
    if 'hbm' in r:
        roof['hbm'] = {k: r['hbm'].get(k) for k in ('algorithmic_bytes', 'achieved_GBps', 'frac')}
    if 'mfma' in r:
        roof['mfma'] = {k: r['mfma'].get(k) for k in ('achieved_TFLOPs', 'frac')}
    comp = {k: out[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data')}
    comp['config'] = cfg
    comp['roofline'] = roof
    comp['cpu_baseline'] = None if cpu is None else {k: cpu.get(k) for k in ('value', 'unit', 'cores', 'kind', 'sample', 'ms_per_step', 'mean_accept_len', 'cpu_model', 'verify_steps', 'fallback_reason') if cpu.get(k) is not None}
    if 'secondary' in out:
        comp['secondary'] = out['secondary']
    comp['notes'] = NOTES
    return comp


def cpu_baseline_leg(shape, sd_cpu, prompt, copies, branch_length, decoding_length, verify_steps=5, threads=None):
    """`cpu_baseline`: the REFERENCE ITSELF (kind "reference", oracle/reference_cpu.py) where /root/reference is importable — the build container —
    and the port of it (kind "port", cpu_baseline_loop below: the oracle loop, pinned token for token to the reference) everywhere else (the GPU box)."""
    why = None
    try:
        from oracle import reference_cpu
        if reference_cpu.reference_available() and shape.n_experts == 0 and not os.environ.get('BENCH_CPU_PORT'):
            r = reference_cpu.reference_lookahead_loop(shape, sd_cpu, prompt, copies, branch_length, decoding_length, verify_steps=verify_steps, threads=threads,
                                                       dtype=next(iter(sd_cpu.values())).dtype)
            r.pop('tokens', None)
            return r
    except Exception as e:                   # noqa: BLE001 — fall back to the port and say why
        why = repr(e)[:200]
    r = cpu_baseline_loop(shape, sd_cpu, prompt, copies, branch_length, decoding_length, verify_steps=verify_steps, threads=threads)
    if why:
        r['fallback_reason'] = why
    return r


def cpu_model_name():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline_loop(shape, sd_cpu, prompt, copies, branch_length, decoding_length, verify_steps=5, threads=None):
    """`cpu_baseline` (kind "port"): the oracle's OWN lookahead loop — oracle/llama_oracle.py::lookahead_generate over
    oracle/trie_oracle.py, the restatement of the reference's transformers CPU path (pretrained_model.py:947-1268 +
    lookahead_cache.py) — on the FULL model (all layers, bf16, the weights the GPU run uses), same prompt, same trie warm-up,
    for `verify_steps` real verify steps after the prefill.  Accepted tok/s = its own accepted tokens / its own step times
    (prefill excluded, like `value`).  dtype and thread count are PINNED (bf16; min(host cores, 16) threads: more threads were
    slower on every box measured in round 1) and recorded with the CPU model, so the number does not move with a search."""
    from oracle import llama_oracle as lo
    from oracle.trie_oracle import TrieOracle
    ncpu = os.cpu_count() or 1
    nt = threads or min(ncpu, 16)
    torch.set_num_threads(nt)
    model = lo.OracleLlama(shape, sd_cpu)
    cache = TrieOracle(eos_ids=[None])
    for c in copies:
        cache.put(c, branch_length=branch_length + 1, mode='output', idx=-1)
    t0 = time.time()
    out = lo.lookahead_generate(model, cache, prompt, len(prompt) + (verify_steps + 1) * (branch_length + 1) + 2, eos_token_id=None,
                                decoding_length=decoding_length, branch_length=branch_length, max_steps=verify_steps + 1)
    wall = time.time() - t0
    fts, edls, dls = out['fts'], out['edls'], out['dls']
    t_dec, n_acc = float(sum(fts[1:])), int(sum(edls[1:]))
    return {'value': round(n_acc / max(t_dec, 1e-9), 3), 'unit': 'tokens/s', 'cores': nt, 'kind': 'port',
            'ms_per_step': round(1e3 * t_dec / max(len(fts) - 1, 1), 1), 'dtype': 'bfloat16', 'cpu_model': cpu_model_name(),
            'host_cores': ncpu, 'verify_steps': len(fts) - 1, 'mean_accept_len': round(n_acc / max(len(fts) - 1, 1), 3),
            'mean_draft_len': round(float(np.mean(dls[1:])) if len(dls) > 1 else 0.0, 2),
            'prefill_s': round(float(fts[0]), 2), 'wall_s': round(wall, 1),
            'sample': f'oracle lookahead loop (port of the reference transformers CPU path) on the full {shape.n_layers}-layer model: prefill of '
                      f'{len(prompt)} tokens + {len(fts) - 1} verify steps with its own drafts (same prompt and trie warm-up as the GPU run), '
                      f'bf16, {nt} threads pinned'}


def cpu_only(args):
    """bench.py --cpu-baseline-only: the CPU leg alone, on host cores, no HIP device touched.  The greedy continuation the noisy trie copies are
    drawn from comes from the oracle (plain greedy with a KV cache) instead of the GPU engine; everything else is the leg of the default run."""
    from oracle import llama_oracle as lo
    from painlessinferenceacceleration_amd.llama_engine import LlamaShape, random_weights
    shape = {'7b': LlamaShape.llama2_7b, '13b': LlamaShape.llama2_13b}[args.model]()
    if args.layers:
        shape.n_layers = args.layers
    P, BL, DL = args.prompt_len, args.branch_length, args.decoding_length
    tdtype = torch.float16 if args.dtype == 'fp16' else torch.bfloat16
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    sd = random_weights(shape, seed=0, device='cpu', decisive=True, dtype=tdtype)
    prompt = phrase_prompt(1234, P, shape.vocab)
    n_truth = (args.cpu_steps + 2) * (BL + 1) + 8
    model = lo.OracleLlama(shape, sd)
    t0 = time.time()
    seq = list(prompt)
    lg, past = model.forward(torch.tensor(seq), torch.tril(torch.ones((P, P), dtype=torch.long)), None)
    for _ in range(n_truth):
        t = int(lg[-1].float().argmax())
        seq.append(t)
        lg, past = model.forward(torch.tensor([t]), torch.ones((1, len(seq)), dtype=torch.long), past)
    t_truth = time.time() - t0
    del model, past
    copies = noisy_copies(prompt[-2:] + seq[P:], args.copies, args.rho, shape.vocab, seed=99)
    cpu = cpu_baseline_leg(shape, sd, prompt, copies, BL, DL, verify_steps=args.cpu_steps)
    print(json.dumps({'metric': 'accepted_tokens_per_sec', 'cpu_baseline': cpu, 'n_layers': shape.n_layers, 'model': args.model,
                      'plain_greedy_s_per_token_cpu': round(t_truth / max(n_truth, 1), 3),
                      'note': 'CPU leg only (bench.py --cpu-baseline-only): no GPU was used; value = accepted tokens / s of the CPU loop'}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=64)
    ap.add_argument('--warmup', type=int, default=8)
    ap.add_argument('--prompt-len', type=int, default=512)
    ap.add_argument('--rho', type=float, default=0.3, help='token corruption rate of the warm-up copies')
    ap.add_argument('--copies', type=int, default=12, help='noisy copies of the continuation put into the trie')
    ap.add_argument('--layers', type=int, default=0, help='debug: override layer count (invalidates the metric)')
    ap.add_argument('--pure-random', action='store_true', help='plain N(0,0.02) init (greedy/lookahead drift apart in bf16)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-steps', type=int, default=5, help='verify steps of the CPU baseline loop')
    ap.add_argument('--model', choices=['7b', '13b', 'mistral', 'mixtral'], default='7b',
                    help='7b = the BASELINE metric; 13b / mistral / mixtral = the config-4 / config-3 / config-5 model shapes')
    ap.add_argument('--batch', type=int, default=1, help='sequences per GPU; > 1: each gets its own 64-token tree per step (la_llama_mstep)')
    ap.add_argument('--device-trie', action='store_true', help='--batch > 1: drafts of all sequences from ONE device launch over the incremental trie mirror (default at --batch 8 on one GPU, where it beats the host trie: profiles/r06_device_trie_wg.txt)')
    ap.add_argument('--host-trie', action='store_true', help='drafts from the host trie also where the on-GPU trie is the default (--batch 8)')
    ap.add_argument('--trie-algo', default='wg', choices=['wg', 'wave'], help='--device-trie: one workgroup per query (round 6, la_trie_wg.hip) | one wavefront per query (rounds 1-5, la_trie_dev.hip)')
    ap.add_argument('--host-trie-update', action='store_true', help='--device-trie: apply the per-step trie update on the host and ship it as a patch (round-2 form) instead of inserting the accepted tokens on the device (la_trie_stream_put_dev)')
    ap.add_argument('--unchained-trie', action='store_true', help='--device-trie: read the drafts back to the host and feed them through la_llama_mstep (round-2 form)')
    ap.add_argument('--strict-gather', action='store_true', help='N > 1: blocking all-gather (reference trie order at query time)')
    ap.add_argument('--strict-trie-order', action='store_true',
                    help='(the default since round 6; kept for old command lines) --batch > 1, host trie: the trie update of step k is applied BEFORE the '
                         'drafts of step k + 1 are retrieved — the reference\'s order and the product loop\'s default')
    ap.add_argument('--deferred-trie-update', action='store_true',
                    help='--batch > 1, host trie: the update of step k runs on the host after the verify pass of step k + 1 has been queued (mstep_async = '
                         'decoding_kwargs[\'overlap_trie_update\'] of the product loop), i.e. drafts see a step\'s tokens one step later — the split-phase '
                         'order the N > 1 job uses; emitted tokens are unaffected.  The round-5 records measured THIS order by default; the default '
                         'secondary legs carry it as mistral:8:deferred beside the reference-order line')
    ap.add_argument('--profile-iters', type=int, default=3)
    ap.add_argument('--attn-split', type=int, default=0, help='key splits of the tree-attention kernel (0 = engine default 8)')
    ap.add_argument('--fuse', type=int, default=0, help='engine cfg.fuse bits (opt-in in-kernel norm->GEMM fusion; 0 = separate kernels)')
    ap.add_argument('--gemm-cfg', default='', help='engine gemm_cfg override (comma list: qkv_rb,qkv_ks,o_rb,o_ks,down_rb,down_ks,lm_rb,gu_variant; 0 = default)')
    ap.add_argument('--secondary', default='mistral:8,mistral:8:host,mistral:8:deferred,13b:4,mixtral:4,13b:1',
                    help='N=1 default workload only: comma list of model:batch legs (BASELINE configs 3-5: a 64-token tree per sequence '
                         'through la_llama_mstep) run AFTER the timed region, each in its own process; their lines are embedded under '
                         '"secondary"; model:batch:dev = the same leg with the drafts from the on-GPU trie (BASELINE config 3 as stated: the host-trie line '
                         'stands beside it); 13b:1 = the 64-row step at a larger launch size (the HBM fraction rises with bytes per launch).  "" = none')
    ap.add_argument('--decoding-length', type=int, default=64, help='tree tokens per sequence and step (BASELINE: 64); > 64 (the reference\'s best '
                    'published setting is 128 with --branch-length 32, lookahead/README.md:100): wide trees through eng.tstep, --batch 1')
    ap.add_argument('--branch-length', type=int, default=12)
    ap.add_argument('--dtype', choices=['bf16', 'fp16'], default='bf16',
                    help='16-bit type of weights / activations / KV cache: bf16 = BASELINE (liblookahead_hip.so), fp16 = the reference\'s own dtype '
                         '(liblookahead_hip_f16.so); the line reports it in "dtype"')
    ap.add_argument('--secondary-layers', type=int, default=0,
                    help='launch-path tests only: run the secondary legs too although --layers truncates the model, each with this many '
                         'layers (their lines carry n_layers; such numbers are NOT the metric)')
    ap.add_argument('--secondary-multi', default=None,
                    help='N > 1 default workload only: model:batch legs run as their own N-rank jobs after the headline (default: '
                         '"13b:4" at --gpus 8 = BASELINE config 4, Llama-2-13B bs=32 batch-sharded over 8 GPUs; "" = none)')
    ap.add_argument('--cpu-baseline-only', action='store_true',
                    help='no GPU: time only the cpu_baseline leg (the reference itself where /root/reference is importable, else the port) on the '
                         'same model shape / prompt / trie warm-up rule and print it — how the "reference" kind is measured in the build container')
    args = ap.parse_args()
    if args.cpu_baseline_only:
        return cpu_only(args)

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ and 'RANK' not in os.environ:
        self_launch(args.gpus, sys.argv[1:])          # does not return

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    dist = None
    dist_on = world > 1 or bool(os.environ.get('BENCH_FORCE_DIST'))    # FORCE: exercise the RCCL path with one rank
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # test-only knobs (1-GPU box): BENCH_DIST_BACKEND=gloo + BENCH_SHARE_GPU=1 run N ranks through the same control
        # flow on ONE device, with the collectives on host tensors; the measured configuration is always nccl (= RCCL)
        backend = os.environ.get('BENCH_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda:0' if os.environ.get('BENCH_SHARE_GPU') else f'cuda:{local_rank}'))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    dev = 'cuda:0' if os.environ.get('BENCH_SHARE_GPU') else f'cuda:{local_rank}'
    comm_dev = dev if os.environ.get('BENCH_DIST_BACKEND', 'nccl') == 'nccl' else 'cpu'
    torch.cuda.set_device(dev)

    from painlessinferenceacceleration_amd.llama_engine import LlamaShape, random_weights
    from painlessinferenceacceleration_amd.lookahead_cache import LookaheadCache
    from painlessinferenceacceleration_amd.modeling_llama import LlamaForCausalLM
    from painlessinferenceacceleration_amd.modeling_llama_batch import LlamaForCausalLM as BatchLlama

    if any(os.environ.get(k) for k in ('LA_PF_KIB', 'LA_DEBUG', 'LA_MB_KS2')):
        from painlessinferenceacceleration_amd import _lib as _libmod
        assert _libmod.LAB_BUILD, 'LA_PF_KIB / LA_DEBUG / LA_MB_KS2 set kernel-lab knobs: the lab build must be the process library (LA_LAB_BUILD=1)'
    # measurement override of the library's idle-window prefetch default (la_debug_set keys 7 / 8 / 9, scripts/gpu_pf_ab.py)
    if os.environ.get('LA_PF_KIB') is not None:
        from painlessinferenceacceleration_amd._lib import check as _check, lab_set as _lab_set      # every loaded build (bf16 / fp16)
        _check(_lab_set(7, int(os.environ['LA_PF_KIB'])), 'debug_set')
        _check(_lab_set(8, int(os.environ.get('LA_PF_DELAY', '0'))), 'debug_set')
        _check(_lab_set(9, int(os.environ.get('LA_PF_TAIL', '0'))), 'debug_set')
    if os.environ.get('LA_DEBUG'):                       # measurement: any la_debug_set keys, "k=v,k=v" (scripts/gpu_knob_sweep.sh)
        from painlessinferenceacceleration_amd._lib import lab_set as _lab_set, check as _check
        for kv in os.environ['LA_DEBUG'].split(','):
            k, v = kv.split('=')
            _check(_lab_set(int(k), int(v)), 'debug_set')
    if os.environ.get('LA_MB_KS2') is not None:          # measurement: 2 K splits for the multi-block slab GEMMs at >= 5 blocks
        from painlessinferenceacceleration_amd._lib import check as _check, lab_set as _lab_set
        _check(_lab_set(12, int(os.environ['LA_MB_KS2'])), 'debug_set')

    shape = {'7b': LlamaShape.llama2_7b, '13b': LlamaShape.llama2_13b, 'mistral': LlamaShape.mistral_7b,
             'mixtral': LlamaShape.mixtral_8x7b}[args.model]()
    model_name = {'7b': 'Llama-2-7B', '13b': 'Llama-2-13B', 'mistral': 'Mistral-7B', 'mixtral': 'Mixtral-8x7B'}[args.model]
    if args.layers:
        shape.n_layers = args.layers
    K, W, P, B = args.steps, args.warmup, args.prompt_len, args.batch
    assert 1 <= B <= 16, 'sequences per GPU: <= 16 KV slots; more than 8 run as two passes of <= 8 blocks per step'
    assert B <= 8 or args.decoding_length <= 64, '--batch above 8: 64-token trees (two passes of <= 8 blocks per step; with --device-trie as the product loop runs them)'
    BL, DL = args.branch_length, args.decoding_length
    if not args.device_trie and not args.host_trie and B == 8 and world == 1 and DL <= 64 and not args.deferred_trie_update:
        args.device_trie = True          # round 6: the workgroup-per-query device trie wins from 8 sequences per GPU on (the product loop's AUTO rule)
    wide = DL > 64                       # trees wider than one 64-row block: the tree is 2-4 chained blocks of one multi-block pass (eng.tstep)
    assert 1 <= DL <= 256 and 1 <= BL <= 39, 'decoding_length <= 256, branch_length <= 39'
    # wide trees in a batch: every sequence's tree is ceil(DL / 64) blocks of the pass (eng.mstep_trees), all of them within the 8-block pass
    assert not wide or B == 1 or (B * ((DL + 63) // 64) <= 8 and not args.device_trie and world == 1), \
        'wide trees with --batch B: B x ceil(decoding_length / 64) <= 8 blocks, host trie, one GPU'
    n_truth = (K + W + 56) * (BL + 1) + 8          # measured steps + native-loop leg (16 steps) + fixed-tree sweep (24 steps) + slack
    max_length = P + n_truth + 2 * DL
    # Mistral (config 2): the checkpoint's sliding window (4096, HF Mistral-7B-v0.1 config.json) with the KV cache as a ring of
    # window + one step of rows per sequence (memory O(window), not O(max_length)); contexts of this workload stay inside the window
    if args.model == 'mistral' and B > 1:
        shape.sliding_window = 4096
    kv_ring = bool(getattr(shape, 'sliding_window', 0)) and B > 1
    if kv_ring:
        max_length = max(max_length, shape.sliding_window + 64 * B * ((DL + 63) // 64) + 64)
    want_cpu = not args.no_cpu_baseline and world == 1 and B == 1      # the CPU leg is timed on rank 0 at N=1 only
    tdtype = torch.float16 if args.dtype == 'fp16' else torch.bfloat16
    sd = random_weights(shape, seed=0, device=dev, decisive=not args.pure_random, dtype=tdtype)
    sd_cpu = {k: v.cpu() for k, v in sd.items()} if want_cpu else None
    gemm_cfg = [int(x) for x in args.gemm_cfg.split(',')] if args.gemm_cfg else None
    if B == 1:
        model = LlamaForCausalLM(shape, sd, device=dev, max_length=max_length, eos_token_id=None, consume_state_dict=True,
                                 fuse=args.fuse, attn_split=args.attn_split, max_blocks=8, gemm_cfg=gemm_cfg)
    else:
        model = BatchLlama(shape, sd, device=dev, max_length=max_length, max_batch=B, eos_token_id=None, consume_state_dict=True,
                           attn_split=args.attn_split, max_blocks=min(B * ((DL + 63) // 64), 8), kv_ring=kv_ring, gemm_cfg=gemm_cfg)
    del sd
    eng = model.engine
    NSEQ = world * B
    gidx = [i * world + rank for i in range(B)]          # global batch indices of this rank's sequences (b mod world == rank)

    # ---- untimed set-up: prompts, ground-truth continuations (plain greedy on the same engine), trie warm-up
    prompts = [phrase_prompt(1234 + b, P, shape.vocab) for b in range(NSEQ)]
    t0 = time.time()
    if B == 1:
        truth_own = [model.greedy_search(torch.tensor([prompts[gidx[0]]]), P + n_truth, eos_token_id=None)[0].tolist()[P:]]
    else:
        truth_own = model.greedy_search(torch.tensor([prompts[b] for b in gidx]), P + n_truth, eos_token_id=None)[:, P:].tolist()
    t_greedy = time.time() - t0
    cache = LookaheadCache(eos_ids=[None])
    model.lookahead_cache = cache
    truths = [None] * NSEQ
    if dist_on:                                  # every replica is warmed with every sequence's (noisy) answers
        tt = torch.tensor(truth_own, dtype=torch.int32, device=comm_dev)
        allt = [torch.empty_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)
        for r in range(world):
            for i in range(B):
                truths[i * world + r] = allt[r][i].cpu().tolist()
    else:
        for i in range(B):
            truths[gidx[i]] = truth_own[i]
    copies0 = None
    for b in range(NSEQ):
        cps = noisy_copies(prompts[b][-2:] + truths[b], args.copies, args.rho, shape.vocab, seed=99 + b)
        if b == 0:
            copies0 = cps
        for c in cps:
            cache.put(c, branch_length=BL + 1, mode='output', idx=-1)

    # ---- the measured loop ----------------------------------------------------------------------------------
    seqs = [list(prompts[b]) for b in gidx]
    for i, b in enumerate(gidx):
        cache.put(seqs[i][1:], branch_length=BL + 1, mode='input', idx=b)
    eng.reset()
    torch.cuda.synchronize()
    t0 = time.time()
    if B == 1 and wide:
        seqs[0].append(eng.mprefill(0, seqs[0]))
    elif B == 1:
        seqs[0].append(eng.prefill(seqs[0]))
    else:
        first = eng.mprefill_many({i: seqs[i] for i in range(B)})
        for i in range(B):
            seqs[i].append(first[i])
    torch.cuda.synchronize()
    t_prefill = time.time() - t0
    gather = None
    if dist_on:
        from painlessinferenceacceleration_amd.distributed import AcceptedTokenGather
        # the N-rank logic (strict / split-phase, batch-index order) lives in the package: step_update() after a step, overlap() once the
        # next pass is queued — the same calls lookahead_generation() makes under decoding_kwargs['gather']
        gather = AcceptedTokenGather(comm_dev, b_loc=B, branch_length=BL, mode='strict' if args.strict_gather else 'split-phase')
    put_q = [None]                    # B > 1, one GPU: the trie update of the last step, applied under the next verify pass
    overlap_put = B > 1 and args.deferred_trie_update and not args.strict_trie_order
    edls, dls, qts, ctxs = [], [], [], []          # ctxs: committed keys per sequence at the start of every step

    def drafts_for(i):
        ubl = min(BL, max_length - len(seqs[i]) - 1)
        ids, rowmask, _, _ = cache.hier_get_packed(seqs[i][-2:], decoding_length=DL, branch_length=ubl, min_input_size=0,
                                                   min_output_size=DL // 2, mode='mix', idx=gidx[i])
        if len(ids) == 0:
            return np.asarray(seqs[i][-1:], dtype=np.int32), np.array([1], dtype=np.uint64)
        return ids.copy(), rowmask.copy()

    dev_trie = None
    if args.device_trie and B > 1:
        from painlessinferenceacceleration_amd.device_trie import DeviceTrie
        dev_put = not args.host_trie_update and not args.unchained_trie and not dist_on
        dev_trie = DeviceTrie(cache, idxs=gidx, device=dev, put_vocab=shape.vocab if dev_put else None, algo=args.trie_algo)
        if dev_put:
            dev_trie.load_stream_buffers()

    def drafts_dev():
        ubl = [min(BL, max_length - len(seqs[i]) - 1) for i in range(B)]
        got = dev_trie.hier_get([seqs[i][-2:] for i in range(B)], idxs=gidx, branch_lengths=ubl, decoding_length=DL, branch_length=BL,
                                min_input_size=0, min_output_size=DL // 2, mode='mix')
        return [(np.asarray(g[0], dtype=np.int32), np.asarray(g[1], dtype=np.uint64)) if len(g[0]) else
                (np.asarray(seqs[i][-1:], dtype=np.int32), np.array([1], dtype=np.uint64)) for i, g in enumerate(got)]

    replay_q = [None, False]          # [puts of the last chained step not yet replayed on the host trie, full image due]

    def one_step_chained():
        # device trie chained in front of the verify pass: patch + query kernels and la_llama_mstep_trie on the engine's stream; the
        # drafts never leave HBM (the host reads back accepted tokens and draft lengths)
        tq = time.time()
        if gather is not None:
            gather.overlap(cache, BL)
        ubl = [min(BL, max_length - len(seqs[i]) - 1) for i in range(B)]
        n_pass = (B + 7) // 8
        if n_pass > 1 and replay_q[0] is not None:
            # more than 8 sequences: two engine passes per step, as pretrained_model_batch.py runs them — the host replays the previous step's
            # update FIRST (one replay covers the stream_put_dev calls of both passes), then ONE synced query launch serves all B sequences
            if not dev_trie.replay(replay_q[0], BL + 1, calls=n_pass):
                replay_q[1] = True
            replay_q[0] = None
        with torch.cuda.stream(eng.stream):
            # round 4: with device-side updates the device image already holds step N's inserts when step N + 1 is queued, so the
            # host's REPLAY of step N (25-50 us per sequence) moves behind the launches and runs while the GPU verifies
            dev_trie.hier_get_dev([seqs[i][-2:] for i in range(B)], idxs=gidx, branch_lengths=ubl, decoding_length=DL, branch_length=BL,
                                  min_input_size=0, min_output_size=DL // 2, mode='mix', sync=replay_q[0] is None)
            qts.append(time.time() - tq)
            toks_all, Ts = [], []
            for g0 in range(0, B, 8):
                grp = list(range(g0, min(B, g0 + 8)))
                eng.mstep_trie_async(dev_trie, g0, grp, [16] * len(grp), [seqs[i][-1] for i in grp],
                                     put_idxs=[gidx[i] for i in grp] if dev_trie.put_vocab else None, put_branch_length=BL + 1)
                if g0 == 0 and replay_q[0] is not None:
                    ok = dev_trie.replay(replay_q[0], BL + 1)
                    replay_q[0] = None
                    if not ok:
                        replay_q[1] = True                   # the host image outgrew the device's: finish this step, then a full image
                t_, T_ = eng.mstep_trie_finish()
                toks_all.extend(t_); Ts.extend(T_)
        for i in range(B):
            seqs[i].extend(toks_all[i])
            dls.append(Ts[i]); edls.append(len(toks_all[i]))
        if dev_trie.put_vocab:
            # the device inserted the accepted tokens into its trie image itself (behind the verify pass, from the step's output
            # block); the host trie repeats the same puts — during the NEXT step — and drops the words it logged
            replay_q[0] = [(gidx[i], toks_all[i]) for i in range(B)]
            if replay_q[1]:                          # rare: capacity passed / squeeze — replay now, the next query syncs a full image
                dev_trie.replay(replay_q[0], BL + 1, calls=n_pass)
                replay_q[0], replay_q[1] = None, False
        elif dist_on:
            gather.step_update(cache, toks_all if B > 1 else toks_all[0], BL)
        else:
            for i in range(B):
                cache.stream_put(toks_all[i], branch_length=BL + 1, final=False, idx=gidx[i])

    def one_step():
        ctxs.append(eng.n_keys if B == 1 else float(np.mean(eng.slot_keys[:B])))
        if dev_trie is not None and not args.unchained_trie:
            return one_step_chained()
        tq = time.time()
        if dev_trie is not None:
            dr = drafts_dev()
        elif B > 1:
            dr = host_batch_drafts(cache, [seqs[i][-2:] for i in range(B)], gidx, DL, BL,
                                   [min(BL, max_length - len(seqs[i]) - 1) for i in range(B)])
        else:
            dr = [drafts_for(0)]
        qts.append(time.time() - tq)
        if B == 1 and wide:
            toks_all = [eng.tstep(dr[0][0], dr[0][1], mode=0)[0]]
        elif B == 1:
            eng.step_async(dr[0][0], dr[0][1], mode=0)
            if gather is not None:       # N > 1, split-phase: the previous step's gather + every trie update run while the GPU verifies
                gather.overlap(cache, BL)
            toks_all = [eng.step_finish()[0]]
        elif wide:
            if gather is not None:
                gather.overlap(cache, BL)
            toks_all = eng.mstep_trees([(i, dr[i][0], dr[i][1], 0, 40) for i in range(B)])
        else:
            # queue the multi-block pass, then do the host work nothing on the device waits for while the GPU verifies: the previous
            # step's gather + puts (N > 1, split-phase) or its trie update (one GPU, unless --strict-trie-order)
            toks_all = []
            for g0 in range(0, B, 8):            # more than 8 sequences: one pass over the weights per group of 8 blocks
                eng.mstep_async([(i, dr[i][0], dr[i][1], 0, 16) for i in range(g0, min(B, g0 + 8))])
                if gather is not None:
                    gather.overlap(cache, BL)
                if put_q[0] is not None:
                    cache.stream_put_many(put_q[0], branch_length=BL + 1, final=False)
                    put_q[0] = None
                toks_all.extend(eng.mstep_finish())
        for i in range(B):
            seqs[i].extend(toks_all[i])
            dls.append(len(dr[i][0])); edls.append(len(toks_all[i]))
        if dist_on:                      # strict: every replica holds every token of the step before the next query; split-phase: collected
            gather.step_update(cache, toks_all if B > 1 else toks_all[0], BL)      # during the next verify step (overlap above)
        elif B > 1 and overlap_put and not wide:
            put_q[0] = [(gidx[i], toks_all[i]) for i in range(B)]
        elif B > 1:
            cache.stream_put_many([(gidx[i], toks_all[i]) for i in range(B)], branch_length=BL + 1, final=False)
        else:
            cache.stream_put(toks_all[0], branch_length=BL + 1, final=False, idx=gidx[0])

    import gc
    gc.collect()
    gc.freeze()          # a generation-2 collection over torch's import graph costs tens of ms on the host thread that
                         # drives the loop; the serving loop allocates nothing that needs cycle collection
    for _ in range(W):
        one_step()
    n0, q0, c0 = len(edls), len(qts), len(ctxs)
    if dist_on:
        dist.barrier()
    if gather is not None:               # the gather's wait statistics cover the timed steps only
        gather.stats = {'collectives': 0, 'wait_s': 0.0, 'wait_s_max': 0.0}
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(K):
        one_step()
    if gather is not None:               # the last step's gather and puts, inside the timed region
        gather.overlap(cache, BL)
    if put_q[0] is not None:             # ... and the last deferred trie update
        cache.stream_put_many(put_q[0], branch_length=BL + 1, final=False)
        put_q[0] = None
    if replay_q[0] is not None:          # the last chained step's trie update, replayed inside the timed region
        dev_trie.replay(replay_q[0], BL + 1, calls=(B + 7) // 8)
        replay_q[0] = None
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    elapsed = time.time() - t0
    accepted = int(sum(edls[n0:]))
    gather_us_per_step = slowest_rank_wait_us = None
    if dist_on:
        # what the first real N-GPU run needs to explain its own curve: host time per step spent WAITING for the accepted-token gather
        # (mean over this job's ranks; split-phase: the wait runs under the next verify pass), and the largest single wait any rank saw
        # (a rank waits for the slowest rank of the step: ragged accept lengths / a slow GPU show up here, not in bandwidth)
        gs = gather.stats if gather is not None else {'collectives': 0, 'wait_s': 0.0, 'wait_s_max': 0.0}
        own_mean_us = 1e6 * gs['wait_s'] / max(gs['collectives'], 1)
        v = torch.tensor([elapsed, float(accepted), own_mean_us, 1e6 * gs['wait_s_max']], dtype=torch.float64, device=comm_dev)
        mx = v.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = v.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, accepted_all = float(mx[0]), float(sm[1])
        gather_us_per_step, slowest_rank_wait_us = round(float(sm[2]) / world, 1), round(float(mx[3]), 1)
    else:
        accepted_all = float(accepted)
    correct = all(seqs[i][P:P + len(truths[gidx[i]])] == truths[gidx[i]][:len(seqs[i]) - P] for i in range(B))
    ctx_timed = float(np.mean(ctxs[c0:c0 + K]))      # mean context of the TIMED steps (the legs below keep extending the sequence)
    ctx_end_timed = eng.n_keys if B == 1 else int(np.mean(eng.slot_keys[:B]))
    native = None
    if B == 1 and not wide and not dist_on and len(seqs[0]) + (BL + 1) * (16 + 24) < max_length - 2 * DL:
        # informational: the same steps through the native loop (la_lookahead_decode, what lookahead_generation() uses when
        # no streamer / processor is attached).  `value` stays on the interpreter loop, which is also what N > 1 runs.
        torch.cuda.synchronize()
        t1 = time.time()
        new, dls_n, edls_n, _, _, _ = eng.decode_native(cache, seqs[0], max_length - 2 * DL, eos_ids=(), decoding_length=DL,
                                                        branch_length=BL, max_query_length=2, idx=gidx[0], max_steps=16)
        dt = time.time() - t1
        seqs[0].extend(new)
        native = {'steps': len(edls_n), 'ms_per_step': round(1e3 * dt / max(len(edls_n), 1), 4),
                  'accepted_tokens_per_sec': round(sum(edls_n) / dt, 2),
                  'equals_greedy': seqs[0][P:P + len(truths[gidx[0]])] == truths[gidx[0]][:len(seqs[0]) - P]}

    # ---- 8(d) fixed draft-tree shape "T64/B8" with the accept sweep: the main chain equals the greedy continuation for a tokens
    sweep = None
    if B == 1 and not wide and not dist_on and not args.pure_random:
        parent, depth, rows64 = fixed_t64b8_tree()
        truth = truths[gidx[0]]
        sweep = {}
        V = shape.vocab
        for a in (0, 3, 6, 12):
            times, ok = [], True
            for _ in range(6):
                k = len(seqs[0]) - P                       # truth[k] is the token after the root
                ids = np.zeros(64, dtype=np.int32)
                ids[0] = seqs[0][-1]
                for j in range(1, 64):
                    d = depth[j]
                    right = truth[k + d - 1]
                    on_main = j <= 12
                    if on_main and d <= a:
                        ids[j] = right
                    else:                                   # a wrong token, distinct from every sibling (siblings differ in j)
                        w = 3 + (right - 3 + 1 + j) % (V - 3)
                        ids[j] = w
                torch.cuda.synchronize(); t1 = time.time()
                toks, _ = eng.step(ids, rows64, mode=0)
                times.append(time.time() - t1)
                ok = ok and toks == truth[k:k + a + 1]
                seqs[0].extend(toks)
                cache.stream_put(toks, branch_length=BL + 1, final=False, idx=gidx[0])
            ms = 1e3 * float(np.mean(times[1:]))
            sweep[f'a={a}'] = {'ms_per_step': round(ms, 4), 'accepted_per_step': a + 1, 'accepted_tokens_per_sec': round((a + 1) / ms * 1e3, 1),
                               'accepted_as_designed': bool(ok)}

    if rank != 0:
        if dist_on:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- roofline ---------------------------------------------------------------------------------------------------
    ctx = ctx_timed                  # the step is priced at the mean context of the timed window, not at where later legs left the cache
    ms_step = 1e3 * elapsed / K
    mean_T = float(np.mean(dls[n0:]))
    mean_acc = float(np.mean(edls[n0:]))
    W_bytes = 2 * shape.n_params_no_embed()
    if B == 1 and not wide:
        # dominant kernel (gate/up GEMM, k_gemm64r<4,SWIGLU,4,8>) from live HIP events on the engine's stream
        ids, rowmask = drafts_for(0)
        prof = eng.profile(ids, rowmask, iters=args.profile_iters)
        gu_ms_step = prof['ms']['gateup'] / max(prof['launches']['gateup'], 1)     # inside the eager step: event packets on both sides
        # the launch duration proper: all layers' gate/up launches back to back inside ONE event pair (kernel + launch boundary)
        gu_ms = eng.profile_gateup(iters=5)
        gu_bytes = 2 * shape.ffn * shape.hidden * 2 + 64 * shape.hidden * 2 + 64 * shape.ffn * 2
        step_bytes = int(algorithmic_bytes(shape, 64, ctx, 64 * shape.vocab * 2))
        gemm_ms = sum(prof['ms'][k] for k in ('qkv', 'o', 'gateup', 'down', 'lm_head'))
        traffic, traffic_src = None, None
        try:        # HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 cannot run inside bench.py)
            pmc = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_latest.json')))
            key = [k for k in pmc['kernels'] if k.startswith('void k_gemm64r<4, 1, 4, 8')][0]      # gate/up launch
            traffic = pmc['kernels'][key]['hbm_bytes_per_launch']
            traffic_src = pmc['source']
        except Exception:
            pass
        fm = floor_model(shape)
        # the launches that are not weight GEMMs (tree attention, the two row kernels per layer, step head / tail, KV commit): their time in
        # the captured step is the graph step minus the GEMM classes, the latter scaled from the eager profile (whose event packets
        # between all kernels inflate every class alike) to the graph step
        nongemm_ms = max(ms_step - gemm_ms * ms_step / max(prof['ms_step'], 1e-9), 0.0)
        fm['nongemm_ms_per_step'] = round(nongemm_ms, 4)
        fm['frac_of_peak_if_nongemm_were_free'] = round(step_bytes / (fm['gemm_floor_ms_per_step'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        fm['frac_of_peak_at_floor'] = round(step_bytes / ((fm['gemm_floor_ms_per_step'] + nongemm_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        roofline = {
            'bound': 'hbm', 'kernel': 'k_gemm64r<4,EPI_SWIGLU,4,8> (gate/up projection + fused SwiGLU, %d launches/step)' % shape.n_layers,
            'achieved': round(gu_bytes / (gu_ms * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(gu_bytes / (gu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), 'traffic': traffic, 'traffic_source': traffic_src,
            'frac_of_achievable': round(gu_bytes / (gu_ms * 1e-3) / 1e9 / HBM_ACHIEVABLE_GBS, 4),
            'bytes_per_launch': gu_bytes, 'ms_per_launch': round(gu_ms, 5),
            'timing': 'HIP events on the engine stream: %d launches (one per layer, own weights) back to back per event pair, 5 passes; '
                      'inside the eager step, with event packets between all kernels, the same launch reads %.5f ms; rocprofv3 kernel '
                      'average of the same command: the newest profiles/r*_profile_raw.txt' % (shape.n_layers, gu_ms_step),
            'ms_per_launch_in_eager_step': round(gu_ms_step, 5),
            # the same launch against the matrix-core roof: 64-row trees keep the kernel far below the MFMA ridge (HBM-bound by design)
            'mfma': {'flops_per_launch': 2 * 2 * shape.ffn * shape.hidden * 64,
                     'achieved_TFLOPs': round(2 * 2 * shape.ffn * shape.hidden * 64 / (gu_ms * 1e-3) / 1e12, 1),
                     'peak_TFLOPs': 2500.0, 'frac': round(2 * 2 * shape.ffn * shape.hidden * 64 / (gu_ms * 1e-3) / 1e12 / 2500.0, 4)},
            'verify_step': {'algorithmic_bytes': step_bytes, 'context_keys_mean_of_timed_steps': round(ctx, 1),
                            'ms_graph_step': round(ms_step, 4),
                            'achieved_GBps': round(step_bytes / (ms_step * 1e-3) / 1e9, 1),
                            'frac': round(step_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            'frac_of_achievable': round(step_bytes / (ms_step * 1e-3) / 1e9 / HBM_ACHIEVABLE_GBS, 4),
                            'floor_model': fm,
                            'ms_eager_step_events': round(prof['ms_step'], 4),
                            'ms_by_class_events': {k: round(v, 4) for k, v in prof['ms'].items()},
                            'all_gemm_GBps_events': round(2 * shape.n_params_no_embed() / (gemm_ms * 1e-3) / 1e9, 1)},
        }
    else:
        # M = 64 B rows per step: near / past the ridge -> the step is priced against BOTH roofs from its wall time;
        # per-kernel durations of the same step: profiles/r02_mblock_kernel_stats_*.txt (rocprofv3 --kernel-trace --stats)
        kv_tok = 2 * shape.n_layers * shape.n_kv_heads * shape.head_dim * 2
        RB_ = (DL + 63) // 64 if wide else 1        # 64-row blocks per sequence (wide trees: the mean tree occupies up to DL rows)
        step_bytes = int(W_bytes + kv_tok * ctx * B + kv_tok * 64 * B * RB_ + 64 * B * RB_ * (shape.hidden * 2 + 8) + 64 * B * RB_ * shape.vocab * 2)
        active = shape.n_params_no_embed() - (shape.n_layers * 3 * shape.ffn * shape.hidden * max(shape.n_experts - shape.top_k, 0) if shape.n_experts else 0)
        step_flops = 2.0 * active * 64 * B * RB_ + 4.0 * shape.n_layers * shape.n_heads * shape.head_dim * 64 * B * RB_ * (ctx + 64)      # MoE: the top-k experts of a row
        hbm_frac = step_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS
        mfma_frac = step_flops / (ms_step * 1e-3) / 1e12 / 2500.0
        bound = 'mfma' if mfma_frac >= hbm_frac else 'hbm'
        traffic, traffic_src = None, None
        try:        # HBM bytes of one steady verify step of this configuration from the committed PMC passes (scripts/gpu_prof_secondary.sh)
            pmc = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_secondary.json')))
            if not args.layers and not wide:
                traffic = pmc['configs']['%s_b%d' % (args.model, B)]['hbm_bytes_per_step']
                traffic_src = pmc['source']
        except Exception:
            pass
        roofline = {
            'bound': bound, 'kernel': 'whole multi-block verify step (k_gemm_fat / k_gemm_wide / k_gemm_mb + k_tree_attn_mb), M = %d rows%s' % (64 * B * RB_, '' if B <= 8 else ' in %d passes over the weights (priced as ONE)' % ((B + 7) // 8)),
            'achieved': round(step_flops / (ms_step * 1e-3) / 1e12, 1) if bound == 'mfma' else round(step_bytes / (ms_step * 1e-3) / 1e9, 1),
            'peak': 2500.0 if bound == 'mfma' else HBM_PEAK_GBS, 'unit': 'TFLOP/s' if bound == 'mfma' else 'GB/s',
            'frac': round(max(mfma_frac, hbm_frac), 4), 'traffic': traffic, 'traffic_source': traffic_src,
            'traffic_note': 'HBM bytes of ONE verify step (FETCH_SIZE / WRITE_SIZE passes, all kernels from one k_build_inputs_mb to the next)',
            'hbm': {'algorithmic_bytes': step_bytes, 'achieved_GBps': round(step_bytes / (ms_step * 1e-3) / 1e9, 1), 'frac': round(hbm_frac, 4)},
            'mfma': {'flops': step_flops, 'achieved_TFLOPs': round(step_flops / (ms_step * 1e-3) / 1e12, 1), 'frac': round(mfma_frac, 4)},
        }
    cpu = None
    if want_cpu:
        cpu = cpu_baseline_leg(shape, sd_cpu, prompts[0], copies0, BL, DL, verify_steps=args.cpu_steps)
    gather_transport = gather.transport if gather is not None else None
    rccl_ranks = world if (gather_transport is not None and ('rccl' in gather_transport or 'nccl' in gather_transport)) else 0
    if dist_on:                      # the job's collectives are over: leave the group before any follow-up job is started
        dist.barrier()
        dist.destroy_process_group()
    secondary = None
    sec_ok = (not args.layers) or args.secondary_layers > 0
    if world == 1 and B == 1 and args.model == '7b' and sec_ok and args.secondary and not os.environ.get('BENCH_IS_SECONDARY'):
        secondary = secondary_legs(args.secondary, layers=args.secondary_layers)
    multi = args.secondary_multi if args.secondary_multi is not None else ('13b:4' if world == 8 else '')
    if world > 1 and B == 1 and args.model == '7b' and sec_ok and multi and not os.environ.get('BENCH_IS_SECONDARY'):
        # BASELINE config 4 (and any other listed leg) as its own N-rank job on the same GPUs: the other ranks of this job have
        # left the group and are exiting; 288 GB per GPU hold both models, so nothing has to be freed first
        secondary = secondary_legs(multi, gpus=world, layers=args.secondary_layers)
    gen = accepted_all / max(world, 1)
    out = {
        'metric': 'accepted_tokens_per_sec', 'value': round(accepted_all / elapsed, 2), 'unit': 'tokens/s',
        'n_gpus': world, 'steps': K, 'warmup': W, 'ms_per_step': round(ms_step, 4), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': model_name + f' {args.dtype} bs={B}/GPU lookahead verify loop, {DL}-token draft tree per sequence (hier, decoding_length={DL}, '
                               f'branch_length={BL}), synthetic weights, {P}-token prompts (notes.workload)',
                   'model': model_name, 'n_layers': shape.n_layers, 'n_layers_truncated': bool(args.layers), 'prompt_len': P, 'rho': args.rho, 'copies': args.copies,
                   'parallelism': f'batch-shard x{world}, {B} sequence(s) per GPU', 'sequences': NSEQ,
                   'kv_cache': (f'ring of {eng.shape.sliding_window} + one step of rows per sequence (sliding window)' if kv_ring else 'linear, max_length keys per sequence'),
                   'gather_mode': None if not dist_on else ('strict' if args.strict_gather else 'split-phase'),
                   'gather_transport': gather_transport, 'rccl_ranks': rccl_ranks,
                   'gather_us_per_step': gather_us_per_step, 'slowest_rank_wait_us': slowest_rank_wait_us,
                   'trie_update': ('device' if (dev_trie is not None and dev_trie.put_vocab) else
                                   ('gathered, ' + gather.mode) if dist_on else
                                   'deferred' if (B > 1 and overlap_put and not wide and dev_trie is None) else 'ref-order'),
                   'draft_retrieval': (('device' + (' (workgroup per query)' if args.trie_algo == 'wg' else ' (wavefront per query)') + ('' if not args.unchained_trie else ', unchained') + ('' if dev_trie.put_vocab else ', host patches')) if dev_trie is not None else 'host'),
                   'device_trie_stats': dev_trie.stats if dev_trie is not None else None,
                   'mean_accept_len': round(mean_acc, 3), 'mean_draft_len': round(mean_T, 2),
                   'verify_steps_per_sec': round(K * world / elapsed, 2), 'context_at_end': ctx_end_timed,
                   'context_mean_timed': round(ctx_timed, 1),
                   'trie_query_ms_mean': round(1e3 * float(np.mean(qts[q0:])), 4),
                   'lookahead_equals_greedy': bool(correct), 'plain_greedy_tokens_per_sec': round(B * len(truth_own[0]) / t_greedy, 2),
                   'native_loop': native,
                   'idle_window_prefetch_kib': _pf_setting()[0], 'idle_window_prefetch_delay': _pf_setting()[1],
                   'gateup_tail_prefetch_kib': _pf_setting()[2],
                   'speed_incl_prefill': {'prefill_ms': round(1e3 * t_prefill, 3), 'prefill_tokens': P * B,
                                          'generated_tokens': int(gen), 'decode_s': round(elapsed, 4),
                                          'tokens_per_sec': round(gen / (t_prefill + elapsed), 2),
                                          'at_256_new_tokens': round(256 * B / (t_prefill + 256 * B / max(accepted_all / world / elapsed, 1e-9)), 2),
                                          'note': 'reference headline definition: generated tokens / (prefill + decode) wall time (benchmarks/benchmark.py:277-328), per GPU'},
                   'fixed_tree_sweep': sweep},
        'roofline': roofline, 'cpu_baseline': cpu,
    }
    if secondary is not None:
        out['secondary'] = secondary
    # the full record first (not the last line), then the compact contract line the driver parses and whose tail it keeps
    detail = json.dumps(out)
    print('BENCH_DETAIL ' + detail, flush=True)
    try:
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        tag = '%s_b%d_n%d%s' % (args.model, B, world, '_dev' if dev_trie is not None else '')
        with open(os.path.join(ROOT, 'gpurun_out', 'bench_detail_%s.json' % tag), 'w') as f:
            f.write(detail + '\n')
    except OSError:
        pass
    print(json.dumps(compact_record(out)), flush=True)


if __name__ == '__main__':
    main()
