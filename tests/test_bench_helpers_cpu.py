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


def test_cpu_baseline_leg_runs_the_oracle_loop_on_a_small_shape():
    """the CPU leg = the oracle's own lookahead loop (full model, own drafts, own accept-len), here on the tiny decisive model"""
    import torch
    from tests.tiny_model import tiny_decisive_weights, tiny_shape
    s = tiny_shape()
    sd = tiny_decisive_weights(0, torch.bfloat16)
    prompt = bench.phrase_prompt(7, 40, s.vocab)
    # greedy continuation of the permutation LM = follow lm_head[pi(t)] = embed[t]; a few noisy copies warm the trie
    from oracle import llama_oracle as lo
    model = lo.OracleLlama(s, sd)
    seq = list(prompt)
    lg, past = model.forward(torch.tensor(seq), torch.tril(torch.ones((40, 40), dtype=torch.long)), None)
    for _ in range(60):
        t = int(lg[-1].float().argmax())
        seq.append(t)
        lg, past = model.forward(torch.tensor([t]), torch.ones((1, len(seq)), dtype=torch.long), past)
    copies = bench.noisy_copies(prompt[-2:] + seq[40:], 6, 0.3, s.vocab, seed=99)
    r = bench.cpu_baseline_loop(s, sd, prompt, copies, 12, 64, verify_steps=4, threads=2)
    assert r['kind'] == 'port' and r['unit'] == 'tokens/s' and r['value'] > 0 and r['cores'] == 2 and r['verify_steps'] == 4
    assert r['ms_per_step'] > 0 and r['mean_accept_len'] > 1.0 and 'cpu_model' in r and 'sample' in r


def test_fixed_t64b8_tree_is_the_survey_shape():
    """T = 64, 8 leaves all at depth 12, row0 = 0x1, row12 = 0x1fff (main chain), first fork row = 0x23ff."""
    parent, depth, rows = bench.fixed_t64b8_tree()
    leaves = [i for i in range(64) if i not in parent]
    assert len(parent) == 64 and len(leaves) == 8 and all(depth[i] == 12 for i in leaves)
    assert int(rows[0]) == 0x1 and int(rows[12]) == 0x1fff and int(rows[13]) == 0x23ff
    assert all(int(rows[i]).bit_count() == depth[i] + 1 for i in range(64))


def test_committed_pmc_summary_is_what_bench_reads():
    pmc = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_latest.json')))
    keys = [k for k in pmc['kernels'] if k.startswith('void k_gemm64r<4, 1, 4, 8')]
    assert len(keys) == 1
    e = pmc['kernels'][keys[0]]
    s7 = LlamaShape.llama2_7b()
    algorithmic = 2 * s7.ffn * s7.hidden * 2 + 64 * s7.hidden * 2 + 64 * s7.ffn * 2
    assert algorithmic <= e['hbm_bytes_per_launch'] <= 1.10 * algorithmic      # nothing is re-read from HBM


def test_self_launch_command_and_clean_environment(monkeypatch):
    """`python bench.py --gpus N` without rank variables re-executes itself under torch.distributed.run (one process per GPU,
    rendezvous on 127.0.0.1) — the shape of the driver's own N > 1 launch; a follow-up job started from inside a rank must not
    inherit that rank's rendezvous variables."""
    import sys
    import bench
    cmd = bench.self_launch_cmd(['--gpus', '8', '--steps', '5'], 8, 29511)
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd and '--nproc-per-node=8' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[cmd.index('--master-port') + 1] == '29511'
    assert cmd[-4:] == ['--gpus', '8', '--steps', '5'] and cmd[-5].endswith('bench.py')
    monkeypatch.setenv('RANK', '3'); monkeypatch.setenv('WORLD_SIZE', '8'); monkeypatch.setenv('MASTER_PORT', '1')
    env = bench.clean_rank_env()
    assert not any(k in env for k in ('RANK', 'WORLD_SIZE', 'MASTER_PORT', 'LOCAL_RANK'))
    assert env['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    p = bench._free_port()
    assert 1024 < p < 65536


def test_self_launch_runs_two_gloo_ranks_end_to_end(tmp_path):
    """The self-launch path itself, on CPU: a stub 'bench' with the same launcher spawns 2 ranks through torch.distributed.run,
    they form a gloo group on 127.0.0.1 and rank 0 prints ONE line."""
    import subprocess
    import sys
    import bench
    stub = tmp_path / 'stub.py'
    stub.write_text(
        "import os, json, torch.distributed as dist\n"
        "dist.init_process_group('gloo')\n"
        "import torch\n"
        "t = torch.tensor([dist.get_rank() + 1]); dist.all_reduce(t)\n"
        "if dist.get_rank() == 0: print(json.dumps({'world': dist.get_world_size(), 'sum': int(t)}), flush=True)\n"
        "dist.barrier(); dist.destroy_process_group()\n")
    cmd = bench.self_launch_cmd([], 2, bench._free_port())
    cmd[cmd.index(os.path.abspath(bench.__file__))] = str(stub)
    r = subprocess.run(cmd, env=bench.clean_rank_env(), capture_output=True, text=True, timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1 and json.loads(lines[0]) == {'world': 2, 'sum': 3}


def test_host_batch_drafts_equals_per_sample_queries():
    """bench.host_batch_drafts: the one-call batch form (la_cache_bat_get_packed) returns exactly the per-sample hier_get_packed drafts
    the timed loop used before (same budget, same min_output_size, per-sequence input slot), falls back to per-sample calls when a
    sequence's branch length is clamped, and substitutes the last token for an empty draft."""
    import numpy as np
    import bench
    from painlessinferenceacceleration_amd.lookahead_cache import LookaheadCache
    rs = np.random.RandomState(8)
    V = 400
    cache = LookaheadCache(eos_ids=[None])
    seqs = [rs.randint(3, V, size=120).tolist() for _ in range(6)]
    for b, s_ in enumerate(seqs):
        for _ in range(4):
            cache.put([t if rs.rand() > 0.15 else int(rs.randint(3, V)) for t in s_], branch_length=13, mode='output', idx=-1)
        cache.put(s_[:60], branch_length=13, mode='input', idx=10 + b)
    idxs = [10 + b for b in range(6)]

    def per_sample(tails, ubls):
        out = []
        for t, ix, u in zip(tails, idxs, ubls):
            ids, rm, _, _ = cache.hier_get_packed(t, decoding_length=64, branch_length=u, min_input_size=0, min_output_size=32,
                                                  mode='mix', idx=ix)
            out.append((ids.copy(), rm.copy()) if len(ids) else (np.asarray(t[-1:], dtype=np.int32), np.array([1], dtype=np.uint64)))
        return out

    hits = 0
    for trial in range(40):
        p = int(rs.randint(2, 100))
        tails = [s_[p - 2:p] for s_ in seqs]
        if trial % 5 == 0:
            tails[0] = [V + 5, V + 6]                     # unknown tokens: no tree -> the last token alone
        ubls = [12] * 6 if trial % 3 else [12, 12, 7, 12, 3, 12]
        got = bench.host_batch_drafts(cache, tails, idxs, 64, 12, ubls)
        want = per_sample(tails, ubls)
        for g, w in zip(got, want):
            assert g[0].tolist() == w[0].tolist() and g[1].tolist() == w[1].tolist()
            hits += len(g[0]) > 8
    assert hits > 60
    assert bench.host_batch_drafts(cache, [[V + 5, V + 6]], [10], 64, 12, [12])[0][0].tolist() == [V + 6]


def test_cpu_baseline_reference_leg_equals_the_port_when_the_reference_is_importable():
    """`cpu_baseline.kind = "reference"` (oracle/reference_cpu.py): the reference's own lookahead_generation + LookaheadCache + LlamaForCausalLM,
    imported in place, must emit the tokens and the per-step draft / accept lengths of the port (oracle loop) on the same model, prompt and trie warm-up —
    the two legs are interchangeable as the reported baseline.  Skipped where /root/reference is absent (the GPU box)."""
    import pytest
    import torch
    from oracle import reference_cpu
    if not reference_cpu.reference_available():
        pytest.skip('/root/reference not present')
    from oracle import llama_oracle as lo
    from oracle.trie_oracle import TrieOracle
    from tests.tiny_model import tiny_decisive_weights, tiny_shape
    s = tiny_shape()
    sd = tiny_decisive_weights(0, torch.float32)
    prompt = bench.phrase_prompt(7, 40, s.vocab)
    model = lo.OracleLlama(s, sd)
    seq = list(prompt)
    lg, past = model.forward(torch.tensor(seq), torch.tril(torch.ones((40, 40), dtype=torch.long)), None)
    for _ in range(60):
        t = int(lg[-1].float().argmax())
        seq.append(t)
        lg, past = model.forward(torch.tensor([t]), torch.ones((1, len(seq)), dtype=torch.long), past)
    copies = bench.noisy_copies(prompt[-2:] + seq[40:], 6, 0.3, s.vocab, seed=99)
    ref = reference_cpu.reference_lookahead_loop(s, sd, prompt, copies, 12, 64, verify_steps=4, threads=2, dtype=torch.float32)
    assert ref['kind'] == 'reference' and ref['verify_steps'] == 4 and ref['value'] > 0 and ref['cores'] == 2
    cache = TrieOracle(eos_ids=[None])
    for c in copies:
        cache.put(c, branch_length=13, mode='output', idx=-1)
    port = lo.lookahead_generate(model, cache, prompt, len(prompt) + 5 * 13 + 2, eos_token_id=None, decoding_length=64, branch_length=12, max_steps=5)
    n = len(ref['tokens'])
    assert n > 4 and port['sequences'][40:40 + n] == ref['tokens'] == seq[40:40 + n]
    assert ref['mean_accept_len'] == round(sum(port['edls'][1:]) / 4, 3) and ref['mean_draft_len'] == round(sum(port['dls'][1:]) / 4, 2)


def test_compact_record_keeps_the_contract_fields_and_fits_the_drivers_captured_tail():
    """The driver keeps the last 8 KB of stdout: the final JSON line (bench.compact_record) must carry the contract fields, `roofline`,
    `cpu_baseline` and ALL secondary legs in compact form within that budget; the prose lives once in `notes`, the full record in the
    BENCH_DETAIL line before it."""
    import json
    roof1 = {'bound': 'hbm', 'kernel': 'k_gemm64r<4,EPI_SWIGLU,4,8> (gate/up projection + fused SwiGLU, 32 launches/step)', 'achieved': 4982.4, 'peak': 8000.0, 'unit': 'GB/s',
             'frac': 0.6228, 'traffic': 190512345, 'traffic_source': 'x' * 300, 'frac_of_achievable': 0.79, 'bytes_per_launch': 182288384, 'ms_per_launch': 0.03658,
             'timing': 'y' * 400, 'mfma': {'flops_per_launch': 1, 'achieved_TFLOPs': 300.0, 'peak_TFLOPs': 2500.0, 'frac': 0.12},
             'verify_step': {'algorithmic_bytes': 13567000000, 'ms_graph_step': 3.66, 'achieved_GBps': 3707.0, 'frac': 0.463, 'frac_of_achievable': 0.59,
                             'floor_model': {'gemm_floor_ms_per_step': 2.9, 'nongemm_ms_per_step': 0.95, 'frac_of_peak_if_nongemm_were_free': 0.58, 'frac_of_peak_at_floor': 0.44, 'model': 'z' * 300},
                             'ms_eager_step_events': 4.0, 'ms_by_class_events': {'qkv': 0.74, 'o': 0.33, 'gateup': 1.17, 'down': 0.75, 'lm_head': 0.05, 'attn': 0.4, 'other': 0.5},
                             'all_gemm_GBps_events': 4400.0}}
    cfg = {'workload': 'Llama-2-7B bf16 bs=1/GPU lookahead verify loop, 64-token draft tree per sequence (hier, decoding_length=64, branch_length=12), synthetic weights, 512-token prompts (notes.workload)',
           'model': 'Llama-2-7B', 'n_layers': 32, 'n_layers_truncated': False, 'prompt_len': 512, 'rho': 0.3, 'copies': 8, 'parallelism': 'batch-shard x1, 1 sequence(s) per GPU', 'sequences': 1,
           'kv_cache': 'linear, max_length keys per sequence', 'gather_mode': None, 'gather_transport': None, 'rccl_ranks': 0, 'gather_us_per_step': None, 'slowest_rank_wait_us': None,
           'trie_update': 'ref-order', 'draft_retrieval': 'host', 'device_trie_stats': None, 'mean_accept_len': 6.45, 'mean_draft_len': 60.1, 'verify_steps_per_sec': 273.0,
           'context_at_end': 770, 'context_mean_timed': 700.0, 'trie_query_ms_mean': 0.02, 'lookahead_equals_greedy': True, 'plain_greedy_tokens_per_sec': 280.0,
           'native_loop': {'steps': 16, 'ms_per_step': 3.7, 'accepted_tokens_per_sec': 1700.0, 'equals_greedy': True}, 'idle_window_prefetch_kib': 0,
           'speed_incl_prefill': {'prefill_ms': 10.9, 'prefill_tokens': 512, 'generated_tokens': 130, 'decode_s': 0.07, 'tokens_per_sec': 1500.0, 'at_256_new_tokens': 1600.0, 'note': 'n' * 200},
           'fixed_tree_sweep': {'a=%d' % a: {'ms_per_step': 3.7, 'accepted_per_step': a + 1, 'accepted_tokens_per_sec': 500.0, 'accepted_as_designed': True} for a in (0, 3, 6, 12)}}
    cpu = {'value': 25.2, 'unit': 'tokens/s', 'cores': 16, 'kind': 'port', 'ms_per_step': 262.0, 'dtype': 'bfloat16', 'cpu_model': 'AMD EPYC 9575F 64-Core Processor', 'host_cores': 128,
           'verify_steps': 5, 'mean_accept_len': 6.6, 'mean_draft_len': 58.0, 'prefill_s': 3.0, 'wall_s': 20.0, 'sample': 's' * 330}
    legs = []
    for name, model, dr in (('mistral:8', 'Mistral-7B', 'host'), ('mistral:8:dev', 'Mistral-7B', 'device'), ('13b:4', 'Llama-2-13B', 'host'), ('mixtral:4', 'Mixtral-8x7B', 'host'), ('13b:1', 'Llama-2-13B', 'host'),
                            ('7b:16', 'Llama-2-7B', 'host')):
        leg_full = {'metric': 'accepted_tokens_per_sec', 'value': 5400.12, 'unit': 'tokens/s', 'n_gpus': 1, 'steps': 24, 'warmup': 4, 'ms_per_step': 9.6234,
                    'config': dict(cfg, model=model, sequences=8, draft_retrieval=dr, trie_update='deferred'),
                    'roofline': {'bound': 'mfma', 'kernel': 'k' * 120, 'achieved': 776.1, 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': 0.3104, 'traffic': 28400000000, 'traffic_note': 't' * 150,
                                 'hbm': {'algorithmic_bytes': 14970000000, 'achieved_GBps': 1555.0, 'frac': 0.194}, 'mfma': {'flops': 7.46e12, 'achieved_TFLOPs': 776.1, 'frac': 0.3104}}}
        legs.append(bench.compact_leg(bench.compact_record(dict(leg_full, higher_is_better=True, scaling='weak', vs_baseline=None, dtype='bf16', data='synthetic', cpu_baseline=None)), name, 12.3))
    out = {'metric': 'accepted_tokens_per_sec', 'value': 1762.5, 'unit': 'tokens/s', 'n_gpus': 1, 'steps': 20, 'warmup': 5, 'ms_per_step': 3.6596, 'higher_is_better': True,
           'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic', 'config': cfg, 'roofline': roof1, 'cpu_baseline': cpu, 'secondary': legs}
    comp = bench.compact_record(out)
    line = json.dumps(comp)
    assert len(line) < 8000, len(line)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in comp
    assert {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'} <= set(comp['roofline']) and comp['roofline']['verify_step']['frac'] == 0.463
    assert {'value', 'unit', 'cores', 'kind', 'sample'} <= set(comp['cpu_baseline'])
    assert len(comp['secondary']) == 6 and all(leg['traffic_ratio'] == round(28400000000 / 14970000000, 3) and leg['equals_greedy'] is True for leg in comp['secondary'])
    assert [leg['draft_retrieval'] for leg in comp['secondary'][:2]] == ['host', 'device'] and comp['secondary'][0]['model'] == 'Mistral-7B'
    assert comp['config']['workload'].startswith('Llama-2-7B') and 'workload' in comp['notes']
