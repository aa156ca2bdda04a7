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
Multi-GPU: one independent sequence per rank (batch sharding, weak scaling); the only exchange is the
per-step all-gather of accepted tokens over RCCL so that every rank's trie replica sees every sequence.  The gather is
split-phase: started after step k, collected while the GPU runs step k+1, then applied for all ranks in global batch-index
order (replicas stay identical; a step's tokens reach the drafts one step later; emitted tokens are unaffected).
"""
import argparse
import json
import os
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


def cpu_baseline(shape, T, ctx, accept_len, budget_s=20.0):
    """The oracle's verify forward (oracle/llama_oracle.py, a port of the reference's transformers CPU path) timed
    on this box's host cores at the full Llama-2-7B layer shape.  Bounded sample (~10-30 s of CPU work): one decoder
    layer + lm_head are timed at T tree tokens / ctx context for each (dtype, thread count) candidate with a per-trial
    cap, the fastest candidate is kept (generous to the CPU: bf16 falls into a slow oneDNN path on hosts without
    AMX), and the step time is composed as t(lm_head part) + n_layers * t(layer); accepted tok/s = steps/s x the
    mean accept-len measured on the GPU run."""
    from oracle import llama_oracle as lo
    from painlessinferenceacceleration_amd.llama_engine import LlamaShape, random_weights
    # one decoder layer is timed on a model with a 64-entry vocabulary (its lm_head is negligible), the embedding + final
    # norm + full lm_head on a 0-layer model: both terms are measured directly (no difference of two noisy timings)
    one = LlamaShape(1, shape.hidden, shape.n_heads, shape.n_kv_heads, shape.ffn, 64, shape.rms_eps)
    zero = LlamaShape(0, shape.hidden, shape.n_heads, shape.n_kv_heads, shape.ffn, shape.vocab, shape.rms_eps)
    sd1_bf16 = random_weights(one, seed=0, device='cpu')
    sd0_bf16 = random_weights(zero, seed=0, device='cpu')
    rs = np.random.RandomState(0)
    hd = shape.head_dim
    ids1 = torch.tensor(rs.randint(3, 64, size=T).tolist())
    ids0 = torch.tensor(rs.randint(3, shape.vocab, size=T).tolist())
    mask = torch.cat([torch.ones((T, ctx), dtype=torch.long), torch.tril(torch.ones((T, T), dtype=torch.long))], 1)
    ncpu = os.cpu_count() or 1
    cands = []
    for dt in (torch.bfloat16, torch.float32):
        for nt in sorted(set([min(ncpu, 8), min(ncpu, 32), min(ncpu, 96)])):
            cands.append((dt, nt))
    t_start = time.time()
    best = None
    tried = []
    for dt, nt in cands:
        if time.time() - t_start > budget_s:
            break
        torch.set_num_threads(nt)
        past = [(torch.randn(shape.n_kv_heads, ctx, hd).to(dt), torch.randn(shape.n_kv_heads, ctx, hd).to(dt))]
        m1 = lo.OracleLlama(one, {k: v.to(dt) for k, v in sd1_bf16.items()})
        m0 = lo.OracleLlama(zero, {k: v.to(dt) for k, v in sd0_bf16.items()})

        def timed(model, ids, p, cap):
            t0 = time.time(); model.forward(ids, mask, p); first = time.time() - t0       # warm-up (page-in, kernels)
            if first > cap:
                return first
            n, t0 = 0, time.time()
            while n < 5 and time.time() - t0 < cap:
                model.forward(ids, mask, p); n += 1
            return (time.time() - t0) / max(n, 1)
        t_layer = timed(m1, ids1, past, 2.5)
        t_head = timed(m0, ids0, [], 1.0)
        step = t_head + shape.n_layers * t_layer
        tried.append(f"{str(dt).split('.')[-1]}x{nt}t:{step * 1e3:.0f}ms")
        if best is None or step < best[0]:
            best = (step, dt, nt)
    step, dt, nt = best
    return {'value': round(accept_len / step, 3), 'unit': 'tokens/s', 'cores': nt, 'kind': 'port',
            'ms_per_step': round(step * 1e3, 1), 'dtype': str(dt).split('.')[-1],
            'sample': f'oracle verify forward at the benchmarked layer shape (hidden {shape.hidden}, ffn {shape.ffn}, {shape.n_layers} layers), T={T}, ctx={ctx}: 1 decoder layer and embedding+lm_head timed separately '
                      f'(<=5 runs each, candidates {" ".join(tried)}), step = lm_head part + {shape.n_layers} x layer; '
                      f'accepted tok/s = steps/s x GPU-run mean accept-len {accept_len:.2f}'}


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
    ap.add_argument('--model', choices=['7b', '13b'], default='7b', help='7b = the BASELINE metric; 13b = the config-4 model shape')
    ap.add_argument('--profile-iters', type=int, default=3)
    ap.add_argument('--attn-split', type=int, default=0, help='key splits of the tree-attention kernel (0 = engine default 8)')
    ap.add_argument('--fuse', type=int, default=0, help='engine cfg.fuse bits (opt-in in-kernel norm->GEMM fusion; 0 = separate kernels)')
    args = ap.parse_args()

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
            dist.init_process_group('nccl', device_id=torch.device(f'cuda:{local_rank}'))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    dev = 'cuda:0' if os.environ.get('BENCH_SHARE_GPU') else f'cuda:{local_rank}'
    comm_dev = dev if os.environ.get('BENCH_DIST_BACKEND', 'nccl') == 'nccl' else 'cpu'
    torch.cuda.set_device(dev)

    from painlessinferenceacceleration_amd.llama_engine import LlamaShape
    from painlessinferenceacceleration_amd.lookahead_cache import LookaheadCache
    from painlessinferenceacceleration_amd.modeling_llama import LlamaForCausalLM

    shape = LlamaShape.llama2_13b() if args.model == '13b' else LlamaShape.llama2_7b()
    model_name = 'Llama-2-13B' if args.model == '13b' else 'Llama-2-7B'
    if args.layers:
        shape.n_layers = args.layers
    K, W, P = args.steps, args.warmup, args.prompt_len
    BL, DL = 12, 64
    n_truth = (K + W) * (BL + 1) + 8
    max_length = P + n_truth + 2 * DL
    model = LlamaForCausalLM.random_init(shape, seed=0, device=dev, max_length=max_length, eos_token_id=None,
                                         decisive=not args.pure_random, fuse=args.fuse, attn_split=args.attn_split)
    eng = model.engine

    # ---- untimed set-up: prompt, ground-truth continuation (plain greedy on the same engine), trie warm-up
    prompts = [phrase_prompt(1234 + r, P, shape.vocab) for r in range(world)]
    prompt = prompts[rank]
    t0 = time.time()
    truth = model.greedy_search(torch.tensor([prompt]), P + n_truth, eos_token_id=None)[0].tolist()[P:]
    t_greedy = time.time() - t0
    cache = LookaheadCache(eos_ids=[None])
    model.lookahead_cache = cache
    if dist_on:                                  # every replica is warmed with every rank's (noisy) answers
        tt = torch.tensor(truth, dtype=torch.int32, device=comm_dev)
        allt = [torch.empty_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)
        truths = [x.cpu().tolist() for x in allt]
    else:
        truths = [truth]
    for r in range(world):
        for c in noisy_copies(prompts[r][-2:] + truths[r], args.copies, args.rho, shape.vocab, seed=99 + r):
            cache.put(c, branch_length=BL + 1, mode='output', idx=-1)

    # ---- the measured loop ----------------------------------------------------------------------------------
    seq = list(prompt)
    cache.put(seq[1:], branch_length=BL + 1, mode='input', idx=rank)
    eng.reset()
    seq.append(eng.prefill(seq))
    if os.environ.get('BENCH_DEBUG'):
        print(f'[debug] prefill tok {seq[-1]} truth0 {truth[0]} nkeys {eng.n_keys}', file=sys.stderr, flush=True)
    gather = None
    if dist_on:
        from painlessinferenceacceleration_amd.distributed import AcceptedTokenGather
        gather = AcceptedTokenGather(comm_dev)
    pending = [False]
    edls, dls, qts = [], [], []

    def one_step():
        tq = time.time()
        ubl = min(BL, max_length - len(seq) - 1)
        ids, rowmask, _, _ = cache.hier_get_packed(seq[-2:], decoding_length=DL, branch_length=ubl, min_input_size=0,
                                                   min_output_size=DL // 2, mode='mix', idx=rank)
        qts.append(time.time() - tq)
        if os.environ.get('BENCH_DEBUG'):
            t1 = time.time()
            cache.hier_get_packed(seq[-2:], decoding_length=DL, branch_length=ubl, min_input_size=0, min_output_size=DL // 2,
                                  mode='mix', idx=rank)
            print(f'[debug] query {1e3 * qts[-1]:.3f} ms, repeated {1e3 * (time.time() - t1):.3f} ms, T {len(ids)} stats {cache.stats()}',
                  file=sys.stderr, flush=True)
        eng.step_async(ids, rowmask, mode=0)
        if pending[0]:       # N > 1: the previous step's gather + every rank's trie update run while the GPU verifies
            gather.finish_into_trie(cache, BL)
            pending[0] = False
        toks, _ = eng.step_finish()
        if os.environ.get('BENCH_DEBUG') and len(edls) < 6:
            k = len(seq) - P
            print(f'[debug] step {len(edls)} ctx {len(seq)} T {len(ids)} ids {ids[:5].tolist()} -> toks {toks[:6]} '
                  f'truth {truth[k:k + 6]} prev-truth {truth[max(k - 2, 0):k]}', file=sys.stderr, flush=True)
        seq.extend(toks)
        dls.append(len(ids)); edls.append(len(toks))
        if dist_on:
            # split-phase RCCL all-gather: started now, collected during the next verify step, then applied for ALL ranks
            # (this one included) in global batch-index order, so every trie replica goes through the same sequence of
            # inserts.  Tokens are unaffected (verification is lossless); drafts see a step's tokens one step later.
            gather.begin(toks)
            pending[0] = True
        else:
            cache.stream_put(toks, branch_length=BL + 1, final=False, idx=rank)

    import gc
    gc.collect()
    gc.freeze()          # a generation-2 collection over torch's import graph costs tens of ms on the host thread that
                         # drives the loop; the serving loop allocates nothing that needs cycle collection
    for _ in range(W):
        one_step()
    n0 = len(edls)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(K):
        one_step()
    if pending[0]:
        gather.finish_into_trie(cache, BL)
        pending[0] = False
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    elapsed = time.time() - t0
    accepted = int(sum(edls[n0:]))
    if dist_on:
        v = torch.tensor([elapsed, float(accepted)], dtype=torch.float64, device=comm_dev)
        mx = v.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = v.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, accepted_all = float(mx[0]), float(sm[1])
    else:
        accepted_all = float(accepted)
    correct = seq[P:P + len(truth)] == truth[:len(seq) - P]       # lookahead output == plain greedy output
    native = None
    if not dist_on and len(seq) + (BL + 1) * 16 < max_length:
        # informational: the same steps through the native loop (la_lookahead_decode, what lookahead_generation() uses when
        # no streamer / processor is attached).  `value` stays on the interpreter loop, which is also what N > 1 runs.
        torch.cuda.synchronize()
        t1 = time.time()
        new, dls_n, edls_n, _, _, _ = eng.decode_native(cache, seq, max_length - 2 * DL, eos_ids=(), decoding_length=DL,
                                                        branch_length=BL, max_query_length=2, idx=rank, max_steps=16)
        dt = time.time() - t1
        seq.extend(new)
        native = {'steps': len(edls_n), 'ms_per_step': round(1e3 * dt / max(len(edls_n), 1), 4),
                  'accepted_tokens_per_sec': round(sum(edls_n) / dt, 2),
                  'equals_greedy': seq[P:P + len(truth)] == truth[:len(seq) - P]}

    if rank != 0:
        if dist_on:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (gate/up GEMM, k_gemm64r<4,SWIGLU,4,8>) from live HIP events ---------
    ctx = eng.n_keys
    ids, rowmask, _, _ = cache.hier_get_packed(seq[-2:], decoding_length=DL, branch_length=BL, min_output_size=DL // 2)
    T_prof = len(ids)
    prof = eng.profile(ids, rowmask, iters=args.profile_iters)
    gu_ms = prof['ms']['gateup'] / max(prof['launches']['gateup'], 1)
    gu_bytes = 2 * shape.ffn * shape.hidden * 2 + 64 * shape.hidden * 2 + 64 * shape.ffn * 2
    ms_step = 1e3 * elapsed / K
    mean_T = float(np.mean(dls[n0:]))
    step_bytes = algorithmic_bytes(shape, 64, ctx, 64 * shape.vocab * 2)
    gemm_ms = sum(prof['ms'][k] for k in ('qkv', 'o', 'gateup', 'down', 'lm_head'))
    traffic, traffic_src = None, None
    try:        # HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 cannot run inside bench.py)
        pmc = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_latest.json')))
        key = [k for k in pmc['kernels'] if k.startswith('void k_gemm64r<4, 1, 4, 8')][0]      # gate/up launch
        traffic = pmc['kernels'][key]['hbm_bytes_per_launch']
        traffic_src = pmc['source']
    except Exception:
        pass
    roofline = {
        'bound': 'hbm', 'kernel': 'k_gemm64r<4,EPI_SWIGLU,4,8> (gate/up projection + fused SwiGLU, 32 launches/step)',
        'achieved': round(gu_bytes / (gu_ms * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
        'frac': round(gu_bytes / (gu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), 'traffic': traffic, 'traffic_source': traffic_src,
        'bytes_per_launch': gu_bytes, 'ms_per_launch': round(gu_ms, 5),
        # the same launch against the matrix-core roof: 64-row trees keep the kernel far below the MFMA ridge (HBM-bound by
        # design); profiles/r01_pmc_SQ_v5.txt holds the SQ_VALU_MFMA_BUSY_CYCLES pass of rocprofv3
        'mfma': {'flops_per_launch': 2 * 2 * shape.ffn * shape.hidden * 64,
                 'achieved_TFLOPs': round(2 * 2 * shape.ffn * shape.hidden * 64 / (gu_ms * 1e-3) / 1e12, 1),
                 'peak_TFLOPs': 2500.0, 'frac': round(2 * 2 * shape.ffn * shape.hidden * 64 / (gu_ms * 1e-3) / 1e12 / 2500.0, 4)},
        'verify_step': {'algorithmic_bytes': step_bytes, 'ms_graph_step': round(ms_step, 4),
                        'achieved_GBps': round(step_bytes / (ms_step * 1e-3) / 1e9, 1),
                        'frac': round(step_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        'ms_eager_step_events': round(prof['ms_step'], 4),
                        'ms_by_class_events': {k: round(v, 4) for k, v in prof['ms'].items()},
                        'all_gemm_GBps_events': round(2 * shape.n_params_no_embed() / (gemm_ms * 1e-3) / 1e9, 1)},
    }
    mean_acc = float(np.mean(edls[n0:]))
    cpu = None
    if not args.no_cpu_baseline and world == 1:      # the CPU leg is timed on rank 0 at N=1 only
        cpu = cpu_baseline(shape, 64, ctx, mean_acc)
    out = {
        'metric': 'accepted_tokens_per_sec', 'value': round(accepted_all / elapsed, 2), 'unit': 'tokens/s',
        'n_gpus': world, 'steps': K, 'warmup': W, 'ms_per_step': round(ms_step, 4), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
        'config': {'workload': model_name + ' bf16 bs=1/GPU lookahead verify loop, 64-token draft tree / 8-12 noisy branches '
                               '(hier, decoding_length=64, branch_length=12), synthetic permutation-LM weights (N(0,0.02); o/down std 1e-4; '
                               'lm_head[pi(t)]=embed[t]) for decisive greedy margins, 512-token phrase-bank prompt',
                   'n_layers': shape.n_layers, 'prompt_len': P, 'rho': args.rho, 'copies': args.copies, 'parallelism': f'batch-shard x{world}',
                   'mean_accept_len': round(mean_acc, 3), 'mean_draft_len': round(mean_T, 2),
                   'verify_steps_per_sec': round(K * world / elapsed, 2), 'context_at_end': ctx,
                   'trie_query_ms_mean': round(1e3 * float(np.mean(qts[n0:])), 4),
                   'lookahead_equals_greedy': bool(correct), 'plain_greedy_tokens_per_sec': round(len(truth) / t_greedy, 2),
                   'native_loop': native},
        'roofline': roofline, 'cpu_baseline': cpu,
    }
    print(json.dumps(out))
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
