# -*- coding: utf-8 -*-
"""-m gpu: the float16 instantiation of the library (liblookahead_hip_f16.so: the same sources compiled with -DLA_DTYPE=1,
v_mfma_f32_32x32x16_f16, fp16 rounding points) — the dtype the reference's own examples and benchmarks run
(lookahead/benchmarks/llama_benchmark.py:27, examples/llama_example.py:19).  Same statements as the bfloat16 tests of
test_gpu_e2e.py / test_gpu_mblock.py, against the oracle evaluated in float16 and against reference runs recorded in float16
(oracle/gen_golden_model.py, oracle/gen_golden_noisy.py).  Tolerance: the stated 2e-2 * max|logit| per row."""
import os

import numpy as np
import pytest
import torch

from oracle import llama_oracle as lo
from painlessinferenceacceleration_amd import _lib
from painlessinferenceacceleration_amd.llama_engine import LlamaShape, LlamaVerifyEngine, random_weights
from painlessinferenceacceleration_amd.lookahead_cache import LookaheadCache
from painlessinferenceacceleration_amd.modeling_llama import LlamaForCausalLM
from painlessinferenceacceleration_amd.modeling_llama_batch import LlamaForCausalLM as BatchLlama
from tests.gpu_utils import random_tree
from tests.test_gpu_e2e import TOL, _check_rows, _mask_from_rows
from tests.tiny_model import GOLDEN, tiny_decisive_weights, tiny_shape, tiny_weights

pytestmark = pytest.mark.gpu
F16 = torch.float16


def _f16_sd(seed=0, cfg=None):
    return {k: v.to(F16) for k, v in tiny_weights(seed, torch.float32, cfg=cfg).items()}


def test_the_fp16_library_is_a_second_build_of_the_same_abi():
    lib16 = _lib.lib_for(F16)
    assert lib16 is not _lib.lib and lib16.la_abi_dtype() == _lib.LA_DTYPE_F16 and _lib.lib.la_abi_dtype() == _lib.LA_DTYPE_BF16
    assert lib16.la_abi_version() == _lib.lib.la_abi_version() == _lib.ABI_VERSION
    with pytest.raises(ValueError):
        _lib.lib_for(torch.float32)


@pytest.mark.parametrize('P', [40, 150])
def test_fp16_engine_logits_match_fp16_oracle_prefill_and_tree(P):
    shape = tiny_shape()
    sd = _f16_sd()
    eng = LlamaVerifyEngine(shape, sd, max_length=512)
    assert eng.dtype == F16 and eng.logits().dtype == F16
    oracle = lo.OracleLlama(shape, sd)
    rs = np.random.RandomState(P)
    prompt = rs.randint(3, shape.vocab, size=P).tolist()
    tok = eng.prefill(prompt)
    logits_o, past = oracle.forward(torch.tensor(prompt), torch.tril(torch.ones((P, P), dtype=torch.long)), None)
    last_blk = (P - 1) // 64 * 64
    _check_rows(eng.logits()[:P - last_blk], logits_o[last_blk:], range(P - last_blk), 'fp16 prefill')
    T = 64
    _, rows = random_tree(rs, T)
    ids = np.concatenate([[tok], rs.randint(3, shape.vocab, size=T - 1)]).astype(np.int32)
    toks, ncommit = eng.step(ids, rows, mode=0)
    full = torch.cat([torch.ones((T, P), dtype=torch.long), torch.from_numpy(_mask_from_rows(rows, T))], 1)
    lg, _ = oracle.forward(torch.tensor(ids.tolist()), full, past)
    _check_rows(eng.logits(), lg, range(T), 'fp16 tree')
    st = eng.state().cpu().numpy()
    am = st[136:136 + T].tolist()
    exp_toks, exp_rows = lo.accept_scan(ids.tolist(), _mask_from_rows(rows, T), am)
    assert toks == exp_toks and ncommit == len(exp_rows)
    # fp16 resolves what bf16 cannot: the error sits well inside the tolerance (11 mantissa bits vs 8)
    err = float(((eng.logits()[:T].float().cpu() - lg.float()).abs().amax(-1) / lg.float().abs().amax(-1)).max())
    print(f'[fp16 tiny, P={P}] max rel err vs the fp16 oracle {err:.4f}')
    assert err < 1e-2


@pytest.mark.parametrize('over', [dict(hidden=256, n_heads=4, n_kv_heads=4), dict(hidden=384, n_heads=4, n_kv_heads=2), dict(hidden=256, n_heads=8, n_kv_heads=8)],
                         ids=['hd64', 'hd96-gqa', 'hd32'])
def test_fp16_narrow_heads_in_padded_lanes_vs_fp16_oracle(over):
    """head_dim < 128 in the float16 build: the softmax scale is a runtime DIVISOR there (fp16(x / sqrt(hd)); the multiply is inexact at
    hd = 32 and 128, tests/test_oracle_llama.py) — prefill rows and a tree step against the fp16 oracle, where fp16 resolves a wrong
    scale or a misplaced lane at once (error well inside 1e-2)."""
    shape = tiny_shape(**over)
    assert shape.head_dim < 128
    sd = _f16_sd(7, cfg=over)
    eng = LlamaVerifyEngine(shape, sd, max_length=512)
    assert eng.dtype == F16
    oracle = lo.OracleLlama(shape, sd)
    rs = np.random.RandomState(shape.head_dim)
    P, T = 90, 64
    prompt = rs.randint(3, shape.vocab, size=P).tolist()
    tok = eng.prefill(prompt)
    logits_o, past = oracle.forward(torch.tensor(prompt), torch.tril(torch.ones((P, P), dtype=torch.long)), None)
    _check_rows(eng.logits()[:P - 64], logits_o[64:], range(P - 64), 'fp16 narrow prefill')
    _, rows = random_tree(rs, T)
    ids = np.concatenate([[tok], rs.randint(3, shape.vocab, size=T - 1)]).astype(np.int32)
    eng.step(ids, rows, mode=0)
    full = torch.cat([torch.ones((T, P), dtype=torch.long), torch.from_numpy(_mask_from_rows(rows, T))], 1)
    lg, _ = oracle.forward(torch.tensor(ids.tolist()), full, past)
    _check_rows(eng.logits(), lg, range(T), 'fp16 narrow tree')
    err = float(((eng.logits()[:T].float().cpu() - lg.float()).abs().amax(-1) / lg.float().abs().amax(-1)).max())
    print(f'[fp16 narrow heads {over}] max rel err vs the fp16 oracle {err:.4f}')
    assert err < 1e-2


def test_fp16_llama7b_shape_two_layers_vs_oracle():
    """Real GEMM shapes (K = 4096 / 11008, N = 12288 / 4096 / 22016 / 32000) on a 2-layer model, everything in float16."""
    torch.set_num_threads(min(16, os.cpu_count() or 8))
    shape = LlamaShape(2, 4096, 32, 32, 11008, 32000, 1e-5)
    sd = random_weights(shape, seed=1, std=0.02, device='cpu', dtype=F16)
    eng = LlamaVerifyEngine(shape, sd, max_length=256)
    assert eng.dtype == F16
    oracle = lo.OracleLlama(shape, sd)
    rs = np.random.RandomState(1)
    prompt = rs.randint(3, 32000, size=64).tolist()
    tok = eng.prefill(prompt)
    lg0, past = oracle.forward(torch.tensor(prompt), torch.tril(torch.ones((64, 64), dtype=torch.long)), None)
    _check_rows(eng.logits(), lg0, range(64), 'fp16 7b-shape prefill')
    T = 64
    _, rows = random_tree(rs, T)
    ids = np.concatenate([[tok], rs.randint(3, 32000, size=T - 1)]).astype(np.int32)
    eng.step(ids, rows)
    full = torch.cat([torch.ones((T, 64), dtype=torch.long), torch.from_numpy(_mask_from_rows(rows, T))], 1)
    lg, _ = oracle.forward(torch.tensor(ids.tolist()), full, past)
    _check_rows(eng.logits(), lg, range(T), 'fp16 7b-shape tree')


def test_fp16_multi_block_prefill_at_the_7b_shape_vs_oracle():
    """The 5-8 block launches in float16 at real GEMM shapes — among them k_gemm_fatd, the paired gate/up launch with the weights streamed into
    MFMA operand registers, which only planned (balanced) images reach: a 2-layer Llama-2-7B shape, a 420-token prompt as ONE pass of 7 chained
    blocks, the rows of its last block against the fp16 oracle, then a 6-block pass (TW = 3) continuing the same sequence."""
    torch.set_num_threads(min(16, os.cpu_count() or 8))
    shape = LlamaShape(2, 4096, 32, 32, 11008, 32000, 1e-5)
    sd = random_weights(shape, seed=3, std=0.02, device='cpu', dtype=F16)
    eng = LlamaVerifyEngine(shape, dict(sd), max_length=1024, n_slots=1, max_blocks=8)
    assert eng.dtype == F16
    oracle = lo.OracleLlama(shape, sd)
    rs = np.random.RandomState(5)
    P = 420
    prompt = rs.randint(3, 32000, size=P).tolist()
    eng.mprefill(0, prompt)
    lg, past = oracle.forward(torch.tensor(prompt), torch.tril(torch.ones((P, P), dtype=torch.long)), None)
    _check_rows(eng.mlogits()[384:P], lg[384:], range(P - 384), 'fp16 7-block prefill, last block')
    more = rs.randint(3, 32000, size=6 * 64 - 20).tolist()
    Q = len(more)
    eng.mprefill(0, more)
    full = torch.cat([torch.ones((Q, P), dtype=torch.long), torch.tril(torch.ones((Q, Q), dtype=torch.long))], 1)
    lg2, _ = oracle.forward(torch.tensor(more), full, past)
    last = (Q - 1) // 64 * 64
    _check_rows(eng.mlogits()[last:Q], lg2[last:], range(Q - last), 'fp16 6-block continuation, last block')


@pytest.mark.parametrize('native', [False, True])
def test_fp16_partial_accept_run_equals_the_reference_fp16_golden(native):
    """oracle/gen_golden_noisy.py, float16: the REFERENCE loop (pretrained_model.py:947-1268) on the decisive tiny model with a noisy
    warm trie, 23 partially accepted steps — every token, dls and edls through the fp16 engine (interpreter and native loop)."""
    g = np.load(os.path.join(GOLDEN, 'llama_tiny_noisy_fp16.npz'))
    shape = tiny_shape()
    model = LlamaForCausalLM(shape, tiny_decisive_weights(0, F16), max_length=256)
    assert model.dtype == F16
    model.lookahead_cache = LookaheadCache(eos_ids=[2])
    for c in g['copies'].tolist():
        model.lookahead_cache.put(c, branch_length=13, mode='output', idx=-1)
    prompt = g['prompt'].tolist()
    max_length = len(prompt) + int(g['max_new'])
    partial = 0
    for r in range(int(g['n_runs'])):
        dk = {'use_lookahead': True, 'decoding_mode': 'hier', 'decoding_length': 64, 'branch_length': 12,
              'max_query_length': 2, 'stop_words': {}, 'native_loop': native}
        out = model.lookahead_generation(torch.tensor([prompt]), stopping_criteria=max_length, eos_token_id=2, pad_token_id=0,
                                         return_dict_in_generate=True, decoding_kwargs=dk)
        assert out.sequences[0].tolist() == g[f'r{r}_sequences'].tolist(), f'request {r}'
        assert out.kwargs['dls'] == g[f'r{r}_dls'].tolist() and out.kwargs['edls'] == g[f'r{r}_edls'].tolist(), f'request {r}'
        partial += sum(1 < e < 13 for e in out.kwargs['edls'][1:])
    assert partial >= 20


def test_fp16_multi_block_and_cursor_batch_steps_vs_oracle():
    """The batch paths in float16: 4 sequences x 64-row trees through la_llama_mstep (wide LDS-DMA GEMMs, multi-block attention) and
    3 sequences sharing one block through la_llama_bstep — every block / slot vs the fp16 oracle run on that sequence alone."""
    shape = tiny_shape()
    sd = _f16_sd(2)
    oracle = lo.OracleLlama(shape, sd)
    rs = np.random.RandomState(31)
    B = 4
    eng = LlamaVerifyEngine(shape, dict(sd), max_length=512, n_slots=B, max_blocks=B)
    blocks, refs = [], []
    for b in range(B):
        p = rs.randint(3, shape.vocab, size=int(rs.randint(20, 200))).tolist()
        tok = eng.mprefill(b, p)
        P = len(p)
        lgp, past = oracle.forward(torch.tensor(p), torch.tril(torch.ones((P, P), dtype=torch.long)), None)
        T = 64 if b == 0 else int(rs.randint(8, 65))
        _, rows = random_tree(rs, T)
        ids = np.concatenate([[tok], rs.randint(3, shape.vocab, size=T - 1)]).astype(np.int32)
        blocks.append((b, ids, rows, 0, 16))
        full = torch.cat([torch.ones((T, P), dtype=torch.long), torch.from_numpy(_mask_from_rows(rows, T))], 1)
        refs.append((oracle.forward(torch.tensor(ids.tolist()), full, past)[0], T))
    eng.mstep(blocks)
    for b, (lg, T) in enumerate(refs):
        _check_rows(eng.mlogits()[b * 64:b * 64 + T], lg, range(T), f'fp16 multi-block step, block {b}')
    del eng
    beng = LlamaVerifyEngine(shape, dict(sd), max_length=512, n_slots=3)
    prompts = {s: rs.randint(3, shape.vocab, size=n).tolist() for s, n in ((0, 90), (1, 37), (2, 130))}
    first = beng.bprefill_many(prompts)
    segs, want = [], {}
    for s, n in ((0, 20), (1, 5), (2, 30)):
        _, rows = random_tree(rs, n)
        ids = np.concatenate([[first[s]], rs.randint(3, shape.vocab, size=n - 1)]).astype(np.int32)
        segs.append((s, ids, np.asarray(rows, dtype=np.uint64), 0, 16))
        P = len(prompts[s])
        _, past = oracle.forward(torch.tensor(prompts[s]), torch.tril(torch.ones((P, P), dtype=torch.long)), None)
        full = torch.cat([torch.ones((n, P), dtype=torch.long), torch.from_numpy(_mask_from_rows(rows, n))], 1)
        want[s] = (oracle.forward(torch.tensor(ids.tolist()), full, past)[0], n)
    beng.bstep(segs)
    base = beng.bstep_rows()
    for s, (lg, n) in want.items():
        _check_rows(beng.logits()[base[s]:base[s] + n], lg, range(n), f'fp16 cursor batch, slot {s}')


def test_fp16_generate_front_door_matches_plain_decoding():
    """from the model wrapper: generate() with lookahead == plain decoding through the same fp16 engine; a bf16 model and an fp16
    model live side by side in one process (two libraries, one ABI)."""
    shape = tiny_shape()
    m16 = LlamaForCausalLM(shape, tiny_decisive_weights(0, F16), max_length=256)
    mbf = LlamaForCausalLM(shape, tiny_decisive_weights(0, torch.bfloat16), max_length=256)
    assert m16.engine._lib is not mbf.engine._lib
    rs = np.random.RandomState(7)
    prompt = torch.tensor([rs.randint(3, shape.vocab, size=50).tolist()])
    dk = {'use_lookahead': True, 'decoding_length': 64, 'branch_length': 12, 'stop_words': {}}
    outs = []
    for m in (m16, mbf):
        plain = m.generate(input_ids=prompt, max_new_tokens=60, eos_token_id=[None], decoding_kwargs={'use_lookahead': False})[0].tolist()
        m.lookahead_cache = LookaheadCache(eos_ids=[None])
        m.lookahead_cache.put(plain[48:], branch_length=13, mode='output', idx=-1)
        la = m.generate(input_ids=prompt, max_new_tokens=60, eos_token_id=[None], decoding_kwargs=dict(dk), return_dict_in_generate=True)
        assert la.sequences[0].tolist()[:len(plain)] == plain[:len(la.sequences[0])]
        assert np.mean(la.kwargs['edls'][1:]) > 3
        outs.append(plain)
    assert outs[0] == outs[1]              # the decisive tiny model decodes identically in both 16-bit types
    bm = BatchLlama(shape, tiny_decisive_weights(0, F16), max_length=256, max_batch=2)
    two = torch.tensor([rs.randint(3, shape.vocab, size=40).tolist() for _ in range(2)])
    ref2 = [m16.generate(input_ids=two[i:i + 1], max_new_tokens=24, eos_token_id=[None], decoding_kwargs={'use_lookahead': False})[0].tolist() for i in range(2)]
    out2 = bm.generate(input_ids=two, attention_mask=torch.ones_like(two), max_new_tokens=24, eos_token_id=[None], pad_token_id=0,
                       decoding_kwargs=dict(dk))
    for i in range(2):
        assert out2[i].tolist()[:len(ref2[i])] == ref2[i][:out2.shape[1]]
