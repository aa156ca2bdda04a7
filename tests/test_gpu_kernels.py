# -*- coding: utf-8 -*-
"""-m gpu: every hand-written kernel through its C-ABI entry point against plain PyTorch on the same inputs
(floating point: stated tolerance; integer/index work: bit-exact against the oracle / reference vectors)."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest
import torch

from painlessinferenceacceleration_amd import _lib
from painlessinferenceacceleration_amd._lib import lib, check
from tests import gpu_utils as gu
from tests.gpu_utils import DEV, ptr, sp

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def bf(t):
    return t.to(torch.bfloat16)


def test_library_is_the_native_one():
    assert lib.la_abi_version() == _lib.ABI_VERSION
    assert torch.cuda.is_available()
    assert 'gfx950' in torch.cuda.get_device_properties(0).gcnArchName


def test_pack_layouts_roundtrip():
    g = torch.Generator(device=DEV).manual_seed(0)
    x = bf(torch.randn(64, 512, generator=g, device=DEV))
    xp = gu.pack_x(x)
    assert torch.equal(gu.from_packed(xp, gu.xp_index(512)), x)
    w = bf(torch.randn(96, 64, generator=g, device=DEV))
    wp = gu.pack_weight(w)
    # WP tile (nb, kb): lane = n%32 + 32*((k%16)/8), e = k%8
    n = np.arange(96)[:, None]; k = np.arange(64)[None, :]
    idx = (((n >> 5) * (64 // 16) + (k >> 4)) * 512 + ((n & 31) + 32 * ((k >> 3) & 1)) * 8 + (k & 7)).astype(np.int64)
    assert torch.equal(gu.from_packed(wp, idx), w)


@pytest.mark.parametrize('variant', [0, 1, 2])
@pytest.mark.parametrize('N,K,rb,ks', [(256, 4096, 1, 1), (256, 4096, 2, 1), (4096, 4096, 1, 2), (4096, 4096, 2, 4),
                                       (512, 11008, 1, 2), (512, 11008, 2, 1), (12288, 4096, 1, 1), (64, 48, 2, 1),
                                       (96, 1104, 1, 4), (4096, 11008, 1, 3)])
def test_gemm64_slab(N, K, rb, ks, variant):
    """out[64][N] = x . W^T, bf16 in / fp32 accumulate; tolerance: 2e-3 of max|out| vs an fp64 product."""
    rb = rb | (variant << 8)
    g = torch.Generator(device=DEV).manual_seed(N + K)
    x = bf(torch.randn(64, K, generator=g, device=DEV))
    w = bf(torch.randn(N, K, generator=g, device=DEV) * 0.05)
    slabs = torch.full((ks, 64, N), float('nan'), dtype=torch.float32, device=DEV)
    wp, xp = gu.pack_weight(w), gu.pack_x(x)          # keep references: temporaries would be recycled by the allocator
    check(lib.la_gemm64_slab(sp(), ptr(wp), ptr(xp), N, K, rb, ks, ptr(slabs)), 'gemm')
    torch.cuda.synchronize()
    got = slabs.sum(0)
    ref = x.double() @ w.double().t()
    assert not torch.isnan(got).any()
    assert gu.rel_err(got, ref) < 2e-3, gu.rel_err(got, ref)
    # asymmetric check (transpose detector): a one-hot x row must reproduce a W column
    x2 = torch.zeros(64, K, dtype=torch.bfloat16, device=DEV)
    x2[5, 7] = 1.0; x2[40, K - 3] = 2.0
    xp2 = gu.pack_x(x2)
    check(lib.la_gemm64_slab(sp(), ptr(wp), ptr(xp2), N, K, rb, ks, ptr(slabs)), 'gemm')
    got = slabs.sum(0)
    assert torch.allclose(got[5], w[:, 7].float(), atol=1e-6) and torch.allclose(got[40], 2 * w[:, K - 3].float(), atol=1e-6)
    assert float(got[6].abs().max()) == 0.0


@pytest.mark.parametrize('variant', [0, 2])
def test_gemm64_swiglu(variant):
    """act = bf16(silu(bf16(g)) * bf16(u)) (LlamaMLP), written in the packed operand order of down_proj."""
    F, K = 1024, 512
    g = torch.Generator(device=DEV).manual_seed(3)
    x = bf(torch.randn(64, K, generator=g, device=DEV))
    wg = bf(torch.randn(F, K, generator=g, device=DEV) * 0.05)
    wu = bf(torch.randn(F, K, generator=g, device=DEV) * 0.05)
    act = torch.zeros(64 * F, dtype=torch.bfloat16, device=DEV)
    wp, xp = gu.pack_weight(wg, wu), gu.pack_x(x)
    check(lib.la_gemm64_swiglu(sp(), ptr(wp), ptr(xp), F, K, ptr(act), variant), 'swiglu')
    torch.cuda.synchronize()
    got = gu.from_packed(act, gu.xp_index(F)).float()
    gg, uu = bf(x.float() @ wg.float().t()), bf(x.float() @ wu.float().t())
    ref = bf(bf(torch.nn.functional.silu(gg.float())).float() * uu.float()).float()
    assert gu.rel_err(got, ref) < 2e-2, gu.rel_err(got, ref)     # one bf16 ulp of the largest value


@pytest.mark.parametrize('rb', [1, 2])
def test_gemm64_logits_argmax(rb):
    V, K = 32000, 512
    g = torch.Generator(device=DEV).manual_seed(5)
    x = bf(torch.randn(64, K, generator=g, device=DEV))
    w = bf(torch.randn(V, K, generator=g, device=DEV) * 0.05)
    logits = torch.zeros(64, V, dtype=torch.bfloat16, device=DEV)
    nt = V // (32 * rb) * 2          # candidate slots: (workgroup, register-group owner), 4-wave workgroups
    cv = torch.zeros(nt * 64, dtype=torch.float32, device=DEV)
    ci = torch.zeros(nt * 64, dtype=torch.int32, device=DEV)
    state = torch.zeros(_lib.LA_ST_WORDS, dtype=torch.int32, device=DEV)
    wp, xp = gu.pack_weight(w), gu.pack_x(x)
    check(lib.la_gemm64_logits(sp(), ptr(wp), ptr(xp), V, K, rb, ptr(logits), ptr(cv), ptr(ci)), 'logits')
    check(lib.la_argmax_finalize(sp(), ptr(cv), ptr(ci), nt, ptr(state)), 'argmax')
    torch.cuda.synchronize()
    ref = x.double() @ w.double().t()
    assert gu.rel_err(logits.float(), ref) < 1e-2
    # argmax is exact with respect to the bf16 logits the kernel itself produced (first maximum wins)
    am = state[_lib.LA_ST_ARGMAX:_lib.LA_ST_ARGMAX + 64].cpu()
    lf = logits.float().cpu()
    exp = torch.tensor([int((row == row.max()).nonzero()[0]) for row in lf], dtype=torch.int32)
    assert torch.equal(am, exp)


@pytest.mark.parametrize('hidden', [256, 4096, 5120])
def test_row_norm_kernels(hidden):
    g = torch.Generator(device=DEV).manual_seed(hidden)
    V = 1000
    embed = bf(torch.randn(V, hidden, generator=g, device=DEV))
    ids = torch.randint(0, V, (64,), generator=g, device=DEV, dtype=torch.int32)
    nw = bf(1 + 0.1 * torch.randn(hidden, generator=g, device=DEV))
    h = torch.zeros(64, hidden, dtype=torch.bfloat16, device=DEV)
    xp = torch.zeros(64 * hidden, dtype=torch.bfloat16, device=DEV)
    eps = 1e-5
    check(lib.la_embed_norm(sp(), ptr(embed), ptr(ids), ptr(nw), hidden, eps, ptr(h), ptr(xp)), 'embed_norm')
    torch.cuda.synchronize()
    h_ref = embed[ids.long()]
    assert torch.equal(h, h_ref)

    def rms(hh):
        v = hh.float().pow(2).mean(-1, keepdim=True)
        return bf(nw.float() * (hh.float() * torch.rsqrt(v + eps)))
    got = gu.from_packed(xp, gu.xp_index(hidden))
    assert gu.rel_err(got.float(), rms(h_ref).float()) < 1e-2
    # residual add of two fp32 slabs, then norm
    slabs = torch.randn(2, 64, hidden, generator=g, device=DEV)
    check(lib.la_resid_norm(sp(), ptr(h), ptr(slabs), 2, ptr(nw), hidden, eps, ptr(xp)), 'resid_norm')
    torch.cuda.synchronize()
    h2 = bf(h_ref.float() + bf(slabs.sum(0)).float())
    assert gu.rel_err(h.float(), h2.float()) < 1e-2 and (h != h2).float().mean() < 0.01
    got = gu.from_packed(xp, gu.xp_index(hidden))
    assert gu.rel_err(got.float(), rms(h).float()) < 1e-2


def _rope_ref(x, pos, theta=10000.0):
    """apply_rotary_pos_emb in bf16 arithmetic (modeling_llama.py:154-169) on x [T, H, 128]."""
    inv = 1.0 / (theta ** (torch.arange(0, 128, 2, dtype=torch.int64).float() / 128))
    ang = (inv[:, None].float() @ pos[None, :].float().cpu()).transpose(0, 1)
    emb = torch.cat((ang, ang), -1)
    cos, sin = bf(emb.cos()).to(x.device)[:, None], bf(emb.sin()).to(x.device)[:, None]
    rot = torch.cat((-x[..., 64:], x[..., :64]), -1)
    return (x * cos) + (rot * sin)


@pytest.mark.parametrize('nh,nkv', [(2, 2), (8, 2)])
def test_qkv_post(nh, nkv):
    from painlessinferenceacceleration_amd.llama_engine import rope_tables
    g = torch.Generator(device=DEV).manual_seed(11)
    N = (nh + 2 * nkv) * 128
    slabs = torch.randn(2, 64, N, generator=g, device=DEV)
    pos = torch.randint(0, 900, (64,), generator=g, device=DEV, dtype=torch.int32)
    rc, rs_ = rope_tables(128, 1024, 10000.0, DEV)
    qf = torch.zeros(nh * 8192, dtype=torch.bfloat16, device=DEV)
    kf = torch.zeros(nkv * 8192, dtype=torch.bfloat16, device=DEV)
    vf = torch.zeros(nkv * 8192, dtype=torch.bfloat16, device=DEV)
    check(lib.la_qkv_post(sp(), ptr(slabs), 2, nh, nkv, ptr(pos), ptr(rc), ptr(rs_), ptr(qf), ptr(kf), ptr(vf)), 'qkv_post')
    torch.cuda.synchronize()
    qkv = bf(slabs.sum(0))
    q = qkv[:, :nh * 128].view(64, nh, 128)
    k = qkv[:, nh * 128:(nh + nkv) * 128].view(64, nkv, 128)
    v = qkv[:, (nh + nkv) * 128:].view(64, nkv, 128)
    q_ref, k_ref = _rope_ref(q, pos.long()), _rope_ref(k, pos.long())
    got_q = gu.from_packed(qf.view(nh, 8192), gu.rf_index(64)).transpose(0, 1)     # [64, nh, 128]
    got_k = gu.from_packed(kf.view(nkv, 8192), gu.rf_index(64)).transpose(0, 1)
    got_v = gu.from_packed(vf.view(nkv, 8192), gu.vf_index(64)).transpose(0, 1)
    assert torch.equal(got_v, v)
    assert torch.equal(got_q, q_ref), (got_q.float() - q_ref.float()).abs().max()
    assert torch.equal(got_k, k_ref)


@pytest.mark.parametrize('nh,nkv,nkeys,nsplit,T', [(2, 2, 0, 1, 64), (2, 2, 70, 2, 64), (4, 1, 333, 4, 17), (2, 2, 640, 8, 64),
                                                   (2, 2, 31, 3, 1), (2, 2, 1500, 8, 64), (2, 2, 1500, 2, 64), (4, 2, 2040, 1, 40),
                                                   (2, 2, 5, 8, 3)])
@pytest.mark.usefixtures('lab_build')
def test_tree_attention(nh, nkv, nkeys, nsplit, T):
    """softmax(QK^T/sqrt(d) + tree mask) V with a mask-free prefix: tolerance 2e-2 relative to max|out|
    (bf16 P and bf16 output rounding) against an fp32 torch attention over the same bf16 inputs.  Checked form: the single-launch
    kernel (la_attn1.hip, the default of the single-sequence step; la_debug_set key 17).  The two key-split forms behind it —
    K/V tiles straight into registers, and staged once per workgroup through LDS (key 10) — must agree BITWISE with each other
    (partials and packed output): same arithmetic in the same order, only the way the tiles reach the MFMA operands differs."""
    rs = np.random.RandomState(nkeys + T)
    g = torch.Generator(device=DEV).manual_seed(nkeys)
    max_keys = 2048
    q = bf(torch.randn(nh, 64, 128, generator=g, device=DEV))
    kmain = bf(torch.randn(nkv, max_keys, 128, generator=g, device=DEV))     # rows >= nkeys are stale garbage
    vmain = bf(torch.randn(nkv, max_keys, 128, generator=g, device=DEV))
    kfr = bf(torch.randn(nkv, 64, 128, generator=g, device=DEV))
    vfr = bf(torch.randn(nkv, 64, 128, generator=g, device=DEV))
    _, rows = gu.random_tree(rs, T)
    rowmask = np.array([int(rows[t]) if t < T else (1 << t) for t in range(64)], dtype=np.uint64)
    qf = gu.to_packed(q, gu.rf_index(64), 8192).reshape(-1)
    km = gu.to_packed(kmain, gu.rf_index(max_keys), max_keys * 128).reshape(-1)
    vm = gu.to_packed(vmain, gu.vf_index(max_keys), max_keys * 128).reshape(-1)
    kf = gu.to_packed(kfr, gu.rf_index(64), 8192).reshape(-1)
    vf = gu.to_packed(vfr, gu.vf_index(64), 8192).reshape(-1)
    state = torch.zeros(_lib.LA_ST_WORDS, dtype=torch.int32, device=DEV)
    state[_lib.LA_ST_NKEYS] = nkeys
    rm = torch.from_numpy(rowmask.view(np.int64)).to(DEV)
    opart = torch.zeros(nh * nsplit * 64 * 128, dtype=torch.float32, device=DEV)
    mpart = torch.zeros(nh * nsplit * 64, dtype=torch.float32, device=DEV)
    lpart = torch.zeros_like(mpart)
    out = torch.zeros(64 * nh * 128, dtype=torch.bfloat16, device=DEV)
    default_form, default_one = lib.la_lab_get(10), lib.la_lab_get(17)
    forms = []
    try:
        # (single launch, -), (key splits + combine, direct), (key splits + combine, staged through LDS)
        for one, staged in ((1, 0), (0, 0), (0, 1)):
            check(lib.la_lab_set(17, one), 'debug_set')
            check(lib.la_lab_set(10, staged), 'debug_set')
            for t in (opart, mpart, lpart, out):
                t.zero_()
            check(lib.la_tree_attn(sp(), ptr(qf), ptr(km), ptr(vm), ptr(kf), ptr(vf), ptr(rm), ptr(state), nh, nkv, max_keys,
                                   nsplit, ptr(opart), ptr(mpart), ptr(lpart), ptr(out)), 'tree_attn')
            torch.cuda.synchronize()
            forms.append((opart.clone(), mpart.clone(), lpart.clone(), out.clone()))
    finally:
        lib.la_lab_set(10, default_form)
        lib.la_lab_set(17, default_one)
    for a_, b_ in zip(forms[1], forms[2]):
        assert torch.equal(a_, b_)
    assert float(forms[0][0].abs().max()) == 0.0, 'the single-launch form writes no split partials'
    # single launch vs key splits: the same per-(row, key) arithmetic summed in another tile order -> a few bf16 ulps apart
    one_, split_ = (gu.from_packed(f[3], gu.xp_index(nh * 128)).float().view(64, nh, 128)[:T] for f in (forms[0], forms[1]))
    assert gu.rel_err(one_, split_) < 1e-2, gu.rel_err(one_, split_)
    out = forms[0][3]
    got = gu.from_packed(out, gu.xp_index(nh * 128)).float().view(64, nh, 128)
    rep = nh // nkv
    K = torch.cat([kmain[:, :nkeys], kfr], 1).float().repeat_interleave(rep, 0)       # [nh, nkeys+64, 128]
    Vv = torch.cat([vmain[:, :nkeys], vfr], 1).float().repeat_interleave(rep, 0)
    mask = torch.ones(64, nkeys + 64, dtype=torch.bool, device=DEV)
    mask[:, nkeys:] = torch.tensor([[(int(rowmask[t]) >> j) & 1 for j in range(64)] for t in range(64)], dtype=torch.bool)
    s = torch.einsum('htd,hkd->htk', q.float(), K) / math.sqrt(128)
    s = s.masked_fill(~mask[None], float('-inf'))
    ref = torch.einsum('htk,hkd->thd', torch.softmax(s, -1), Vv)
    err = gu.rel_err(got[:T], ref[:T])
    assert err < 2e-2, err
    assert not torch.isnan(got).any()


def test_accept_scan_matches_reference_vectors():
    """Bit-exact: the reference's own outputs (tests/golden/accept_scan.json, recorded from
    _lookahead_update_model_kwargs_for_generation) for tokens, count and kept KV rows."""
    vecs = json.load(open(os.path.join(GOLDEN, 'accept_scan.json')))
    for v in vecs:
        T = len(v['ids'])
        ids = torch.zeros(64, dtype=torch.int32); ids[:T] = torch.tensor(v['ids'], dtype=torch.int32)
        rm = np.array([v['rows'][t] if t < T else (1 << t) for t in range(64)], dtype=np.uint64)
        state = torch.zeros(_lib.LA_ST_WORDS, dtype=torch.int32)
        ctx = v['context_length']
        state[_lib.LA_ST_NKEYS] = ctx - 1
        state[_lib.LA_ST_T] = T
        state[_lib.LA_ST_ARGMAX:_lib.LA_ST_ARGMAX + T] = torch.tensor(v['argmax'], dtype=torch.int32)
        d_ids, d_state = ids.to(DEV), state.to(DEV)
        d_rm = torch.from_numpy(rm.view(np.int64)).to(DEV)
        check(lib.la_accept_scan(sp(), ptr(d_ids), ptr(d_rm), ptr(d_state)), 'accept')
        torch.cuda.synchronize()
        st = d_state.cpu().numpy()
        n = int(st[_lib.LA_ST_NOUT])
        assert st[_lib.LA_ST_OUTTOK:_lib.LA_ST_OUTTOK + n].tolist() == v['next_token_list'], v
        assert n == v['edls'][0] and int(st[_lib.LA_ST_NCOMMIT]) == n
        src = st[_lib.LA_ST_SRCIDX:_lib.LA_ST_SRCIDX + n].tolist()
        kept = list(range(ctx - 1)) + [ctx - 1 + s for s in src]           # committed keys after the step
        assert kept == v['kept_kv'], (kept, v['kept_kv'])
        assert int(st[_lib.LA_ST_NKEYS]) == ctx - 1 + n and int(st[_lib.LA_ST_DSTBASE]) == ctx - 1


def test_kv_commit_moves_rows():
    L, nkv, max_keys = 3, 2, 256
    g = torch.Generator(device=DEV).manual_seed(2)
    kfr = bf(torch.randn(L * nkv, 64, 128, generator=g, device=DEV))
    vfr = bf(torch.randn(L * nkv, 64, 128, generator=g, device=DEV))
    kf = gu.to_packed(kfr, gu.rf_index(64), 8192).reshape(-1)
    vf = gu.to_packed(vfr, gu.vf_index(64), 8192).reshape(-1)
    km = torch.zeros(L * nkv * max_keys * 128, dtype=torch.bfloat16, device=DEV)
    vm = torch.zeros_like(km)
    src = [0, 5, 6, 40, 63]
    state = torch.zeros(_lib.LA_ST_WORDS, dtype=torch.int32)
    state[_lib.LA_ST_NCOMMIT] = len(src); state[_lib.LA_ST_DSTBASE] = 30
    state[_lib.LA_ST_SRCIDX:_lib.LA_ST_SRCIDX + len(src)] = torch.tensor(src, dtype=torch.int32)
    d_state = state.to(DEV)
    check(lib.la_kv_commit(sp(), ptr(kf), ptr(vf), ptr(km), ptr(vm), ptr(d_state), L, nkv, max_keys), 'commit')
    torch.cuda.synchronize()
    K = gu.from_packed(km.view(L * nkv, -1), gu.rf_index(max_keys))
    V = gu.from_packed(vm.view(L * nkv, -1), gu.vf_index(max_keys))
    for i, s in enumerate(src):
        assert torch.equal(K[:, 30 + i], kfr[:, s]) and torch.equal(V[:, 30 + i], vfr[:, s])
    assert float(K[:, :30].abs().max()) == 0 and float(K[:, 35:].abs().max()) == 0 and float(V[:, 35:].abs().max()) == 0


@pytest.mark.parametrize('nh,nkv', [(2, 2), (8, 2)])
def test_gemm64_qkv_fused_equals_slab_plus_qkv_post(nh, nkv):
    """The fused QKV epilogue (RoPE + fragment writes inside the GEMM) must reproduce la_gemm64_slab + la_qkv_post
    bit for bit: same fp32 sums (ksplit 1, same k order), same rounding points."""
    from painlessinferenceacceleration_amd.llama_engine import rope_tables
    K = 512
    N = (nh + 2 * nkv) * 128
    g = torch.Generator(device=DEV).manual_seed(21)
    x = bf(torch.randn(64, K, generator=g, device=DEV))
    w = bf(torch.randn(N, K, generator=g, device=DEV) * 0.05)
    pos = torch.randint(0, 900, (64,), generator=g, device=DEV, dtype=torch.int32)
    rc, rs_ = rope_tables(128, 1024, 10000.0, DEV)
    xp = gu.pack_x(x)
    outs = []
    for fused in (False, True):
        qf = torch.zeros(nh * 8192, dtype=torch.bfloat16, device=DEV)
        kf = torch.zeros(nkv * 8192, dtype=torch.bfloat16, device=DEV)
        vf = torch.zeros(nkv * 8192, dtype=torch.bfloat16, device=DEV)
        if fused:
            perm = np.zeros(N, dtype=np.int32)
            check(lib.la_qkv_row_perm(nh, nkv, perm.ctypes.data_as(_lib.pi32)), 'perm')
            assert sorted(perm.tolist()) == list(range(N))
            wp = gu.pack_weight(w[torch.from_numpy(perm.astype(np.int64)).to(DEV)].contiguous())
            check(lib.la_gemm64_qkv(sp(), ptr(wp), ptr(xp), nh, nkv, K, ptr(pos), ptr(rc), ptr(rs_), ptr(qf), ptr(kf), ptr(vf), 0), 'qkv')
        else:
            wp = gu.pack_weight(w)
            slabs = torch.zeros(1, 64, N, dtype=torch.float32, device=DEV)
            check(lib.la_gemm64_slab(sp(), ptr(wp), ptr(xp), N, K, 2, 1, ptr(slabs)), 'gemm')
            check(lib.la_qkv_post(sp(), ptr(slabs), 1, nh, nkv, ptr(pos), ptr(rc), ptr(rs_), ptr(qf), ptr(kf), ptr(vf)), 'post')
        torch.cuda.synchronize()
        outs.append((qf, kf, vf))
    for a, b, name in zip(outs[0], outs[1], 'qkv'):
        assert torch.equal(a, b), (name, int((a != b).sum()))


@pytest.mark.parametrize('F,K,nwg', [(688, 512, 16), (11008, 4096, 256), (13824, 256, 256)])
def test_gemm64r_swiglu_balanced(F, K, nwg):
    """One workgroup per CU with partial row-blocks (R = F/nwg): same result as the reference formula."""
    g = torch.Generator(device=DEV).manual_seed(F)
    x = bf(torch.randn(64, K, generator=g, device=DEV))
    wg = bf(torch.randn(F, K, generator=g, device=DEV) * 0.05)
    wu = bf(torch.randn(F, K, generator=g, device=DEV) * 0.05)
    act = torch.zeros(64 * F, dtype=torch.bfloat16, device=DEV)
    wp, xp = gu.pack_planned(1, [wg, wu], nwg), gu.pack_x(x)
    check(lib.la_gemm64r_swiglu(sp(), ptr(wp), ptr(xp), F, K, nwg, ptr(act)), 'swiglu_r')
    torch.cuda.synchronize()
    got = gu.from_packed(act, gu.xp_index(F)).float()
    gg, uu = bf(x.float() @ wg.float().t()), bf(x.float() @ wu.float().t())
    ref = bf(bf(torch.nn.functional.silu(gg.float())).float() * uu.float()).float()
    assert gu.rel_err(got, ref) < 2e-2, gu.rel_err(got, ref)
    assert float((got != ref).float().mean()) < 0.02


@pytest.mark.parametrize('V,K,nwg', [(1000, 512, 8), (32000, 512, 256)])
def test_gemm64r_logits_balanced(V, K, nwg):
    g = torch.Generator(device=DEV).manual_seed(V)
    x = bf(torch.randn(64, K, generator=g, device=DEV))
    w = bf(torch.randn(V, K, generator=g, device=DEV) * 0.05)
    logits = torch.zeros(64, V, dtype=torch.bfloat16, device=DEV)
    cv = torch.zeros(nwg * 8 * 64, dtype=torch.float32, device=DEV)
    ci = torch.zeros(nwg * 8 * 64, dtype=torch.int32, device=DEV)
    state = torch.zeros(_lib.LA_ST_WORDS, dtype=torch.int32, device=DEV)
    wp, xp = gu.pack_planned(0, [w], nwg), gu.pack_x(x)
    check(lib.la_gemm64r_logits(sp(), ptr(wp), ptr(xp), V, K, nwg, ptr(logits), ptr(cv), ptr(ci)), 'logits_r')
    check(lib.la_argmax_finalize(sp(), ptr(cv), ptr(ci), nwg, ptr(state)), 'argmax')      # one candidate per (workgroup, token)
    torch.cuda.synchronize()
    assert gu.rel_err(logits.float(), x.double() @ w.double().t()) < 1e-2
    lf = logits.float().cpu()
    exp = torch.tensor([int((row == row.max()).nonzero()[0]) for row in lf], dtype=torch.int32)
    assert torch.equal(state[_lib.LA_ST_ARGMAX:_lib.LA_ST_ARGMAX + 64].cpu(), exp)


@pytest.mark.parametrize('nh,nkv,nwg', [(3, 1, 16), (32, 32, 256), (8, 2, 32)])
def test_gemm64r_qkv_balanced_equals_unfused(nh, nkv, nwg):
    """Balanced fused QKV (R RoPE pairs per workgroup, spanning head boundaries) vs la_gemm64_slab + la_qkv_post:
    same rounding points; the fp32 sums differ by the K split (8 waves vs 4), so results agree to one bf16 ulp on a
    small fraction of elements."""
    from painlessinferenceacceleration_amd.llama_engine import rope_tables
    K = 512
    N = (nh + 2 * nkv) * 128
    g = torch.Generator(device=DEV).manual_seed(nh * 7 + nkv)
    x = bf(torch.randn(64, K, generator=g, device=DEV))
    w = bf(torch.randn(N, K, generator=g, device=DEV) * 0.05)
    pos = torch.randint(0, 900, (64,), generator=g, device=DEV, dtype=torch.int32)
    rc, rs_ = rope_tables(128, 1024, 10000.0, DEV)
    xp = gu.pack_x(x)
    outs = []
    for bal in (False, True):
        qf = torch.zeros(nh * 8192, dtype=torch.bfloat16, device=DEV)
        kf = torch.zeros(nkv * 8192, dtype=torch.bfloat16, device=DEV)
        vf = torch.zeros(nkv * 8192, dtype=torch.bfloat16, device=DEV)
        if bal:
            wp = gu.pack_planned(2, [w], nwg)
            check(lib.la_gemm64r_qkv(sp(), ptr(wp), ptr(xp), nh, nkv, K, nwg, ptr(pos), ptr(rc), ptr(rs_), ptr(qf), ptr(kf), ptr(vf)), 'qkv_r')
        else:
            wp = gu.pack_weight(w)
            slabs = torch.zeros(1, 64, N, dtype=torch.float32, device=DEV)
            check(lib.la_gemm64_slab(sp(), ptr(wp), ptr(xp), N, K, 2, 1, ptr(slabs)), 'gemm')
            check(lib.la_qkv_post(sp(), ptr(slabs), 1, nh, nkv, ptr(pos), ptr(rc), ptr(rs_), ptr(qf), ptr(kf), ptr(vf)), 'post')
        torch.cuda.synchronize()
        outs.append((qf, kf, vf))
    for a, b, name in zip(outs[0], outs[1], 'qkv'):
        assert gu.rel_err(a.float(), b.float()) < 1e-2, name
        assert float((a != b).float().mean()) < 0.02, name
