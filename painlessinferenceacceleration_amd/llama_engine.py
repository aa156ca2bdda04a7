# -*- coding: utf-8 -*-
"""LlamaVerifyEngine — host-side owner of the device memory behind la_llama_* (the verify forward).

Plumbing only: torch allocates HBM (weights repacked into MFMA-fragment order, KV caches, scratch),
provides the stream and pinned staging buffers; every kernel of the step is hand-written HIP inside
liblookahead_hip.so (csrc/la_kernels.hip, csrc/la_engine.cpp).  There is no torch fallback: without a
GPU the constructor raises.

Reference counterpart: LlamaForCausalLM.forward under the rank-4 mask hook
(lookahead/lookahead/models/llama/modeling_llama.py:544-677, 710-794), called once per verify step
from lookahead_generation (common/pretrained_model.py:1176-1181).
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import _lib
from ._lib import lib, check


class LlamaShape(object):
    """The fields of LlamaConfig the path uses."""

    def __init__(self, n_layers=32, hidden=4096, n_heads=32, n_kv_heads=None, ffn=11008, vocab=32000,
                 rms_eps=1e-5, rope_theta=10000.0, head_dim=None, n_experts=0, top_k=2, norm_cast_first=False, sliding_window=0):
        self.n_layers, self.hidden, self.n_heads = n_layers, hidden, n_heads
        self.n_experts, self.top_k = n_experts, top_k       # > 0: Mixtral sparse-MoE MLP
        # RMSNorm flavour: False = LlamaRMSNorm (one rounding), True = Mistral/MixtralRMSNorm (normalised value rounded
        # to the activation dtype before the weight multiply, mixtral/modeling_mixtral.py:160-165)
        self.norm_cast_first = bool(norm_cast_first)
        # 0 = full attention (what the reference's lookahead path does for every family); > 0 = sliding window (extension)
        self.sliding_window = int(sliding_window)
        self.n_kv_heads = n_kv_heads if n_kv_heads is not None else n_heads
        self.ffn, self.vocab, self.rms_eps, self.rope_theta = ffn, vocab, rms_eps, rope_theta
        self.head_dim = head_dim if head_dim is not None else hidden // n_heads

    @classmethod
    def llama2_7b(cls):
        return cls(32, 4096, 32, 32, 11008, 32000, 1e-5)

    @classmethod
    def llama2_13b(cls):
        return cls(40, 5120, 40, 40, 13824, 32000, 1e-5)

    @classmethod
    def mistral_7b(cls):
        return cls(32, 4096, 32, 8, 14336, 32000, 1e-5, rope_theta=10000.0, norm_cast_first=True)

    @classmethod
    def mixtral_8x7b(cls):
        return cls(32, 4096, 32, 8, 14336, 32000, 1e-5, rope_theta=1e6, n_experts=8, top_k=2, norm_cast_first=True)

    @classmethod
    def from_hf(cls, cfg):
        """LlamaConfig / MistralConfig / MixtralConfig of transformers 4.3x (the reference's pins) or 5.x (rope_theta moved
        into cfg.rope_parameters)."""
        rope_theta = getattr(cfg, 'rope_theta', None)
        if rope_theta is None:
            rp = getattr(cfg, 'rope_parameters', None) or {}
            rope_theta = rp.get('rope_theta', 10000.0) if isinstance(rp, dict) else getattr(rp, 'rope_theta', 10000.0)
        scaling = getattr(cfg, 'rope_scaling', None)
        if scaling and (scaling.get('rope_type', scaling.get('type', 'default')) not in ('default', None)):
            raise NotImplementedError(f'rope_scaling {scaling} is not supported by the RoPE tables of this path')
        head_dim = getattr(cfg, 'head_dim', None) or cfg.hidden_size // cfg.num_attention_heads
        return cls(cfg.num_hidden_layers, cfg.hidden_size, cfg.num_attention_heads,
                   getattr(cfg, 'num_key_value_heads', None), cfg.intermediate_size, cfg.vocab_size,
                   cfg.rms_norm_eps, float(rope_theta), head_dim=head_dim,
                   n_experts=getattr(cfg, 'num_local_experts', 0) or 0, top_k=getattr(cfg, 'num_experts_per_tok', 2),
                   norm_cast_first=getattr(cfg, 'model_type', 'llama') in ('mistral', 'mixtral'))

    def n_params_no_embed(self):
        hd = self.head_dim
        mlp = 3 * self.ffn * self.hidden * max(self.n_experts, 1) + self.n_experts * self.hidden
        per_layer = (self.n_heads + 2 * self.n_kv_heads) * hd * self.hidden + self.n_heads * hd * self.hidden \
            + mlp + 2 * self.hidden
        return self.n_layers * per_layer + self.hidden + self.vocab * self.hidden


def legacy_state_dict(sd, shape):
    """HF-named state dict -> the transformers-4.36 parameter names the engine consumes (the reference's model files:
    block_sparse_moe.gate / experts.{e}.w1|w3|w2, mixtral/modeling_mixtral.py:668-759).  transformers 5.x stores the experts of
    a layer fused: mlp.gate.weight [E, hidden], mlp.experts.gate_up_proj [E, 2*ffn, hidden] (gate rows first),
    mlp.experts.down_proj [E, hidden, ffn]; tied lm_head falls back to the embedding."""
    out = {}
    for k, v in sd.items():
        v = v.detach()
        if shape.n_experts > 0 and k.endswith('.mlp.gate.weight'):
            out[k.replace('.mlp.gate.weight', '.block_sparse_moe.gate.weight')] = v
        elif shape.n_experts > 0 and k.endswith('.mlp.experts.gate_up_proj'):
            p = k[:-len('mlp.experts.gate_up_proj')] + 'block_sparse_moe.experts.'
            for e in range(shape.n_experts):
                out[f'{p}{e}.w1.weight'] = v[e, :shape.ffn]
                out[f'{p}{e}.w3.weight'] = v[e, shape.ffn:]
        elif shape.n_experts > 0 and k.endswith('.mlp.experts.down_proj'):
            p = k[:-len('mlp.experts.down_proj')] + 'block_sparse_moe.experts.'
            for e in range(shape.n_experts):
                out[f'{p}{e}.w2.weight'] = v[e]
        else:
            out[k] = v
    if 'lm_head.weight' not in out:
        out['lm_head.weight'] = out['model.embed_tokens.weight']
    return out


def load_hf_checkpoint(model_dir):
    """(transformers config, HF-named state dict on CPU) of a checkpoint directory: config.json plus *.safetensors (sharded or
    not) or pytorch_model*.bin — what LlamaForCausalLM.from_pretrained(model_dir, ...) of the reference's examples resolves
    (examples/llama_example.py:19-24).  Tensors are read shard by shard; nothing is instantiated as an nn.Module."""
    import glob
    import json
    import os
    from transformers import AutoConfig
    cfg = AutoConfig.from_pretrained(model_dir)
    sd = {}
    # consolidated*.safetensors (Mistral / Mixtral repos ship them NEXT to the HF shards, under other key names) would double
    # the host memory and leave keys nobody consumes
    st_files = sorted(f for f in glob.glob(os.path.join(model_dir, '*.safetensors'))
                      if not os.path.basename(f).startswith('consolidated'))
    idx = os.path.join(model_dir, 'model.safetensors.index.json')
    if os.path.exists(idx):
        names = sorted(set(json.load(open(idx))['weight_map'].values()))
        st_files = [os.path.join(model_dir, n) for n in names]
    if st_files:
        from safetensors import safe_open
        for f in st_files:
            with safe_open(f, framework='pt', device='cpu') as fh:
                for k in fh.keys():
                    sd[k] = fh.get_tensor(k)
    else:
        bins = sorted(glob.glob(os.path.join(model_dir, 'pytorch_model*.bin')))
        if not bins:
            raise FileNotFoundError(f'no *.safetensors or pytorch_model*.bin under {model_dir}')
        for f in bins:
            sd.update(torch.load(f, map_location='cpu', weights_only=True))
    return cfg, sd


def rope_tables(head_dim, max_pos, theta, device, dtype=torch.bfloat16):
    """cos/sin exactly as LlamaRotaryEmbedding.forward (modeling_llama.py:93-126): fp32 outer product,
    cos()/sin() in fp32, cast to the activation dtype (bfloat16 or float16).  Only the first head_dim/2 columns are stored
    (emb = cat(freqs, freqs))."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    pos = torch.arange(max_pos, dtype=torch.int64).float()
    freqs = (inv_freq[:, None].float() @ pos[None, :].float()).transpose(0, 1)      # [max_pos, hd/2]
    return freqs.cos().to(dtype).to(device).contiguous(), freqs.sin().to(dtype).to(device).contiguous()


def random_weights(shape, seed=0, std=0.02, device='cpu', dtype=torch.bfloat16, decisive=False):
    """Random-init weights in HF Llama naming (initializer_range=0.02, norms = 1), generated layer by layer on
    `device` (no checkpoints are available).

    decisive=True builds the synthetic "permutation LM" used by bench.py: same shapes and byte counts, but
    o_proj / down_proj are drawn with std 1e-4 (the residual stream stays dominated by the token embedding) and
    lm_head[pi(t)] = embed[t] for a fixed random permutation pi, so greedy decoding has a logit margin of several
    units instead of the ~3 bf16 ulps of a pure random-init model.  With pure random init, greedy and tree-verify
    decoding legitimately drift apart after ~20 tokens in bf16 (the reference README warns of the same in fp16/bf16,
    lookahead/README.md:45), which would make a trie warmed on the greedy continuation useless."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    hd = shape.head_dim

    def w(n, k, s=std):
        return (torch.randn(n, k, generator=g, device=device, dtype=torch.float32) * s).to(dtype)

    out_std = 1e-4 if decisive else std
    sd = {'model.embed_tokens.weight': w(shape.vocab, shape.hidden)}
    for i in range(shape.n_layers):
        p = f'model.layers.{i}.'
        sd[p + 'self_attn.q_proj.weight'] = w(shape.n_heads * hd, shape.hidden)
        sd[p + 'self_attn.k_proj.weight'] = w(shape.n_kv_heads * hd, shape.hidden)
        sd[p + 'self_attn.v_proj.weight'] = w(shape.n_kv_heads * hd, shape.hidden)
        sd[p + 'self_attn.o_proj.weight'] = w(shape.hidden, shape.n_heads * hd, out_std)
        if shape.n_experts > 0:                      # HF Mixtral naming (w1 = gate, w3 = up, w2 = down)
            sd[p + 'block_sparse_moe.gate.weight'] = w(shape.n_experts, shape.hidden)
            for e in range(shape.n_experts):
                q = p + f'block_sparse_moe.experts.{e}.'
                sd[q + 'w1.weight'] = w(shape.ffn, shape.hidden)
                sd[q + 'w3.weight'] = w(shape.ffn, shape.hidden)
                sd[q + 'w2.weight'] = w(shape.hidden, shape.ffn, out_std)
        else:
            sd[p + 'mlp.gate_proj.weight'] = w(shape.ffn, shape.hidden)
            sd[p + 'mlp.up_proj.weight'] = w(shape.ffn, shape.hidden)
            sd[p + 'mlp.down_proj.weight'] = w(shape.hidden, shape.ffn, out_std)
        sd[p + 'input_layernorm.weight'] = torch.ones(shape.hidden, device=device, dtype=dtype)
        sd[p + 'post_attention_layernorm.weight'] = torch.ones(shape.hidden, device=device, dtype=dtype)
    sd['model.norm.weight'] = torch.ones(shape.hidden, device=device, dtype=dtype)
    if decisive:
        # pi fixes the special ids 0..2 and is ONE cycle over [3, V): a sequence that starts from ordinary tokens never
        # emits token 0, whose drafts the reference mis-roots (`match_token_id or self.token_id`, lookahead_cache.py:129,
        # H1g), and never repeats itself within V-3 tokens (a short cycle would turn the warm-up text periodic)
        order = 3 + torch.randperm(shape.vocab - 3, generator=g, device=device)
        perm = torch.arange(shape.vocab, device=device)
        perm[order] = torch.roll(order, -1)
        head = torch.empty_like(sd['model.embed_tokens.weight'])
        head[perm] = sd['model.embed_tokens.weight']          # lm_head[pi(t)] = embed[t]
        sd['lm_head.weight'] = head
    else:
        sd['lm_head.weight'] = w(shape.vocab, shape.hidden)
    return sd


DEFAULT_LAB = False      # engines created without lab=... take the product library; the lab_build fixture of tests/conftest.py flips it for a variant test


class LlamaVerifyEngine(object):
    """One GPU: packed weights + KV cache + the captured step graph.  n_slots = 1: one sequence (the bs=1 loop).
    n_slots > 1: the cursor-batch path — every slot owns a max_keys region of the KV cache and the 64 rows of a
    verify block are shared by the active slots (bstep)."""

    def __init__(self, shape, state_dict, max_length=2048, device='cuda:0', attn_split=0, gemm_cfg=None,
                 consume_state_dict=False, balanced=True, n_slots=1, fuse=0, max_blocks=0, kv_ring=False, dtype=None, lab=None):
        if not torch.cuda.is_available():
            raise RuntimeError('LlamaVerifyEngine needs an MI355X: the verify step has no CPU fallback')
        # The engine computes in the checkpoint's own 16-bit type: bfloat16 (BASELINE) or float16 (what the reference's examples and
        # benchmarks load, benchmarks/llama_benchmark.py:27) — one library build per type (_lib.lib_for), never a silent conversion.
        # dtype=None: the dtype of the state dict's lm_head (fp32 state dicts run as bfloat16, as before).
        if dtype is None:
            dtype = state_dict['lm_head.weight'].dtype if state_dict and 'lm_head.weight' in state_dict else torch.bfloat16
            if dtype not in (torch.bfloat16, torch.float16):
                dtype = torch.bfloat16
        self.dtype = dtype
        # lab=True: the LAB build of the same sources (measurement knobs / A/B switches, _lib.lab_set) instead of the product library
        self.lab = DEFAULT_LAB if lab is None else bool(lab)
        self._lib = _lib.lab_lib_for(dtype) if self.lab else _lib.lib_for(dtype)
        self.shape = shape
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.kv_ring = bool(kv_ring)
        if self.kv_ring:
            # sliding-window ring (extension, SURVEY H3 / BASELINE config 3): KV memory O(window) per sequence instead of O(max_length)
            win = int(getattr(shape, 'sliding_window', 0))
            assert win > 0, 'kv_ring needs shape.sliding_window > 0'
            self.max_keys = int(math.ceil((win + 64 * max(int(max_blocks), 1) + 64) / 32.0)) * 32
            self.max_pos = max_length + 64 + 64 + 2      # _capacity() = max_length + 65, as the linear cache admits
        else:
            self.max_keys = int(math.ceil((max_length + 64 + 1) / 32.0)) * 32
            self.max_pos = self.max_keys + 64
        # a dedicated non-default stream: hipStreamBeginCapture is illegal on the legacy NULL stream that
        # torch.cuda.current_stream() returns by default
        self.stream = torch.cuda.Stream(self.device)
        sp = C.c_void_p(self.stream.cuda_stream)
        # Every kernel lays a head out as a 128-feature lane (RoPE pairs (d, d + 64), 8192-element Q / K / V fragments).  A model whose
        # heads are narrower (LlamaAttention is shape-generic, modeling_llama.py:189-308) runs in the same lanes: its q / k / v rows and
        # o_proj columns are spread over the lanes by la_head_lane_map (feature d -> lane d, its rotary partner d + hd/2 -> lane 64 + d,
        # zeros elsewhere) BEFORE packing; zero lanes add exact zeros to every dot product, and cfg.head_dim — the real one — sets the
        # softmax scale.  hd below is the LANE width every size is computed from.
        self.head_dim = int(shape.head_dim)
        if not (8 <= self.head_dim <= 128 and self.head_dim % 2 == 0):
            raise ValueError(f'head_dim={self.head_dim}: even values from 8 to 128 are supported (128-feature lanes)')
        hd = 128
        lane_src = np.zeros(128, dtype=np.int32)
        check(self._lib.la_head_lane_map(self.head_dim, lane_src.ctypes.data_as(_lib.pi32)), 'head_lane_map')
        self._lane_src = lane_src

        def to_lanes(w, n_h, dim):
            """[n_h * head_dim, K] rows (dim 0) or [N, n_h * head_dim] columns (dim 1) -> the same with 128 lanes per head"""
            if self.head_dim == 128:
                return w
            w = w.to(self.device)
            src = torch.from_numpy(lane_src.astype(np.int64)).to(self.device)
            idx = (torch.arange(n_h, device=self.device)[:, None] * self.head_dim + src.clamp(min=0)[None, :]).reshape(-1)
            keep = (src >= 0).repeat(n_h)
            out = w.index_select(dim, idx)
            shape_ = [1, 1]
            shape_[dim] = -1
            return out * keep.reshape(shape_).to(out.dtype)
        self._keep = []

        def dev(t):
            t = t.to(device=self.device, dtype=self.dtype).contiguous()
            self._keep.append(t)
            return t

        def pack(w, w2=None):
            w = w.to(device=self.device, dtype=self.dtype).contiguous()
            n, k = w.shape
            if w2 is not None:
                w2 = w2.to(device=self.device, dtype=self.dtype).contiguous()
            out = torch.empty((2 if w2 is not None else 1) * n * k, dtype=self.dtype, device=self.device)
            torch.cuda.synchronize(self.device)          # w was produced on torch's stream
            check(self._lib.la_pack_weight(sp, w.data_ptr(), w2.data_ptr() if w2 is not None else None, n, k,
                                     1 if w2 is not None else 0, out.data_ptr()), 'pack_weight')
            self.stream.synchronize()                    # w / w2 may be freed by the caller right after
            self._keep.append(out)
            return out

        def take(name):
            return state_dict.pop(name) if consume_state_dict else state_dict[name]

        # Balanced GEMMs: one workgroup per CU, rows dealt out by la_rowplan (0 = shape not balanceable -> classic grid)
        n_cu = torch.cuda.get_device_properties(self.device).multi_processor_count
        self.balanced_wg = [0, 0, 0]
        plans = {}
        plan_kinds = {}
        if balanced and not (gemm_cfg and len(gemm_cfg) > 1 and gemm_cfg[1] < 0):
            for slot, (kind, n_rows) in enumerate([(2, (shape.n_heads + 2 * shape.n_kv_heads) * hd), (1, shape.ffn),
                                                   (0, shape.vocab)]):
                nwg = n_cu
                n = self._lib.la_rowplan(kind, n_rows, nwg, None)
                if n > 0:
                    plan = np.zeros(n, dtype=np.int32)
                    check(min(self._lib.la_rowplan(kind, n_rows, nwg, plan.ctypes.data_as(_lib.pi32)), 0), 'rowplan')
                    plans[slot] = torch.from_numpy(plan).to(self.device)
                    plan_kinds[slot] = (kind, n_rows, nwg)
                    self.balanced_wg[slot] = nwg

        # Multi-block step (192-512 rows, MFMA-bound): when the one-workgroup-per-CU QKV plan leaves the 32-row MFMA blocks less
        # than 60 % full (GQA: Mistral / Mixtral 24 rows = 12 RoPE pairs per workgroup), a SECOND image over pairs / 32 workgroups
        # (every block full; 50 MB per layer at the Mistral shape) serves that path: Mistral bs=8 11.26 -> 10.45 ms per step
        # (scripts/gpu_qkv_wg_ab.sh, profiles/r03_qkv_mb_plan.txt).  LA_QKV_MB_WG overrides (0 = off).
        self.qkv_mb_wg = 0
        pairs = (shape.n_heads + 2 * shape.n_kv_heads) * hd // 2
        if max_blocks > 1 and self.balanced_wg[0] and hd == 128:
            per_wg = pairs // self.balanced_wg[0]
            # ... or leaves a pair count per workgroup that is not a multiple of 4 (Llama-2-13B: 30): the RoPE epilogue then stores 4-byte pieces and
            # 1/16 of the MFMA rows are padding; 240 full workgroups instead: 13B bs=4 10.57 -> 10.45 ms, bs=8 16.00 -> 15.60 (round 6, GPU call 16)
            want = pairs // 32 if (pairs % 32 == 0 and (per_wg / 32.0 < 0.6 or (per_wg % 4 != 0 and pairs // 32 >= 0.9 * self.balanced_wg[0]))) else 0
            if os.environ.get('LA_QKV_MB_WG') is not None:
                want = int(os.environ['LA_QKV_MB_WG'])
            if want > 0 and want % 16 == 0 and want < self.balanced_wg[0] and pairs % want == 0 and pairs // want <= 32:
                n = self._lib.la_rowplan(2, 2 * pairs, want, None)
                if n > 0:
                    plan = np.zeros(n, dtype=np.int32)
                    check(min(self._lib.la_rowplan(2, 2 * pairs, want, plan.ctypes.data_as(_lib.pi32)), 0), 'rowplan')
                    plans['qkv_mb'] = torch.from_numpy(plan).to(self.device)
                    plan_kinds['qkv_mb'] = (2, 2 * pairs, want)
                    self.qkv_mb_wg = want

        def pack_planned(slot, mats):
            """compact workgroup-major packing by the plan (la_pack_planned)"""
            kind, n_rows, nwg = plan_kinds[slot]
            mats = [m.to(device=self.device, dtype=self.dtype).contiguous() for m in mats]
            K = mats[0].shape[1]
            out = torch.empty(self._lib.la_planned_elems(kind, n_rows, K, nwg), dtype=self.dtype, device=self.device)
            torch.cuda.synchronize(self.device)
            check(self._lib.la_pack_planned(sp, mats[0].data_ptr(), mats[1].data_ptr() if len(mats) > 1 else None,
                                      plans[slot].data_ptr(), kind, n_rows, K, nwg, out.data_ptr()), 'pack_planned')
            self.stream.synchronize()
            self._keep.append(out)
            return out

        # rows of [Wq;Wk;Wv] are gathered so that each GEMM workgroup owns RoPE pairs (d, d+64): the QKV epilogue
        # applies RoPE and writes the attention fragments itself (la_gemm64_qkv)
        self.qkv_fused = not (gemm_cfg and len(gemm_cfg) > 1 and gemm_cfg[1] < 0)
        n_qkv = (shape.n_heads + 2 * shape.n_kv_heads) * hd
        perm_np = np.zeros(n_qkv, dtype=np.int32)
        check(self._lib.la_qkv_row_perm(shape.n_heads, shape.n_kv_heads, perm_np.ctypes.data_as(_lib.pi32)), 'qkv_row_perm')
        qkv_perm = torch.from_numpy(perm_np.astype(np.int64)).to(self.device)
        layers = (_lib.LlamaLayerWeightsC * shape.n_layers)()
        for i in range(shape.n_layers):
            p = f'model.layers.{i}.'
            qkv = torch.cat([to_lanes(take(p + 'self_attn.q_proj.weight').to(self.device), shape.n_heads, 0),
                             to_lanes(take(p + 'self_attn.k_proj.weight').to(self.device), shape.n_kv_heads, 0),
                             to_lanes(take(p + 'self_attn.v_proj.weight').to(self.device), shape.n_kv_heads, 0)], 0)
            if self.balanced_wg[0]:
                layers[i].wqkv = pack_planned(0, [qkv]).data_ptr()
                if self.qkv_mb_wg:
                    layers[i].wqkv_mb = pack_planned('qkv_mb', [qkv]).data_ptr()
            else:
                if self.qkv_fused:
                    qkv = qkv.index_select(0, qkv_perm)
                layers[i].wqkv = pack(qkv).data_ptr()
            del qkv
            layers[i].wo = pack(to_lanes(take(p + 'self_attn.o_proj.weight'), shape.n_heads, 1)).data_ptr()

            def pack_gateup(gate, up):
                return (pack_planned(1, [gate, up]) if self.balanced_wg[1] else pack(gate, up)).data_ptr()

            if shape.n_experts > 0:
                layers[i].router = dev(take(p + 'block_sparse_moe.gate.weight')).data_ptr()
                gu = (_lib.vp * shape.n_experts)()
                dn = (_lib.vp * shape.n_experts)()
                # the expert images of a layer are packed at equal spacing in ONE buffer per projection, which lets the
                # engine run all experts of a stage in a single launch (la_engine.cpp: ex_merged)
                gu_all = dn_all = None
                for e in range(shape.n_experts):
                    q = p + f'block_sparse_moe.experts.{e}.'
                    one_gu = pack_planned(1, [take(q + 'w1.weight'), take(q + 'w3.weight')]) if self.balanced_wg[1] \
                        else pack(take(q + 'w1.weight'), take(q + 'w3.weight'))
                    one_dn = pack(take(q + 'w2.weight'))
                    if gu_all is None:
                        gstride = (one_gu.numel() + 63) // 64 * 64
                        dstride = (one_dn.numel() + 63) // 64 * 64
                        gu_all = torch.zeros(shape.n_experts * gstride, dtype=self.dtype, device=self.device)
                        dn_all = torch.zeros(shape.n_experts * dstride, dtype=self.dtype, device=self.device)
                    gu_all[e * gstride:e * gstride + one_gu.numel()].copy_(one_gu)
                    dn_all[e * dstride:e * dstride + one_dn.numel()].copy_(one_dn)
                    gu[e] = gu_all.data_ptr() + 2 * e * gstride
                    dn[e] = dn_all.data_ptr() + 2 * e * dstride
                    self._keep = [t for t in self._keep if t is not one_gu and t is not one_dn]
                    del one_gu, one_dn
                torch.cuda.synchronize(self.device)
                self._keep.extend([gu, dn, gu_all, dn_all])
                layers[i].ex_gateup = C.cast(gu, C.POINTER(_lib.vp))
                layers[i].ex_down = C.cast(dn, C.POINTER(_lib.vp))
            else:
                layers[i].wgateup = pack_gateup(take(p + 'mlp.gate_proj.weight'), take(p + 'mlp.up_proj.weight'))
                layers[i].wdown = pack(take(p + 'mlp.down_proj.weight')).data_ptr()
            layers[i].norm1 = dev(take(p + 'input_layernorm.weight')).data_ptr()
            layers[i].norm2 = dev(take(p + 'post_attention_layernorm.weight')).data_ptr()
            torch.cuda.synchronize(self.device)
        self._layers = layers
        self.embed = dev(take('model.embed_tokens.weight'))
        self.rope_cos, self.rope_sin = rope_tables(self.head_dim, self.max_pos, shape.rope_theta, self.device, self.dtype)
        if self.head_dim < 128:          # 64 columns per position: the lanes past head_dim / 2 only ever meet zeros
            padc = torch.ones((self.max_pos, 64), dtype=self.dtype, device=self.device)
            pads = torch.zeros((self.max_pos, 64), dtype=self.dtype, device=self.device)
            padc[:, :self.head_dim // 2] = self.rope_cos
            pads[:, :self.head_dim // 2] = self.rope_sin
            self.rope_cos, self.rope_sin = padc.contiguous(), pads.contiguous()
        w = _lib.LlamaWeightsC()
        w.embed = self.embed.data_ptr()
        w.lm_head = (pack_planned(2, [take('lm_head.weight')]) if self.balanced_wg[2] else pack(take('lm_head.weight'))).data_ptr()
        w.final_norm = dev(take('model.norm.weight')).data_ptr()
        w.rope_cos, w.rope_sin = self.rope_cos.data_ptr(), self.rope_sin.data_ptr()
        w.layers = C.cast(layers, C.POINTER(_lib.LlamaLayerWeightsC))
        torch.cuda.synchronize(self.device)

        cfg = _lib.LlamaConfigC()
        cfg.n_layers, cfg.hidden, cfg.n_heads, cfg.n_kv_heads = shape.n_layers, shape.hidden, shape.n_heads, shape.n_kv_heads
        cfg.head_dim, cfg.ffn, cfg.vocab = self.head_dim, shape.ffn, shape.vocab
        cfg.max_keys, cfg.max_pos, cfg.attn_split, cfg.rms_eps = self.max_keys, self.max_pos, attn_split, shape.rms_eps
        for i, v in enumerate(gemm_cfg or []):
            cfg.gemm_cfg[i] = int(v)
        for i in range(3):
            cfg.balanced_wg[i] = self.balanced_wg[i]
        assert 1 <= n_slots <= _lib.LA_MAX_SEQ
        self.n_slots = int(n_slots)
        cfg.n_slots = self.n_slots
        cfg.n_experts, cfg.top_k = shape.n_experts, shape.top_k
        cfg.norm_cast_first = int(shape.norm_cast_first)
        cfg.qkv_mb_wg = int(self.qkv_mb_wg)
        cfg.fuse = int(fuse)
        assert 0 <= max_blocks <= _lib.LA_MB_MAX
        self.max_blocks = int(max_blocks) if max_blocks > 1 else 0
        cfg.max_blocks = self.max_blocks
        cfg.sliding_window = int(getattr(shape, 'sliding_window', 0))
        cfg.kv_ring = int(self.kv_ring)
        self._cfg = cfg
        nbytes = self._lib.la_llama_workspace_bytes(C.byref(cfg))
        if nbytes <= 0:
            raise _lib.LookaheadHipError(f'la_llama_workspace_bytes: {_lib.last_error()}')
        self.workspace = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        self._h = self._lib.la_llama_create(C.byref(cfg), C.byref(w), self.workspace.data_ptr(), nbytes)
        if not self._h:
            raise _lib.LookaheadHipError(f'la_llama_create: {_lib.last_error()}')
        self.host_in = torch.zeros(_lib.LA_IN_WORDS, dtype=torch.int32).pin_memory()
        self.host_out = torch.zeros(_lib.LA_ST_OUTTOK + 64, dtype=torch.int32).pin_memory()
        self._in_np = self.host_in.numpy()
        self._out_np = self.host_out.numpy()
        self._in_rm = self._in_np[_lib.LA_IN_ROWMASK:_lib.LA_IN_ROWMASK + 128].view(np.uint64)
        self.host_bin = torch.zeros(_lib.LA_BIN_WORDS, dtype=torch.int32).pin_memory()
        self.host_bout = torch.zeros(_lib.LA_BST_DST, dtype=torch.int32).pin_memory()
        self._bin_np = self.host_bin.numpy()
        self._bout_np = self.host_bout.numpy()
        self._bin_rm = self._bin_np[_lib.LA_BIN_ROWMASK:_lib.LA_BIN_ROWMASK + 128].view(np.uint64)
        if self.max_blocks:
            self.host_min = torch.zeros(_lib.LA_MIN_WORDS, dtype=torch.int32).pin_memory()
            self.host_mout = torch.zeros(_lib.LA_MOUT_DST, dtype=torch.int32).pin_memory()
            self._min_np = self.host_min.numpy()
            self._mout_np = self.host_mout.numpy()
            self._min_rm = self._min_np[_lib.LA_MIN_ROWMASK:_lib.LA_MIN_ROWMASK + 2 * 64 * _lib.LA_MB_MAX].view(np.uint64)
            self._min_xm = self._min_np[_lib.LA_MIN_XMASK:_lib.LA_MIN_XMASK + 2 * 3 * 64 * _lib.LA_MB_MAX].view(np.uint64).reshape(_lib.LA_MB_MAX * 64, 3)
        self.slot_keys = [0] * self.n_slots
        self.n_keys = 0
        self.reset()

    def __del__(self):
        h = getattr(self, '_h', None)
        if h:
            self._lib.la_llama_destroy(h)
            self._h = None

    # ------------------------------------------------------------------------------------------------
    def _sp(self):
        return C.c_void_p(self.stream.cuda_stream)

    def _capacity(self):
        """keys a sequence may hold: the cache size, or (ring) whatever the RoPE tables cover"""
        return self.max_pos - 65 if self.kv_ring else self.max_keys

    def reset(self):
        check(self._lib.la_llama_reset(self._h, self._sp()), 'llama_reset')
        self.n_keys = 0
        self.slot_keys = [0] * self.n_slots

    # ---- cursor batch ------------------------------------------------------------------------------------
    def reset_slot(self, slot):
        check(self._lib.la_llama_reset_slot(self._h, self._sp(), int(slot)), 'reset_slot')
        if slot < 0:
            self.slot_keys = [0] * self.n_slots
        else:
            self.slot_keys[slot] = 0

    def bstep(self, segments, eager=False):
        """One verify block shared by several sequences.  segments: list of (slot, ids, local_rowmask, mode, limit);
        local_rowmask bit j of row i = tree row i sees tree row j OF THE SAME SEGMENT.  Rows are laid out segment after
        segment (sum of lengths <= 64).  -> {slot: list of emitted tokens}."""
        a = self._bin_np
        a[_lib.LA_BIN_SEQ:_lib.LA_BIN_SEQ + 64] = -1
        a[_lib.LA_BIN_MODE:_lib.LA_BIN_MODE + 16] = 0
        a[_lib.LA_BIN_LIMIT:_lib.LA_BIN_LIMIT + 16] = 16
        row = 0
        seen = set()
        self._bstep_first = {}
        for slot, ids, rowmask, mode, limit in segments:
            n = len(ids)
            assert 0 <= slot < self.n_slots and slot not in seen and n >= 1, 'one segment per slot'
            seen.add(slot)
            self._bstep_first[slot] = row
            assert row + n <= _lib.LA_TREE_MAX, 'a verify block holds 64 rows'
            assert self.slot_keys[slot] + n <= self._capacity(), 'KV cache capacity of the slot exceeded'
            a[_lib.LA_BIN_IDS + row:_lib.LA_BIN_IDS + row + n] = ids
            self._bin_rm[row:row + n] = np.asarray(rowmask, dtype=np.uint64) << np.uint64(row)
            a[_lib.LA_BIN_SEQ + row:_lib.LA_BIN_SEQ + row + n] = slot
            a[_lib.LA_BIN_MODE + slot] = mode
            a[_lib.LA_BIN_LIMIT + slot] = max(1, min(16, int(limit)))
            row += n
        a[_lib.LA_BIN_T] = row
        fn = self._lib.la_llama_bstep_eager if eager else self._lib.la_llama_bstep
        check(fn(self._h, self._sp(), self.host_bin.data_ptr(), self.host_bout.data_ptr()), 'llama_bstep')
        self.stream.synchronize()
        o = self._bout_np
        out = {}
        for slot in seen:
            n_out = int(o[_lib.LA_BST_NOUT + slot])
            self.slot_keys[slot] = int(o[_lib.LA_BST_NKEYS + slot])
            out[slot] = o[_lib.LA_BST_OUTTOK + 16 * slot:_lib.LA_BST_OUTTOK + 16 * slot + n_out].tolist()
        return out

    def bstep_rows(self):
        """{slot: first block row} of the last bstep (segments are laid out one after another)."""
        return dict(self._bstep_first)

    def bcommit(self, kept):
        """After a bstep whose segments ran in mode 2 (forward only): kept = {slot: [local tree rows to keep, root first]};
        their K/V move to the slot's cursor (la_llama_bcommit), cursors advance."""
        keep = np.full(64, -1, dtype=np.int32)
        for slot, rows in kept.items():
            base = self._bstep_first[slot]
            for k, r in enumerate(rows):
                keep[base + int(r)] = k
            assert self.slot_keys[slot] + len(rows) <= self._capacity(), 'KV cache capacity of the slot exceeded'
        check(self._lib.la_llama_bcommit(self._h, self._sp(), keep.ctypes.data_as(_lib.pi32), self.host_bout.data_ptr()), 'llama_bcommit')
        for slot in kept:
            self.slot_keys[slot] = int(self._bout_np[_lib.LA_BST_NKEYS + slot])

    def mcommit(self, kept):
        """After an mstep whose blocks ran in mode 2: kept = list (one entry per block, block order) of the tree rows to keep,
        root first (la_llama_mcommit)."""
        slots = self._mstep_slots
        assert len(kept) == len(slots)
        keep = np.full(64 * len(kept), -1, dtype=np.int32)
        for b, rows in enumerate(kept):
            for k, r in enumerate(rows):
                keep[64 * b + int(r)] = k
        check(self._lib.la_llama_mcommit(self._h, self._sp(), len(kept), keep.ctypes.data_as(_lib.pi32), self.host_mout.data_ptr()),
              'llama_mcommit')
        for slot in slots:
            self.slot_keys[slot] = int(self._mout_np[_lib.LA_MOUT_NKEYS + slot])
        if 0 in slots:
            self.n_keys = self.slot_keys[0]

    def bprefill(self, slot, prompt_ids, eager=False):
        """Prompt of one slot as chains of <= 64 rows; -> first generated token."""
        prompt_ids = [int(x) for x in prompt_ids]
        tok = None
        for s in range(0, len(prompt_ids), 64):
            blk = prompt_ids[s:s + 64]
            tok = self.bstep([(slot, np.asarray(blk, dtype=np.int32), self._CHAIN[:len(blk)], 1, 1)], eager=eager)[slot][0]
        return tok

    def bprefill_many(self, prompts, eager=False):
        """prompts: {slot: token list}.  The chains of all slots are packed greedily into shared 64-row blocks (a
        slot contributes one contiguous run per block and continues in the next).  -> {slot: first generated token}."""
        todo = {s: [int(x) for x in p] for s, p in prompts.items()}
        assert all(len(p) > 0 for p in todo.values())
        done, first = {s: 0 for s in todo}, {}
        order = sorted(todo)
        while any(done[s] < len(todo[s]) for s in order):
            segments, room = [], _lib.LA_TREE_MAX
            for s in order:
                left = len(todo[s]) - done[s]
                if left == 0 or room == 0:
                    continue
                n = min(left, room)
                segments.append((s, np.asarray(todo[s][done[s]:done[s] + n], dtype=np.int32), self._CHAIN[:n], 1, 1))
                done[s] += n
                room -= n
            out = self.bstep(segments, eager=eager)
            for s, _, _, _, _ in segments:
                if done[s] == len(todo[s]):
                    first[s] = out[s][0]
        return first

    # ---- multi-block step: every block is a full 64-row verify block of its own (la_llama_mstep) ---------------------
    def mstep(self, blocks, eager=False):
        """blocks: list of (slot, ids, rowmask, mode, limit) — one sequence's draft tree each (mode 0), or consecutive
        64-token pieces of one prompt (mode 1, same slot: a causal chain; every piece but the last holds 64 rows).
        All blocks run in ONE pass over the weights (M = 64 * len(blocks) rows).  -> list of emitted token lists.
        = mstep_async + mstep_finish."""
        self.mstep_async(blocks, eager=eager)
        return self.mstep_finish()

    def mstep_async(self, blocks, eager=False):
        """Queue the multi-block pass (input block H2D, captured graph, result header D2H) and return at once: the host is free —
        the previous step's trie update (LookaheadCache.stream_put_many), the accepted-token gather of an N-rank job
        (AcceptedTokenGather.finish_into_trie), bookkeeping — until mstep_finish()."""
        assert self.max_blocks and 1 <= len(blocks) <= self.max_blocks, 'engine was created with max_blocks < len(blocks)'
        a = self._min_np
        a[_lib.LA_MIN_NBLK] = len(blocks)
        used = {}
        for b, (slot, ids, rowmask, mode, limit) in enumerate(blocks):
            n = len(ids)
            assert 0 <= slot < self.n_slots and 1 <= n <= 64
            prev = used.get(slot)
            assert prev is None or (mode == 1 and prev == (1, 64)), 'several blocks of one slot must form a prefill chain of full blocks'
            used[slot] = (mode, n)
            a[_lib.LA_MIN_BLK + 4 * b:_lib.LA_MIN_BLK + 4 * b + 4] = (slot, n, mode, max(1, min(_lib.LA_MOUT_TOKS, int(limit))))
            a[_lib.LA_MIN_IDS + 64 * b:_lib.LA_MIN_IDS + 64 * b + n] = ids
            self._min_rm[64 * b:64 * b + n] = rowmask
        rows = {}
        self._mstep_slots = [blk[0] for blk in blocks]
        for slot, ids, _, _, _ in blocks:
            rows[slot] = rows.get(slot, 0) + len(ids)
        for slot, n in rows.items():
            assert self.slot_keys[slot] + n <= self._capacity(), 'KV cache capacity of the slot exceeded'
        fn = self._lib.la_llama_mstep_eager if eager else self._lib.la_llama_mstep
        check(fn(self._h, self._sp(), self.host_min.data_ptr(), self.host_mout.data_ptr()), 'llama_mstep')
        self._mstep_pending = (len(blocks), list(rows))

    def mstep_finish(self):
        """Wait for the pass queued by mstep_async -> list of emitted token lists (block order)."""
        nb, slots = self._mstep_pending
        self._mstep_pending = None
        self.stream.synchronize()
        o = self._mout_np
        for slot in slots:
            self.slot_keys[slot] = int(o[_lib.LA_MOUT_NKEYS + slot])
        if 0 in slots:
            self.n_keys = self.slot_keys[0]
        return [o[_lib.LA_MOUT_OUTTOK + _lib.LA_MOUT_TOKS * b:_lib.LA_MOUT_OUTTOK + _lib.LA_MOUT_TOKS * b + int(o[_lib.LA_MOUT_NOUT + b])].tolist()
                for b in range(nb)]

    def mstep_trie(self, dev_trie, q0, slots, limits, last_tokens, put_idxs=None, put_branch_length=None):
        """One multi-block verify step whose drafts are the results of the LAST dev_trie.hier_get_dev(...) launch, taken on the
        device: block b = query q0 + b of that launch (ids / row masks / count stay in HBM; nothing but the result header crosses
        PCIe).  The trie kernels must have been queued on this engine's stream.  slots / limits / last_tokens: per block.
        put_idxs: the trie slot (batch index) of each block — the step's accepted tokens are then inserted into the DEVICE trie
        image by the device, straight from the step's output block (DeviceTrie.stream_put_dev, queued behind the step; the host
        waits for the result header only and must dev_trie.replay() the same tokens before its next trie update).
        -> (emitted token lists, draft lengths).  = mstep_trie_async + mstep_trie_finish."""
        self.mstep_trie_async(dev_trie, q0, slots, limits, last_tokens, put_idxs, put_branch_length)
        return self.mstep_trie_finish()

    def mstep_trie_async(self, dev_trie, q0, slots, limits, last_tokens, put_idxs=None, put_branch_length=None):
        """Queue the chained step (and the device-side trie update behind it) and return at once: the host is free — e.g. to
        replay the PREVIOUS step's trie update on its own trie (DeviceTrie.replay) — until mstep_trie_finish()."""
        nb = len(slots)
        assert self.max_blocks and 1 <= nb <= self.max_blocks
        assert getattr(dev_trie, 'rows', 64) == 64, 'the chained step reads 64-row result blocks (DeviceTrie(max_rows=64), decoding_length <= 64)'
        arr = lambda v: (C.c_int32 * nb)(*[int(x) for x in v])
        lim = [max(1, min(_lib.LA_MOUT_TOKS, int(x))) for x in limits]
        check(self._lib.la_llama_mstep_trie(self._h, self._sp(), nb, arr(slots), arr(lim), arr(last_tokens),
                                      C.c_void_p(dev_trie.out_ids.data_ptr() + 4 * 64 * q0), C.c_void_p(dev_trie.out_rm.data_ptr() + 8 * 64 * q0),
                                      C.c_void_p(dev_trie.out_n.data_ptr() + 4 * q0), self.host_mout.data_ptr()), 'llama_mstep_trie')
        self._trie_wait = None
        if put_idxs is not None:
            ev = getattr(self, '_hdr_event', None)
            if ev is None:
                ev = self._hdr_event = torch.cuda.Event()
            ev.record(self.stream)                      # behind the D2H of the result header
            mo = getattr(self, '_mout_ptr', None)
            if mo is None:
                mo = self._mout_ptr = self.mout().data_ptr()
            with torch.cuda.stream(self.stream):
                dev_trie.stream_put_dev(mo + 4 * _lib.LA_MOUT_OUTTOK, _lib.LA_MOUT_TOKS, mo + 4 * _lib.LA_MOUT_NOUT, put_idxs,
                                        put_branch_length)
            self._trie_wait = ev
        self._trie_slots = list(slots)

    def mstep_trie_finish(self):
        """Wait for the result header of the step queued by mstep_trie_async -> (emitted token lists, draft lengths)."""
        if self._trie_wait is not None:
            self._trie_wait.synchronize()
        else:
            self.stream.synchronize()
        slots = self._trie_slots
        nb = len(slots)
        o = self._mout_np
        self._mstep_slots = list(slots)
        for slot in slots:
            self.slot_keys[slot] = int(o[_lib.LA_MOUT_NKEYS + slot])
        if 0 in slots:
            self.n_keys = self.slot_keys[0]
        toks = [o[_lib.LA_MOUT_OUTTOK + _lib.LA_MOUT_TOKS * b:_lib.LA_MOUT_OUTTOK + _lib.LA_MOUT_TOKS * b + int(o[_lib.LA_MOUT_NOUT + b])].tolist()
                for b in range(nb)]
        return toks, [int(o[_lib.LA_MOUT_T + b]) for b in range(nb)]

    # ---- wide trees: one sequence's tree of up to LA_TREE_WIDE_MAX rows as consecutive blocks of one multi-block pass -------
    def tstep(self, ids, rowmask, slot=0, mode=0, limit=None, eager=False):
        """One verify step of ONE sequence whose draft tree may be wider than a 64-row block (the reference grid-searches
        decoding_length up to 256 and publishes its best numbers at decoding_length=128, branch_length=32: README.md:100,
        benchmarks/benchmark.py:256-288).  ids int32[T], rowmask uint64[T] or uint64[T][W] (word w of row i = tree columns
        64 w ...): rows 64 p .. 64 p + 63 form block p of the pass — block 0 in `mode` (0 verify / 2 forward only), the others as
        LA_MODE_TREE_PIECE with their ancestor words over the earlier blocks in LA_MIN_XMASK.  -> (emitted tokens, keys kept)."""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        T = len(ids)
        rm = np.asarray(rowmask, dtype=np.uint64)
        if rm.ndim == 1:
            rm = rm[:, None]
        nb = (T + 63) // 64
        assert 1 <= T <= _lib.LA_TREE_WIDE_MAX and nb <= min(self.max_blocks, 4), \
            f'a {T}-row tree needs an engine created with max_blocks >= {nb} (has {self.max_blocks}; at most 4 blocks per tree)'
        assert 0 <= slot < self.n_slots and mode in (0, 2)
        assert self.slot_keys[slot] + T <= self._capacity(), 'KV cache capacity of the slot exceeded'
        lim = _lib.LA_MOUT_TOKS if limit is None else max(1, min(_lib.LA_MOUT_TOKS, int(limit)))
        a = self._min_np
        a[_lib.LA_MIN_NBLK] = nb
        for p in range(nb):
            r0, r1 = 64 * p, min(T, 64 * p + 64)
            n = r1 - r0
            a[_lib.LA_MIN_BLK + 4 * p:_lib.LA_MIN_BLK + 4 * p + 4] = (slot, n, mode if p == 0 else _lib.LA_MODE_TREE_PIECE, lim)
            a[_lib.LA_MIN_IDS + 64 * p:_lib.LA_MIN_IDS + 64 * p + n] = ids[r0:r1]
            self._min_rm[64 * p:64 * p + n] = rm[r0:r1, p] if p < rm.shape[1] else 0
            for q in range(p):
                self._min_xm[64 * p:64 * p + n, q] = rm[r0:r1, q]
        self._mstep_slots = [slot] * nb
        fn = self._lib.la_llama_mstep_eager if eager else self._lib.la_llama_mstep
        check(fn(self._h, self._sp(), self.host_min.data_ptr(), self.host_mout.data_ptr()), 'llama_mstep')
        self.stream.synchronize()
        o = self._mout_np
        before = self.slot_keys[slot]
        self.slot_keys[slot] = int(o[_lib.LA_MOUT_NKEYS + slot])
        if slot == 0:
            self.n_keys = self.slot_keys[0]
        return o[_lib.LA_MOUT_OUTTOK:_lib.LA_MOUT_OUTTOK + int(o[_lib.LA_MOUT_NOUT])].tolist(), self.slot_keys[slot] - before

    def mstep_trees(self, trees, eager=False):
        """Several sequences' draft trees in ONE pass, each tree of up to LA_TREE_WIDE_MAX rows (a batch whose per-sample budget
        exceeds a 64-row block: the reference's bat_get hands every sample (decoding_length // bs) // bs rows, any size,
        lookahead_cache.py:534-541).  trees: list of (slot, ids int32[T], rowmask uint64[T] or uint64[T][W], mode 0 / 2, limit):
        a tree occupies ceil(T / 64) consecutive blocks of the pass (block 0 in `mode`, the others LA_MODE_TREE_PIECE), one tree
        per slot, all blocks together <= max_blocks.  -> list of emitted token lists (tree order)."""
        a = self._min_np
        b0s, slots, b = [], set(), 0
        for slot, ids, rowmask, mode, limit in trees:
            ids = np.ascontiguousarray(ids, dtype=np.int32)
            T = len(ids)
            rm = np.asarray(rowmask, dtype=np.uint64)
            if rm.ndim == 1:
                rm = rm[:, None]
            nb = (T + 63) // 64
            assert 1 <= T <= _lib.LA_TREE_WIDE_MAX and nb <= 4, f'a tree holds 1..{_lib.LA_TREE_WIDE_MAX} rows (got {T})'
            assert 0 <= slot < self.n_slots and slot not in slots and mode in (0, 2), 'one tree per slot, mode 0 or 2'
            assert self.slot_keys[slot] + T <= self._capacity(), 'KV cache capacity of the slot exceeded'
            slots.add(slot)
            lim = max(1, min(_lib.LA_MOUT_TOKS, int(limit)))
            b0s.append((b, slot))
            for p_ in range(nb):
                r0, r1 = 64 * p_, min(T, 64 * p_ + 64)
                n = r1 - r0
                a[_lib.LA_MIN_BLK + 4 * b:_lib.LA_MIN_BLK + 4 * b + 4] = (slot, n, mode if p_ == 0 else _lib.LA_MODE_TREE_PIECE, lim)
                a[_lib.LA_MIN_IDS + 64 * b:_lib.LA_MIN_IDS + 64 * b + n] = ids[r0:r1]
                self._min_rm[64 * b:64 * b + n] = rm[r0:r1, p_] if p_ < rm.shape[1] else 0
                for q in range(p_):
                    self._min_xm[64 * b:64 * b + n, q] = rm[r0:r1, q] if q < rm.shape[1] else 0
                b += 1
        assert self.max_blocks and 1 <= b <= self.max_blocks, f'{b} blocks in one pass need an engine created with max_blocks >= {b}'
        a[_lib.LA_MIN_NBLK] = b
        self._mstep_slots = []
        for (b0, slot), (_, ids, _, _, _) in zip(b0s, trees):
            self._mstep_slots.extend([slot] * ((len(ids) + 63) // 64))
        fn = self._lib.la_llama_mstep_eager if eager else self._lib.la_llama_mstep
        check(fn(self._h, self._sp(), self.host_min.data_ptr(), self.host_mout.data_ptr()), 'llama_mstep')
        self.stream.synchronize()
        o = self._mout_np
        for slot in slots:
            self.slot_keys[slot] = int(o[_lib.LA_MOUT_NKEYS + slot])
        if 0 in slots:
            self.n_keys = self.slot_keys[0]
        return [o[_lib.LA_MOUT_OUTTOK + _lib.LA_MOUT_TOKS * b0:_lib.LA_MOUT_OUTTOK + _lib.LA_MOUT_TOKS * b0 + int(o[_lib.LA_MOUT_NOUT + b0])].tolist()
                for b0, _ in b0s]

    def mcommit_trees(self, kept, n_rows):
        """After mstep_trees(mode 2): kept[i] = the tree rows of tree i to keep (root first, path order), n_rows[i] = its row count
        — the host-walked sequential accept path for several (possibly wide) trees of one pass (la_llama_mcommit)."""
        nbs = [(int(n) + 63) // 64 for n in n_rows]
        keep = np.full(64 * sum(nbs), -1, dtype=np.int32)
        b0 = 0
        for rows, nb in zip(kept, nbs):
            for k, r in enumerate(rows):
                keep[64 * b0 + int(r)] = k
            b0 += nb
        check(self._lib.la_llama_mcommit(self._h, self._sp(), sum(nbs), keep.ctypes.data_as(_lib.pi32), self.host_mout.data_ptr()),
              'llama_mcommit')
        for slot in set(self._mstep_slots):
            self.slot_keys[slot] = int(self._mout_np[_lib.LA_MOUT_NKEYS + slot])
        if 0 in self._mstep_slots:
            self.n_keys = self.slot_keys[0]

    def tcommit(self, rows, n_rows):
        """After tstep(mode=2) over an n_rows-row tree: keep the tree rows `rows` (root first, path order) — the host-walked
        sequential accept path on a wide tree (la_llama_mcommit over the tree's blocks)."""
        nb = (int(n_rows) + 63) // 64
        keep = np.full(64 * nb, -1, dtype=np.int32)
        for k, r in enumerate(rows):
            keep[int(r)] = k
        check(self._lib.la_llama_mcommit(self._h, self._sp(), nb, keep.ctypes.data_as(_lib.pi32), self.host_mout.data_ptr()), 'llama_mcommit')
        slot = self._mstep_slots[0]
        self.slot_keys[slot] = int(self._mout_np[_lib.LA_MOUT_NKEYS + slot])
        if slot == 0:
            self.n_keys = self.slot_keys[0]

    def mprefill(self, slot, prompt_ids, eager=False):
        """Prompt of one slot in passes of up to max_blocks x 64 tokens (one pass over the weights each); -> first
        generated token.  Slot 0 hands its cursor to the single-sequence step as well (la_llama_set_nkeys)."""
        prompt_ids = [int(x) for x in prompt_ids]
        tok = None
        per = 64 * self.max_blocks
        for s in range(0, len(prompt_ids), per):
            piece = prompt_ids[s:s + per]
            blocks = [(slot, np.asarray(piece[i:i + 64], dtype=np.int32), self._CHAIN[:len(piece[i:i + 64])], 1, 1)
                      for i in range(0, len(piece), 64)]
            tok = self.mstep(blocks, eager=eager)[-1][0]
        if slot == 0:
            check(self._lib.la_llama_set_nkeys(self._h, self._sp(), 0, self.slot_keys[0]), 'set_nkeys')
            self.n_keys = self.slot_keys[0]
        return tok

    def mprefill_many(self, prompts, eager=False):
        """prompts: {slot: token list}.  The 64-token pieces of all prompts are dealt, slot after slot, into passes of up to
        max_blocks blocks (one pass over the weights each); a slot whose prompt spans several passes continues from its
        committed keys.  -> {slot: first generated token}."""
        todo = {s: [int(x) for x in p] for s, p in prompts.items()}
        assert all(len(p) > 0 for p in todo.values())
        pieces = []                                   # (slot, piece, is_last)
        for s in sorted(todo):
            p = todo[s]
            for i in range(0, len(p), 64):
                pieces.append((s, p[i:i + 64], i + 64 >= len(p)))
        first = {}
        for i in range(0, len(pieces), self.max_blocks):
            group = pieces[i:i + self.max_blocks]
            out = self.mstep([(s, np.asarray(pc, dtype=np.int32), self._CHAIN[:len(pc)], 1, 1) for s, pc, _ in group], eager=eager)
            for (s, _, last), toks in zip(group, out):
                if last:
                    first[s] = toks[0]
        if 0 in todo:
            check(self._lib.la_llama_set_nkeys(self._h, self._sp(), 0, self.slot_keys[0]), 'set_nkeys')
            self.n_keys = self.slot_keys[0]
        return first

    def mlogits(self):
        """bf16 [max_blocks * 64][vocab] of the last multi-block step (row = 64 * block + tree row)."""
        return self._view(11, _lib.LA_MB_MAX * 64 * self.shape.vocab * 2, self.dtype).view(_lib.LA_MB_MAX * 64, self.shape.vocab)

    def mout(self):
        return self._view(12, _lib.LA_MOUT_WORDS * 4, torch.int32)

    def bstate(self):
        return self._view(8, _lib.LA_BST_WORDS * 4, torch.int32)

    def _fill(self, ids, rowmask, mode):
        T = len(ids)
        assert 1 <= T <= _lib.LA_TREE_MAX
        assert self.n_keys + T <= self._capacity(), 'KV cache capacity exceeded'
        a = self._in_np
        a[_lib.LA_IN_T] = T
        a[_lib.LA_IN_MODE] = mode
        a[_lib.LA_IN_NKEYS_HINT] = self.n_keys               # host-side hint: which tree-attention form the step graph uses
        a[_lib.LA_IN_IDS:_lib.LA_IN_IDS + T] = ids
        self._in_rm[:T] = rowmask

    def step_async(self, ids, rowmask, mode=0, eager=False):
        """Enqueue one block (tree of T<=64 tokens) on the current stream; results land in host_out."""
        self._fill(ids, rowmask, mode)
        fn = self._lib.la_llama_step_eager if eager else self._lib.la_llama_step
        self._eager_pending = bool(eager)
        check(fn(self._h, self._sp(), self.host_in.data_ptr(), self.host_out.data_ptr()), 'llama_step')

    def step(self, ids, rowmask, mode=0, eager=False):
        """-> list of emitted tokens (accepted path + bonus), list of accepted tree rows."""
        self.step_async(ids, rowmask, mode, eager)
        return self.step_finish()

    def _wait(self):
        # captured step: the last kernel writes the result block into pinned host memory and bumps its sequence word, which
        # la_llama_wait polls (no D2H copy command, no sleep on the stream); eager steps use copies + a stream sync
        if getattr(self, '_eager_pending', False):
            self.stream.synchronize()
        else:
            check(self._lib.la_llama_wait(self._h, self._sp()), 'llama_wait')

    def step_finish(self):
        """Wait for the block enqueued by step_async and return its result (host work can run in between)."""
        self._wait()
        o = self._out_np
        n_out = int(o[_lib.LA_ST_NOUT])
        self.n_keys = int(o[_lib.LA_ST_NKEYS])
        return o[_lib.LA_ST_OUTTOK:_lib.LA_ST_OUTTOK + n_out].tolist(), int(o[_lib.LA_ST_NCOMMIT])

    def decode_native(self, cache, seq, max_length, eos_ids=(), decoding_length=64, branch_length=12, max_query_length=2,
                      mode=_lib.LA_MODE_MIX, idx=0, max_steps=1 << 30):
        """Run verify steps in the native loop (la_lookahead_decode: trie query -> captured graph -> trie update, no
        interpreter in between) until max_length / eos / max_steps.  `seq` = prompt + first generated token, already
        prefilled.  -> (new tokens, dls, edls, fts, qts, finished)."""
        p = _lib.DecodeParamsC()
        p.decoding_length, p.branch_length, p.max_query_length = int(decoding_length), int(branch_length), int(max_query_length)
        p.mode, p.idx, p.max_length = int(mode), int(idx), int(max_length)
        eos_ids = [int(e) for e in eos_ids if e is not None]
        if len(eos_ids) > 8:
            raise ValueError('decode_native: la_decode_params holds at most 8 eos ids (the interpreter loop has no such limit)')
        p.n_eos = len(eos_ids)
        for i, e in enumerate(eos_ids):
            p.eos[i] = e
        cap = max(1, min(int(max_steps), max(int(max_length) - len(seq), 0) + 1))
        p.max_steps = cap
        buf = np.zeros(max(int(max_length), len(seq)) + 32, dtype=np.int32)
        buf[:len(seq)] = seq
        n = C.c_int32(len(seq))
        dls = np.zeros(cap, dtype=np.int32)
        edls = np.zeros(cap, dtype=np.int32)
        fts = np.zeros(cap, dtype=np.float64)
        qts = np.zeros(cap, dtype=np.float64)
        steps, fin = C.c_int32(0), C.c_int32(0)
        dp = C.POINTER(C.c_double)
        check(self._lib.la_lookahead_decode(self._h, cache._h, self._sp(), C.byref(p), buf.ctypes.data_as(_lib.pi32), C.byref(n),
                                      self.host_in.data_ptr(), self.host_out.data_ptr(), dls.ctypes.data_as(_lib.pi32),
                                      edls.ctypes.data_as(_lib.pi32), C.byref(steps), C.byref(fin),
                                      fts.ctypes.data_as(dp), qts.ctypes.data_as(dp)), 'lookahead_decode')
        k = steps.value
        self.n_keys = int(self._out_np[_lib.LA_ST_NKEYS]) if k else self.n_keys
        return (buf[len(seq):n.value].tolist(), dls[:k].tolist(), edls[:k].tolist(), fts[:k].tolist(), qts[:k].tolist(),
                bool(fin.value))

    def verify_only(self, ids, rowmask, eager=False):
        """Forward of one block without the device accept walk (mode 2): logits() holds one row per tree token, nothing is
        committed until commit()."""
        self.step_async(ids, rowmask, mode=2, eager=eager)
        self._wait()

    def commit(self, rows):
        """Keep the K/V of tree rows `rows` (root first) of the last verify_only block."""
        arr = np.ascontiguousarray(rows, dtype=np.int32)
        assert self.n_keys + len(arr) <= self._capacity(), 'KV cache capacity exceeded'
        check(self._lib.la_llama_commit(self._h, self._sp(), arr.ctypes.data_as(_lib.pi32), len(arr), self.host_out.data_ptr()),
              'llama_commit')
        self.n_keys = int(self._out_np[_lib.LA_ST_NKEYS])

    _CHAIN = np.array([(2 << t) - 1 for t in range(63)] + [0xFFFFFFFFFFFFFFFF], dtype=np.uint64)

    def prefill(self, prompt_ids, eager=False, fast=None):
        """Process the prompt; returns the first generated token (argmax of the last prompt row,
        pretrained_model.py:783-798).  fast (default: engines created with max_blocks > 1, prompts of more than one block):
        chains of up to max_blocks x 64 tokens per pass over the weights (la_llama_mstep); otherwise 64-token steps, whose
        last block's logits stay readable through logits()."""
        prompt_ids = [int(x) for x in prompt_ids]
        if fast is None:
            fast = bool(self.max_blocks) and len(prompt_ids) > 64
        if fast:
            return self.mprefill(0, prompt_ids, eager=eager)
        tok = None
        for s in range(0, len(prompt_ids), 64):
            blk = prompt_ids[s:s + 64]
            toks, _ = self.step(np.asarray(blk, dtype=np.int32), self._CHAIN[:len(blk)], mode=1, eager=eager)
            tok = toks[0]
        return tok

    # ---- introspection for parity tests ----------------------------------------------------------------
    def _view(self, which, nbytes, dtype):
        ptr = self._lib.la_llama_buffer(self._h, which)
        off = ptr - self.workspace.data_ptr()
        return self.workspace[off:off + nbytes].view(dtype)

    def logits(self):
        """bf16 [64][vocab] of the last block (row t = tree token t)."""
        return self._view(0, 64 * self.shape.vocab * 2, self.dtype).view(64, self.shape.vocab)

    def state(self):
        return self._view(1, _lib.LA_ST_WORDS * 4, torch.int32)

    def hidden(self):
        return self._view(2, 64 * self.shape.hidden * 2, self.dtype).view(64, self.shape.hidden)

    def route_weights(self):
        """fp32 [n_layers][64][8]: routing weight of every (layer, block row, expert) of the last block (0 = not routed)."""
        L = self.shape.n_layers
        return self._view(9, L * 64 * _lib.LA_MOE_MAX_E * 4, torch.float32).view(L, 64, _lib.LA_MOE_MAX_E)

    def mroute_weights(self):
        """fp32 [n_layers][LA_MB_MAX * 64][8]: routing weights of the last multi-block step (row = 64 * block + block row)"""
        L = self.shape.n_layers
        return self._view(14, L * _lib.LA_MB_MAX * 64 * _lib.LA_MOE_MAX_E * 4, torch.float32).view(L, _lib.LA_MB_MAX * 64, _lib.LA_MOE_MAX_E)

    def profile_gateup(self, iters=5):
        """mean ms of one gate/up launch, every layer's launch back to back inside one HIP-event pair (la_llama_profile_gateup)"""
        ms = C.c_float(0)
        check(self._lib.la_llama_profile_gateup(self._h, self._sp(), int(iters), C.byref(ms)), 'profile_gateup')
        return float(ms.value)

    def profile(self, ids, rowmask, iters=3):
        """HIP-event timing per kernel class (see la_llama_profile)."""
        self._fill(ids, rowmask, 0)
        ms = (C.c_float * 8)()
        launches = (C.c_int32 * 7)()
        check(self._lib.la_llama_profile(self._h, self._sp(), self.host_in.data_ptr(), int(iters), ms, launches), 'profile')
        names = ['qkv', 'o', 'gateup', 'down', 'lm_head', 'attn', 'other']
        return {'ms': {n: ms[i] for i, n in enumerate(names)}, 'ms_step': ms[7],
                'launches': {n: launches[i] for i, n in enumerate(names)}}
