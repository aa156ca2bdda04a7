# -*- coding: utf-8 -*-
"""ctypes binding of liblookahead_hip.so (the C ABI declared in include/lookahead_hip.h).

The library is the product: there is no Python/CPU fallback for any device entry point.  If the
shared object is missing this module raises at import time with the build command.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# The PRODUCT libraries: every lab knob a constexpr default, no la_lab_* entry point (csrc/la_knobs.h).  The LAB builds of the same sources
# (-DLA_LAB=1) carry the measurement knobs / A/B switches of include/lookahead_hip_lab.h: engines take them with lab=True (the variant tests),
# or the whole process with LA_LAB_BUILD=1 in the environment (the A/B scripts under scripts/, bench.py with LA_DEBUG / LA_PF_KIB).
LAB_BUILD = os.environ.get('LA_LAB_BUILD', '') not in ('', '0')
LAB_PATH = os.path.join(_HERE, "liblookahead_hip_lab.so")
LAB_PATH_F16 = os.path.join(_HERE, "liblookahead_hip_lab_f16.so")
LIB_PATH = LAB_PATH if LAB_BUILD else os.path.join(_HERE, "liblookahead_hip.so")       # bfloat16 build (BASELINE's dtype; also serves the dtype-free trie calls)
LIB_PATH_F16 = LAB_PATH_F16 if LAB_BUILD else os.path.join(_HERE, "liblookahead_hip_f16.so")      # float16 build: the same sources compiled with -DLA_DTYPE=1
LA_DTYPE_BF16, LA_DTYPE_F16 = 0, 1

LA_OK = 0
ABI_VERSION = 12        # LA_ABI_VERSION of include/lookahead_hip.h these bindings were written against
LA_MODE_INPUT, LA_MODE_OUTPUT, LA_MODE_MIX = 0, 1, 2
LA_TREE_MAX = 64
LA_MOE_MAX_E = 8
# device step-state words (include/lookahead_hip.h)
LA_ST_NKEYS, LA_ST_T, LA_ST_MODE, LA_ST_NOUT, LA_ST_DSTBASE, LA_ST_NCOMMIT, LA_ST_MAXKEYS, LA_ST_SEQ = 0, 1, 2, 3, 4, 5, 6, 7
LA_ST_OUTTOK, LA_ST_SRCIDX, LA_ST_ARGMAX, LA_ST_WORDS = 8, 72, 136, 200
LA_IN_T, LA_IN_MODE, LA_IN_IDS, LA_IN_ROWMASK, LA_IN_WORDS = 0, 1, 4, 68, 196
LA_IN_NKEYS_HINT = 2
# cursor-batch blocks
LA_MAX_SEQ = 16
LA_BIN_T, LA_BIN_IDS, LA_BIN_ROWMASK, LA_BIN_SEQ, LA_BIN_MODE, LA_BIN_LIMIT, LA_BIN_WORDS = 0, 4, 68, 196, 260, 276, 292
LA_MB_MAX = 8
LA_MIN_NBLK, LA_MIN_BLK, LA_MIN_IDS, LA_MIN_ROWMASK, LA_MIN_XMASK, LA_MIN_WORDS = 0, 4, 36, 548, 1572, 4644
LA_TREE_WIDE_MAX, LA_MODE_TREE_PIECE, LA_MOUT_TOKS = 256, 3, 40
LA_MOUT_NOUT, LA_MOUT_NKEYS, LA_MOUT_OUTTOK, LA_MOUT_T, LA_MOUT_DST, LA_MOUT_ARGMAX, LA_MOUT_WORDS = 0, 8, 24, 344, 352, 864, 1376
LA_BST_NKEYS, LA_BST_NOUT, LA_BST_OUTTOK, LA_BST_DST, LA_BST_ARGMAX, LA_BST_SEQ, LA_BST_WORDS = 0, 16, 32, 288, 352, 416, 480


class LookaheadHipError(RuntimeError):
    pass


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7).  liblookahead_hip.so must
    share that ONE runtime instance with torch (streams and device pointers cross the boundary); if the system copy
    under /opt/rocm were loaded first the process would hold two HIP runtimes.  Loading torch's copy first (without
    importing torch) makes the dynamic loader resolve our NEEDED libamdhip64.so.7 to it."""
    import importlib.util
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return None
    libdir = os.path.join(os.path.dirname(spec.origin), 'lib')
    for name in ('libhsa-runtime64.so', 'libamdhip64.so'):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return None
    return libdir


def _load(path=LIB_PATH):
    _preload_torch_hip_runtime()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build it with "
            f"`bash {os.path.join(_HERE, 'csrc', 'build.sh')}` (hipcc --offload-arch=gfx950) or "
            f"`python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU fallback.")
    return C.CDLL(path)


lib = _load()

vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
pi32, pi64, pu64, pf32 = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_uint64), C.POINTER(C.c_float)


class LlamaConfigC(C.Structure):
    _fields_ = [("n_layers", i32), ("hidden", i32), ("n_heads", i32), ("n_kv_heads", i32), ("head_dim", i32),
                ("ffn", i32), ("vocab", i32), ("max_keys", i32), ("max_pos", i32), ("attn_split", i32),
                ("rms_eps", f32), ("gemm_cfg", i32 * 8), ("balanced_wg", i32 * 3), ("n_slots", i32),
                ("n_experts", i32), ("top_k", i32), ("fuse", i32), ("sliding_window", i32), ("kv_ring", i32), ("max_blocks", i32),
                ("norm_cast_first", i32), ("qkv_mb_wg", i32)]


class LlamaLayerWeightsC(C.Structure):
    _fields_ = [("wqkv", vp), ("wo", vp), ("wgateup", vp), ("wdown", vp), ("norm1", vp), ("norm2", vp),
                ("router", vp), ("ex_gateup", C.POINTER(vp)), ("ex_down", C.POINTER(vp)), ("wqkv_mb", vp)]


class LlamaWeightsC(C.Structure):
    _fields_ = [("embed", vp), ("lm_head", vp), ("final_norm", vp), ("rope_cos", vp), ("rope_sin", vp),
                ("layers", C.POINTER(LlamaLayerWeightsC))]


class TrieImageC(C.Structure):
    """la_trie_image: the device arrays of the trie mirror (device pointers as integers)"""
    _fields_ = [('tok', C.c_void_p), ('fo', C.c_void_p), ('fi', C.c_void_p), ('fi_stride', C.c_int64), ('n_planes', C.c_int32),
                ('cstart', C.c_void_p), ('ccount', C.c_void_p), ('ccap', C.c_void_p), ('meta', C.c_void_p), ('cap', C.c_int32),
                ('root_of', C.c_void_p), ('n_root_of', C.c_int32)]


class TrieQueryC(C.Structure):
    """la_trie_query: one launch of the workgroup-per-query retrieval (la_trie_hier_get_wg)"""
    _fields_ = [('tok', C.c_void_p), ('fo', C.c_void_p), ('fi', C.c_void_p), ('fi_stride', C.c_int64), ('cstart', C.c_void_p),
                ('ccount', C.c_void_p), ('n_records', C.c_int32), ('root_of', C.c_void_p), ('n_root_of', C.c_int32),
                ('queries', C.c_void_p), ('nq', C.c_void_p), ('plane', C.c_void_p), ('branch_lengths', C.c_void_p), ('B', C.c_int32),
                ('decoding_length', C.c_int32), ('branch_length', C.c_int32), ('min_in', C.c_int32), ('min_out', C.c_int32),
                ('mode', C.c_int32), ('stop', C.c_void_p), ('n_stop', C.c_int32), ('scratch_i', C.c_void_p), ('scratch_v', C.c_void_p),
                ('out_ids', C.c_void_p), ('out_rowmask', C.c_void_p), ('row_stride', C.c_int32), ('mask_words', C.c_int32),
                ('out_n', C.c_void_p), ('out_sizes', C.c_void_p), ('out_nsizes', C.c_void_p),
                ('lds_level_cap', C.c_int32), ('lds_cand_cap', C.c_int32), ('one_wave_cap', C.c_int32)]


LA_TRIE_OBUF = 128


class DecodeParamsC(C.Structure):
    _fields_ = [("decoding_length", i32), ("branch_length", i32), ("max_query_length", i32), ("mode", i32), ("idx", i32),
                ("max_length", i32), ("max_steps", i32), ("n_eos", i32), ("eos", i32 * 8)]


def _proto(name, restype, *argtypes, dll=None):
    fn = getattr(lib if dll is None else dll, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


# every symbol declared in include/lookahead_hip.h (tests/test_abi.py checks this list against the header)
PROTOTYPES = {
    "la_abi_version": (i32,),
    "la_abi_dtype": (i32,),
    "la_last_error": (C.c_char_p,),
    "la_debug_set": (i32, i32, i32),
    "la_debug_get": (i32, i32),
    "la_cache_create": (vp, i32, i32),
    "la_cache_destroy": (None, vp),
    "la_cache_set_limits": (i32, vp, i32, i32),
    "la_cache_set_eos": (i32, vp, pi32, i32),
    "la_cache_set_stop_words": (i32, vp, pi32, i32),
    "la_cache_fresh": (i32, vp),
    "la_cache_put": (i32, vp, pi32, i32, i32, i32, i32, i32),
    "la_cache_stream_put": (i32, vp, pi32, i32, i32, i32, i32),
    "la_cache_stream_put_many": (i32, vp, pi32, pi32, pi32, i32, i32, i32),
    "la_cache_hier_get": (i32, vp, pi32, i32, i32, i32, i32, i32, i32, i32, i32, pi32, pi32, pu64, pi64, pi32, pi32, pi32),
    "la_cache_one_get": (i32, vp, pi32, i32, i32, i32, i32, i32, i32, pi32, pi32, pi32, pi32),
    "la_cache_par_get": (i32, vp, pi32, i32, i32, i32, i32, i32, i32, i32, i32, pi32, pu64, pi64, pi32, pi32, pi32),
    "la_cache_bat_get_packed": (i32, vp, pi32, pi32, i32, i32, i32, i32, i32, pi32, i32, i32, pi32, pu64, pi32, pi32, pi32),
    "la_cache_reset_input_freqs": (i32, vp, i32),
    "la_cache_squeeze": (i32, vp),
    "la_cache_stats": (i32, vp, pi64, pi64, pi64, pi64),
    "la_cache_tree_counters": (i32, vp, i32, pi64, pi64),
    "la_cache_save": (i32, vp, C.c_char_p),
    "la_cache_load": (i32, vp, C.c_char_p),
    "la_cache_export": (i32, vp, i32, i32, pi32, C.POINTER(C.c_double), C.POINTER(C.c_double), pi32, pi32, pi32),
    "la_trie_hier_get_dev": (i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp, vp, vp, vp, vp, vp),
    "la_build_tree_inputs": (i32, vp, vp, vp, vp, vp, vp),
    "la_accept_scan": (i32, vp, vp, vp, vp),
    "la_kv_commit": (i32, vp, vp, vp, vp, vp, vp, i32, i32, i32),
    "la_pack_weight": (i32, vp, vp, vp, i32, i32, i32, vp),
    "la_pack_x": (i32, vp, vp, i32, vp),
    "la_gemm64_slab": (i32, vp, vp, vp, i32, i32, i32, i32, vp),
    "la_gemm64_swiglu": (i32, vp, vp, vp, i32, i32, vp, i32),
    "la_gemm64_qkv": (i32, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, i32),
    "la_qkv_row_perm": (i32, i32, i32, pi32),
    "la_head_lane_map": (i32, i32, pi32),
    "la_rowplan": (i32, i32, i32, i32, pi32),
    "la_planned_elems": (i64, i32, i32, i32, i32),
    "la_pack_planned": (i32, vp, vp, vp, vp, i32, i32, i32, i32, vp),
    "la_gemm64r_swiglu": (i32, vp, vp, vp, i32, i32, i32, vp),
    "la_gemm64r_logits": (i32, vp, vp, vp, i32, i32, i32, vp, vp, vp),
    "la_gemm64r_qkv": (i32, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp),
    "la_gemm64_logits": (i32, vp, vp, vp, i32, i32, i32, vp, vp, vp),
    "la_argmax_finalize": (i32, vp, vp, vp, i32, vp),
    "la_embed_norm": (i32, vp, vp, vp, vp, i32, f32, vp, vp),
    "la_resid_norm": (i32, vp, vp, vp, i32, vp, i32, f32, vp),
    "la_qkv_post": (i32, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp),
    "la_tree_attn": (i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp),
    "la_llama_workspace_bytes": (i64, C.POINTER(LlamaConfigC)),
    "la_llama_create": (vp, C.POINTER(LlamaConfigC), C.POINTER(LlamaWeightsC), vp, i64),
    "la_llama_destroy": (None, vp),
    "la_llama_reset": (i32, vp, vp),
    "la_llama_step": (i32, vp, vp, vp, vp),
    "la_llama_wait": (i32, vp, vp),
    "la_llama_step_eager": (i32, vp, vp, vp, vp),
    "la_llama_commit": (i32, vp, vp, pi32, i32, vp),
    "la_lookahead_decode": (i32, vp, vp, vp, C.POINTER(DecodeParamsC), pi32, pi32, vp, vp, pi32, pi32, pi32, pi32,
                            C.POINTER(C.c_double), C.POINTER(C.c_double)),
    "la_llama_buffer": (vp, vp, i32),
    "la_llama_profile": (i32, vp, vp, vp, i32, pf32, pi32),
    "la_llama_profile_gateup": (i32, vp, vp, i32, pf32),
    "la_resid_norm_router": (i32, vp, vp, vp, i32, vp, i32, f32, vp, vp, i32, i32, vp, vp),
    "la_moe_accum": (i32, vp, vp, i32, vp, i32, i32, vp, i32),
    "la_resid_norm_addend": (i32, vp, vp, vp, vp, i32, f32, vp),
    "la_build_batch_inputs": (i32, vp, vp, vp, vp, vp, vp),
    "la_accept_scan_batch": (i32, vp, vp, vp, vp, vp, i32, i32),
    "la_kv_commit_batch": (i32, vp, vp, vp, vp, vp, vp, i32, i32, i32),
    "la_tree_attn_batch": (i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp),
    "la_llama_bstep": (i32, vp, vp, vp, vp),
    "la_llama_mstep": (i32, vp, vp, vp, vp),
    "la_llama_mstep_eager": (i32, vp, vp, vp, vp),
    "la_llama_mstep_trie": (i32, vp, vp, i32, pi32, pi32, pi32, vp, vp, vp, vp),
    "la_llama_set_nkeys": (i32, vp, vp, i32, i32),
    "la_llama_bcommit": (i32, vp, vp, pi32, vp),
    "la_llama_mcommit": (i32, vp, vp, i32, pi32, vp),
    "la_cache_mirror_enable": (i32, vp, pi32, i32),
    "la_cache_mirror_state": (i32, vp, pi32, pi32, pi32, pi32),
    "la_cache_mirror_image": (i32, vp, i32, pi32, C.POINTER(C.c_double), C.POINTER(C.c_double), pi32, pi32),
    "la_cache_mirror_patch": (i32, vp, pi32, pi32, C.POINTER(C.c_double)),
    "la_trie_patch_dev": (i32, vp, vp, vp, vp, i64, vp, vp, vp, vp, i32, vp, vp, i32),
    "la_trie_one_get_dev2": (i32, vp, vp, vp, vp, i64, vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, vp, i32, vp, vp, vp, vp, vp),
    "la_cache_mirror_ccap": (i32, vp, i32, pi32),
    "la_cache_mirror_discard": (i32, vp, pi32),
    "la_cache_stream_buffer": (i32, vp, i32, i32, pi32, pi32),
    "la_trie_root_index_dev": (i32, vp, C.POINTER(TrieImageC), i32),
    "la_trie_stream_put_dev": (i32, vp, C.POINTER(TrieImageC), vp, vp, vp, i32, vp, vp, i32, i32, vp, i32, vp, i32, vp),
    "la_trie_hier_get_wg": (i32, vp, C.POINTER(TrieQueryC)),
    "la_trie_hier_get_dev2": (i32, vp, vp, vp, vp, i64, vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp, vp, vp, vp, vp, vp),
    "la_comm_unique_id": (i32, vp),
    "la_comm_create": (vp, vp, i32, i32),
    "la_comm_destroy": (i32, vp),
    "la_gather_accepted": (i32, vp, vp, vp, i32, i32, vp),
    "la_mb_gemm": (i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32),
    "la_llama_bstep_eager": (i32, vp, vp, vp, vp),
    "la_llama_reset_slot": (i32, vp, vp, i32),
}

# the kernel lab (include/lookahead_hip_lab.h): measurement knobs / A/B switches for scripts/ and the bitwise-identity tests — not part
# of the product boundary
LAB_PROTOTYPES = {
    "la_lab_set": (i32, i32, i32),
    "la_lab_get": (i32, i32),
    "la_lab_set_ptr": (i32, i32, vp),
}

def _bind(dll, path, want_dtype, lab=False):
    for _n, _sig in list(PROTOTYPES.items()) + (list(LAB_PROTOTYPES.items()) if lab else []):
        _proto(_n, _sig[0], *_sig[1:], dll=dll)
    if dll.la_abi_version() != ABI_VERSION or dll.la_abi_dtype() != want_dtype:
        raise ImportError(f"{path} implements ABI {dll.la_abi_version()} / dtype {dll.la_abi_dtype()}, the bindings expect {ABI_VERSION} / "
                          f"{want_dtype}: rebuild it with `bash {os.path.join(_HERE, 'csrc', 'build.sh')}`")
    return dll


_bind(lib, LIB_PATH, LA_DTYPE_BF16, lab=LAB_BUILD)
_lib_f16 = None
_lab_libs = {}       # dtype name -> lab build loaded beside the product libraries (LAB_BUILD: the process libraries themselves)
_knobs = {}          # (entry point, key) -> value: every knob set through lab_set / debug_set, replayed into a library loaded later


def loaded_libs():
    """every liblookahead_hip build this process has loaded (bf16 always; fp16 once an fp16 engine exists; lab builds once asked for)"""
    out = [d for d in (lib, _lib_f16) if d is not None]
    return out + [d for d in _lab_libs.values() if d not in out]


def loaded_lab_libs():
    return [d for d in loaded_libs() if getattr(d, '_is_lab', False)]


def _replay(dll, path, lab):
    """knobs set before this build was loaded must not be dropped silently by it"""
    for (fn, key), value in _knobs.items():
        if fn.startswith('la_lab') and not lab:
            continue
        rc = getattr(dll, fn)(key, value)
        if rc != LA_OK:
            raise LookaheadHipError(f'{path}: {fn}({key}, {value}) replayed from the builds loaded earlier was refused (status {rc})')


def _set_everywhere(fn, getter, key, value, libs):
    """One knob on EVERY build of `libs`, all or nothing: a build that refuses the value must not leave the builds before it on the new value
    (bf16 and fp16 engines of one process would then run different kernels).  The libraries already set are rolled back to the value they
    reported before (la_lab_get; la_debug_set has no getter: its only key is replayed from the recorded value, default 0)."""
    key, value = int(key), int(value)
    done = []
    for d in libs:
        before = getattr(d, getter)(key) if getter else _knobs.get((fn, key), 0)
        r = getattr(d, fn)(key, value)
        if r != LA_OK:
            for dd, old in done:
                getattr(dd, fn)(key, int(old))
            return r
        done.append((d, before))
    _knobs[(fn, key)] = value
    return LA_OK


def lab_lib_for(dtype):
    """The LAB build for a torch dtype (liblookahead_hip_lab.so / _lab_f16.so: the same sources with -DLA_LAB=1), loaded on first use beside
    the product library — its own knob storage, graph epoch and kernels; engines created with lab=True run on it."""
    name = str(dtype).replace('torch.', '')
    if LAB_BUILD:
        return lib_for(dtype)
    if name not in ('bfloat16', 'float16'):
        raise ValueError(f'no liblookahead_hip build for dtype {dtype}: bfloat16 and float16 exist')
    if name not in _lab_libs:
        path = LAB_PATH if name == 'bfloat16' else LAB_PATH_F16
        if not os.path.exists(path):
            raise ImportError(f'{path} is missing: build it with `bash {os.path.join(_HERE, "csrc", "build.sh")}` (lab builds are skipped under LA_SKIP_LAB=1)')
        dll = _bind(_load(path), path, LA_DTYPE_BF16 if name == 'bfloat16' else LA_DTYPE_F16, lab=True)
        dll._is_lab = True
        _replay(dll, path, True)
        _lab_libs[name] = dll
    return _lab_libs[name]


def lab_set(key, value):
    """la_lab_set on EVERY loaded lab build (each .so has its own knob globals and graph epoch) and on lab builds loaded later — a knob
    set before an fp16 lab engine exists must not be silently ignored by it.  The bf16 lab build is loaded if none is yet.  All or nothing:
    -> the status of the library that refused (the others are rolled back).  The PRODUCT libraries have no knobs: an engine created without
    lab=True (and outside LA_LAB_BUILD=1) is not affected — by construction."""
    lab_lib_for('bfloat16')
    return _set_everywhere('la_lab_set', 'la_lab_get', key, value, loaded_lab_libs())


def debug_set(key, value):
    """la_debug_set (the product header's depth probe) on every loaded build, replayed like lab_set"""
    return _set_everywhere('la_debug_set', None, key, value, loaded_libs())


def lab_get(key, dtype=None):
    """la_lab_get of the lab build serving `dtype` (default: bf16)"""
    return int(lab_lib_for('bfloat16' if dtype is None else dtype).la_lab_get(int(key)))


def lib_for(dtype):
    """The library instantiated for a torch dtype: bfloat16 -> liblookahead_hip.so, float16 -> liblookahead_hip_f16.so (loaded on
    first use; same ABI, same sources, v_mfma_f32_32x32x16_f16 and fp16 rounding points).  Anything else is refused: the engine
    computes in the checkpoint's 16-bit type, never in a silently converted one."""
    global _lib_f16
    name = str(dtype).replace('torch.', '')
    if name == 'bfloat16':
        return lib
    if name == 'float16':
        if _lib_f16 is None:
            dll = _bind(_load(LIB_PATH_F16), LIB_PATH_F16, LA_DTYPE_F16, lab=LAB_BUILD)
            dll._is_lab = LAB_BUILD
            _replay(dll, LIB_PATH_F16, LAB_BUILD)
            _lib_f16 = dll
        return _lib_f16
    raise ValueError(f'no liblookahead_hip build for dtype {dtype}: bfloat16 and float16 exist')


lib._is_lab = LAB_BUILD


def last_error() -> str:
    """Text of the last error on this thread (each loaded library keeps its own; the non-empty one is reported)."""
    out = []
    for dll in loaded_libs():
        if dll is not None:
            s = dll.la_last_error()
            if s:
                out.append(s.decode("utf-8", "replace"))
    return " | ".join(out)


def check(rc: int, what: str = ""):
    """Raise on a negative LA_E_* status.  LA_E_ARG maps to AssertionError, mirroring the reference's
    `assert` on bad arguments (lookahead_cache.py:34,67,377,521-524)."""
    if rc == LA_OK:
        return
    msg = f"{what or 'liblookahead_hip'}: status {rc}: {last_error()}"
    if rc == -1:
        raise AssertionError(msg)
    raise LookaheadHipError(msg)
