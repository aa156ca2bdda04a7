// la_knobs.h — every A/B switch and measurement knob of the kernel lab, ONE table.
//   product build (liblookahead_hip.so, liblookahead_hip_f16.so): each knob is a named `constexpr` holding the library default — the variant
//     code behind a non-default value is dead, its kernels are not instantiated, and the library exports no la_lab_* symbol;
//   lab build (-DLA_LAB=1: liblookahead_hip_lab.so, liblookahead_hip_lab_f16.so, the same sources + la_lab.cpp): each knob is an int that
//     la_lab_set(key, value) changes (include/lookahead_hip_lab.h documents the keys); the A/B scripts under scripts/ and the
//     "variant == default" tests load this build.
// X(name, default, la_lab key, lowest, highest accepted value)
#pragma once
#define LA_KNOB_TABLE(X) \
    X(g_la_dbg_noepi,        0,     0, -2147483647, 2147483647) /* GEMM kernels return before the cross-wave reduction and epilogue (timing probe) */ \
    X(g_la_kskew,            0,     1, 0, 64)      /* K share (1/64ths) of waves 0..3 in the 8-wave GEMMs */ \
    X(g_la_prio_hi,          0,     2, 0, 3)       /* s_setprio level of waves 4..7 in the 8-wave GEMMs */ \
    X(g_la_mb_narrow,        0,     3, 0, 1)       /* multi-block GEMMs always on the K-split kernels */ \
    X(g_la_mb_dbg,           0,     4, 0, 6)       /* k_gemm_wide measurement builds */ \
    X(g_la_mb_mode,          0,     5, 0, 3)       /* unused */ \
    X(g_la_mb_pair,      12657,     6, 0, 32767)   /* forms of the wide multi-block launches: 1 | 16 | 32 | 64 | 256 | 4096 | 8192 (lookahead_hip_lab.h key 6) */ \
    X(g_la_pf_kib,           0,     7, 0, 128)     /* idle-window weight prefetch, KiB per consumer workgroup */ \
    X(g_la_pf_delay,         0,     8, 0, 16)      /* its start delay in s_sleep(32) rounds */ \
    X(g_la_pf_tail_kib,      0,     9, 0, 64)      /* tail prefetch of down_proj from the gate/up launch */ \
    X(g_la_attn_staged,      0,    10, 0, 1)       /* tree attention with K/V staged through LDS once per workgroup */ \
    X(g_la_graph_reps,       1,    11, 1, 8)       /* repetitions of the step inside the single-sequence graph */ \
    X(g_la_mb_ks2,           0,    12, 0, 1)       /* slab GEMMs of the multi-block step with 2 K splits at >= 5 blocks */ \
    X(g_la_split_head_tail,  0,    14, 0, 1)       /* separate build-inputs / embed / argmax / accept / publish kernels */ \
    X(g_la_gemm_4w,          0,    15, 0, 7)       /* bit 0: gate/up as 4 waves x 8 tile-sets */ \
    X(g_la_ex_split,         0,    16, 0, 7)       /* gathered MoE: 1 = one launch per expert and stage, 4 = plan and gather as two launches */ \
    X(g_la_attn_one,         1,    17, 0, 1)       /* single-launch tree attention on the single-sequence step (0 = key splits + combine) */ \
    X(g_la_attn1_var,        0,    18, 0, 7)       /* variants of the single-launch attention */ \
    X(g_la_norm4,            0,    19, 0, 1)       /* residual + RMSNorm with four workgroups per row */ \
    X(g_la_mb_attn_vring,    0,    20, 0, 1)       /* multi-block attention with the next tile's V in flight through an LDS ring */ \
    X(g_la_mb_attn_rot,      0,    21, 0, 1)       /* GQA: query heads of a kv head start their key-tile lists at different offsets */ \
    X(g_la_ex_down_ks,       0,    22, 0, 4)       /* K splits of the gathered experts' down projection (0 = the engine's choice; 3 refused) */ \
    X(g_la_slab_wt,          0,    23, 0, 1)       /* split-K slabs of the 64-row o_proj / down_proj stored write-through */ \
    X(g_la_mb_sch,           1,    24, 0, 1)       /* 1 = round-4 schedule of the wide GEMMs, 0 = the round-2 schedule */ \
    X(g_la_ex_d4,           13,    25, 0, 15)      /* merged-expert launches: two workgroups per CU / two weight regions per workgroup */ \
    X(g_la_attn_ride_kib,    0,    31, 0, 128)     /* KiB per o_proj workgroup pulled into L2 by rider workgroups of the attention launch */ \
    X(g_la_attn_ride_delay,  0,    32, 0, 16)      /* their start delay */ \
    X(g_la_attn_merge_ns,    0,    33, 0, 4)       /* 2 | 4 = key-split attention merged on load by o_proj (k_oproj_merge); 1, 3 refused */ \
    X(g_la_oproj_probe,      0,    34, 0, 63)      /* TIMING PROBE of a full-K o_proj with the norm folded away (results are garbage) */ \
    X(g_la_fatd,             1,    35, 0, 1)       /* paired gate/up launch at 5-8 blocks with the weights streamed into MFMA operand registers (k_gemm_fatd; 0 = k_gemm_fat) */ \
    X(g_la_fatx,             1,    36, 0, 1)       /* slab launches at 5-8 blocks (two token tiles per wave) with the x fragments streamed into MFMA operand registers (k_gemm_fat, STG = 2) */

#if LA_LAB
#define LA_KNOB_DECL(name, dflt, key, lo, hi) extern int name;
LA_KNOB_TABLE(LA_KNOB_DECL)
extern int g_la_fork_pf[5];            // la_lab_set keys 26..30: forked weight-prefetch branch of the single-sequence graph (KiB per stage)
extern long long* g_la_dbg_times;      // la_lab_set_ptr key 0: device buffer the GEMM / trie kernels stamp with wall_clock64()
#else
#define LA_KNOB_DECL(name, dflt, key, lo, hi) static constexpr int name = dflt;
LA_KNOB_TABLE(LA_KNOB_DECL)
static constexpr int g_la_fork_pf[5] = {0, 0, 0, 0, 0};
static constexpr long long* g_la_dbg_times = nullptr;
#endif
#undef LA_KNOB_DECL
// not knobs: the capture epoch of the step graphs, and the depth probe of the parity tests (la_debug_set key 13 of the PRODUCT header)
extern int g_la_graph_epoch;
extern int g_la_stop_layers;
