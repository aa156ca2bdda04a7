// la_lab.cpp — the kernel lab's entry points (include/lookahead_hip_lab.h): measurement knobs and A/B switches.  Not declared by
// the product header; the knobs themselves live next to the kernels that read them.
#include <hip/hip_runtime.h>
#include "../../include/lookahead_hip.h"
#include "../../include/lookahead_hip_lab.h"

extern int g_la_dbg_noepi;
extern int g_la_kskew;
extern int g_la_prio_hi;
extern int g_la_mb_narrow;
extern int g_la_mb_dbg;
extern int g_la_mb_mode;
extern int g_la_mb_sch;
extern int g_la_ex_d4;
extern int g_la_mb_pair;
extern int g_la_mb_ks2;
extern int g_la_pf_kib, g_la_pf_delay, g_la_pf_tail_kib, g_la_attn_staged, g_la_graph_epoch, g_la_graph_reps, g_la_stop_layers, g_la_split_head_tail, g_la_gemm_4w, g_la_ex_split, g_la_attn_one, g_la_attn1_var, g_la_norm4, g_la_mb_attn_vring, g_la_mb_attn_rot, g_la_ex_down_ks, g_la_slab_wt;
extern long long* g_la_dbg_times;
extern int g_la_fork_pf[5], g_la_attn_ride_kib, g_la_attn_ride_delay, g_la_attn_merge_ns, g_la_oproj_probe;

extern "C" {
// Every knob is read when a step graph is CAPTURED (kernel arguments / launch shapes are baked in): each change bumps the
// capture epoch, and la_llama_step / la_llama_bstep / la_llama_mstep re-capture a graph whose epoch is stale.
int la_lab_set(int key, int value) {
    if (key == 0) { g_la_dbg_noepi = value; ++g_la_graph_epoch; return LA_OK; }
    if (key == 1 && value >= 0 && value <= 64) { g_la_kskew = value; ++g_la_graph_epoch; return LA_OK; }
    if (key == 2 && value >= 0 && value <= 3) { g_la_prio_hi = value; ++g_la_graph_epoch; return LA_OK; }
    if (key == 3 && value >= 0 && value <= 1) { g_la_mb_narrow = value; ++g_la_graph_epoch; return LA_OK; }
    if (key == 4 && value >= 0 && value <= 6) { g_la_mb_dbg = value; ++g_la_graph_epoch; return LA_OK; }
    if (key == 5 && value >= 0 && value <= 3) { g_la_mb_mode = value; ++g_la_graph_epoch; return LA_OK; }
    if (key == 6 && value >= 0 && value <= 32767) { g_la_mb_pair = value; ++g_la_graph_epoch; return LA_OK; }
    if (key == 7 && value >= 0 && value <= 128) { g_la_pf_kib = value; ++g_la_graph_epoch; return LA_OK; }
    if (key == 8 && value >= 0 && value <= 16) { g_la_pf_delay = value; ++g_la_graph_epoch; return LA_OK; }
    if (key == 9 && value >= 0 && value <= 64) { g_la_pf_tail_kib = value; ++g_la_graph_epoch; return LA_OK; }
    if (key == 10 && value >= 0 && value <= 1) { g_la_attn_staged = value; ++g_la_graph_epoch; return LA_OK; }
    if (key == 12 && value >= 0 && value <= 1) { g_la_mb_ks2 = value; ++g_la_graph_epoch; return LA_OK; }
    if (key == 11 && value >= 1 && value <= 8) { g_la_graph_reps = value; ++g_la_graph_epoch; return LA_OK; }
    if (key == 13 && value >= 0) { g_la_stop_layers = value; ++g_la_graph_epoch; return LA_OK; }
    if (key == 14 && value >= 0 && value <= 1) { g_la_split_head_tail = value; ++g_la_graph_epoch; return LA_OK; }
    if (key == 15 && value >= 0 && value <= 7) { g_la_gemm_4w = value; ++g_la_graph_epoch; return LA_OK; }
    if (key == 16 && value >= 0 && value <= 7) { g_la_ex_split = value; ++g_la_graph_epoch; return LA_OK; }
    if (key == 17 && value >= 0 && value <= 1) { g_la_attn_one = value; ++g_la_graph_epoch; return LA_OK; }
    if (key == 24 && value >= 0 && value <= 1) { g_la_mb_sch = value; ++g_la_graph_epoch; return LA_OK; }
    if (key == 25 && value >= 0 && value <= 15) { g_la_ex_d4 = value; ++g_la_graph_epoch; return LA_OK; }
    if (key == 18 && value >= 0 && value <= 7) { g_la_attn1_var = value; ++g_la_graph_epoch; return LA_OK; }
    if (key == 19 && value >= 0 && value <= 1) { g_la_norm4 = value; ++g_la_graph_epoch; return LA_OK; }
    if (key == 20 && value >= 0 && value <= 1) { g_la_mb_attn_vring = value; ++g_la_graph_epoch; return LA_OK; }
    if (key == 21 && value >= 0 && value <= 1) { g_la_mb_attn_rot = value; ++g_la_graph_epoch; return LA_OK; }
    if (key == 23 && value >= 0 && value <= 1) { g_la_slab_wt = value; ++g_la_graph_epoch; return LA_OK; }
    if (key == 34 && value >= 0 && value <= 63) { g_la_oproj_probe = value; ++g_la_graph_epoch; return LA_OK; }
    if (key == 33 && (value == 0 || value == 2 || value == 4)) { g_la_attn_merge_ns = value; ++g_la_graph_epoch; return LA_OK; }
    if (key == 32 && value >= 0 && value <= 16) { g_la_attn_ride_delay = value; ++g_la_graph_epoch; return LA_OK; }
    if (key == 31 && value >= 0 && value <= 128) { g_la_attn_ride_kib = value; ++g_la_graph_epoch; return LA_OK; }
    if (key >= 26 && key <= 30 && value >= 0 && value <= 128) { g_la_fork_pf[key - 26] = value; ++g_la_graph_epoch; return LA_OK; }
    if (key == 22 && (value == 0 || value == 1 || value == 2 || value == 4)) { g_la_ex_down_ks = value; ++g_la_graph_epoch; return LA_OK; }
    return LA_E_ARG;
}
int la_lab_get(int key) {
    switch (key) {
        case 0: return g_la_dbg_noepi; case 1: return g_la_kskew; case 2: return g_la_prio_hi; case 3: return g_la_mb_narrow;
        case 4: return g_la_mb_dbg; case 5: return g_la_mb_mode; case 6: return g_la_mb_pair; case 7: return g_la_pf_kib; case 8: return g_la_pf_delay; case 9: return g_la_pf_tail_kib; case 10: return g_la_attn_staged; case 11: return g_la_graph_reps; case 12: return g_la_mb_ks2; case 13: return g_la_stop_layers; case 14: return g_la_split_head_tail; case 15: return g_la_gemm_4w; case 16: return g_la_ex_split; case 17: return g_la_attn_one; case 18: return g_la_attn1_var; case 19: return g_la_norm4; case 20: return g_la_mb_attn_vring; case 21: return g_la_mb_attn_rot; case 22: return g_la_ex_down_ks; case 23: return g_la_slab_wt; case 24: return g_la_mb_sch; case 25: return g_la_ex_d4;
        case 31: return g_la_attn_ride_kib; case 32: return g_la_attn_ride_delay; case 33: return g_la_attn_merge_ns; case 34: return g_la_oproj_probe;
        case 26: case 27: case 28: case 29: case 30: return g_la_fork_pf[key - 26];
        default: return LA_E_ARG;
    }
}
int la_lab_set_ptr(int key, void* d_ptr) {
    if (key == 0) { g_la_dbg_times = (long long*)d_ptr; return LA_OK; }
    return LA_E_ARG;
}
}
