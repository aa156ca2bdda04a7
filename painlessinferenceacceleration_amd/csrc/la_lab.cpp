// la_lab.cpp — the kernel lab's entry points (include/lookahead_hip_lab.h): measurement knobs and A/B switches.  Compiled ONLY into the lab
// build (-DLA_LAB=1: liblookahead_hip_lab.so / _lab_f16.so); the product libraries hold the defaults as constexprs (la_knobs.h) and export
// none of this.  The knob storage lives here, generated from the table.
#if !LA_LAB
#error "la_lab.cpp belongs to the lab build (build.sh compiles it with -DLA_LAB=1 only)"
#endif
#include <hip/hip_runtime.h>
#include "../../include/lookahead_hip.h"
#include "../../include/lookahead_hip_lab.h"
#include "la_knobs.h"

#define LA_KNOB_DEF(name, dflt, key, lo, hi) int name = dflt;
LA_KNOB_TABLE(LA_KNOB_DEF)
int g_la_fork_pf[5] = {0, 0, 0, 0, 0};
long long* g_la_dbg_times = nullptr;

extern "C" {
// Every knob is read when a step graph is CAPTURED (kernel arguments / launch shapes are baked in): each change bumps the
// capture epoch, and la_llama_step / la_llama_bstep / la_llama_mstep re-capture a graph whose epoch is stale.
int la_lab_set(int key, int value) {
    if (key == 13 && value >= 0) { g_la_stop_layers = value; ++g_la_graph_epoch; return LA_OK; }      // the product header's depth probe, also here
    if (key >= 26 && key <= 30 && value >= 0 && value <= 128) { g_la_fork_pf[key - 26] = value; ++g_la_graph_epoch; return LA_OK; }
    if ((key == 22 && value == 3) || (key == 33 && (value == 1 || value == 3))) return LA_E_ARG;
#define LA_KNOB_SET(name, dflt, k, lo, hi) if (key == k) { if (value < lo || value > hi) return LA_E_ARG; name = value; ++g_la_graph_epoch; return LA_OK; }
    LA_KNOB_TABLE(LA_KNOB_SET)
    return LA_E_ARG;
}
int la_lab_get(int key) {
    if (key == 13) return g_la_stop_layers;
    if (key >= 26 && key <= 30) return g_la_fork_pf[key - 26];
#define LA_KNOB_GET(name, dflt, k, lo, hi) if (key == k) return name;
    LA_KNOB_TABLE(LA_KNOB_GET)
    return LA_E_ARG;
}
int la_lab_set_ptr(int key, void* d_ptr) {
    if (key == 0) { g_la_dbg_times = (long long*)d_ptr; return LA_OK; }
    return LA_E_ARG;
}
}
