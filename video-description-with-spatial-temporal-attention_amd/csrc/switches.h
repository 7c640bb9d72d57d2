// Runtime switches of libstattn: ONE list.
//
// A product build (the Makefile's default) reads exactly the environment variables named in STATTN_PRODUCT_SWITCHES below -- each of
// them selects between two paths that are BOTH covered by the GPU parity suite, or is a documented fallback.  Everything else the
// sources consult (tile forcing, ablations, ring depths, A/B of a launch rule ...) is a TOOL switch: sw_tool() reads the environment
// only in a tools build (-DSTATTN_TOOL_SWITCHES: tools/build_tools_lib.sh -> tools/_var/libstattn_tools.so, same kernels, host code that
// listens; also implied by -DSTATTN_PROBES) and is the constant nullptr in the product library, so a stray variable can not move the
// product off its measured paths.  (the library read 38 variables.)
#pragma once
#include <cstdlib>
#include <cstring>

namespace stattn {

// name                       effect when set                                                             covered by
#define STATTN_PRODUCT_SWITCHES(X)                                                                                                     \
    X(STATTN_NO_RIDER)         /* no GEMM rider workgroups in the attention launches (fwd + bwd)           tests/test_gpu_bf16.py (child process), bench legs */ \
    X(STATTN_NO_UPDATE_RIDER)  /* beam update in its own launch instead of riding the next word's attention tests/test_gpu_parity.py */ \
    X(STATTN_NO_ROW_WG)        /* one update workgroup per video instead of one per hypothesis             tests/test_gpu_parity.py */ \
    X(STATTN_NO_PANELS)        /* recurrent GEMMs on the LDS-tiled kernels instead of the row-panel ones    tests/test_gpu_parity.py */ \
    X(STATTN_GEMM_NOGROUP)     /* one launch per GEMM problem instead of grouped launches                  tests/test_gpu_z3_switches.py, bench.py accounting */ \
    X(STATTN_READOUT_NOPAIR)   /* readout as two GEMMs instead of one K-concatenated launch                tests/test_gpu_z3_switches.py */ \
    X(STATTN_WIDE_STATS_FROM)  /* first beam width whose vocabulary launch uses the wide statistics path   tests/test_gpu_z3_switches.py (65 = stored logits) */ \
    X(STATTN_BEAM_NOGRAPH)     /* word loop of stattn_beam_search without hipGraph capture                 tests/test_gpu_z3_switches.py */ \
    X(STATTN_COMM_NO_OVERLAP)  /* one all-reduce after the backward pass instead of five overlapped ones   fallback, tests/test_gpu_dp2.py */ \
    X(STATTN_DBG_REDZONE)      /* canary zones around every device buffer, checked after every API call    tests/test_gpu_z1_redzone.py */

inline bool sw_is_product(const char* name) {
#define X(n) if (!strcmp(name, #n)) return true;
    STATTN_PRODUCT_SWITCHES(X)
#undef X
    return false;
}
inline int sw_product_count() {
    int n = 0;
#define X(n_) ++n;
    STATTN_PRODUCT_SWITCHES(X)
#undef X
    return n;
}
inline const char* sw_product_name(int i) {
    int n = 0;
#define X(n_) if (n++ == i) return #n_;
    STATTN_PRODUCT_SWITCHES(X)
#undef X
    return nullptr;
}

// a switch of the product library (must be on the list: a typo is a null switch in every build, caught by tests/test_abi_and_host.py)
inline const char* sw_product(const char* name) { return sw_is_product(name) ? getenv(name) : nullptr; }

// a tool switch: the environment in tools / probe builds, nothing in the product
#if defined(STATTN_TOOL_SWITCHES) || defined(STATTN_PROBES)
inline const char* sw_tool(const char* name) { return getenv(name); }
#else
inline const char* sw_tool(const char*) { return nullptr; }
#endif

}  // namespace stattn
