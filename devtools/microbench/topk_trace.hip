// Per-wave phase clock of score_topk_kernel: builds the product source with its lap points enabled.
#define RBG_TOPK_TRACE 1
#ifdef NOPASS
#define RBG_TOPK_TRACE_NOPASS 1
#endif
#include "../../recbole-gnn_amd/csrc/topk.hip"

extern "C" int mb_topk_trace_set(unsigned long long *trace) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(rbg::g_topk_trace), &trace, sizeof(trace));
}

extern "C" int mb_topk_debug_set(int bits) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(rbg::g_topk_debug), &bits, sizeof(bits)); }
