// Per-wave clock of the column-slab kernels (sell_spmm_kernel, sell_stream_kernel): the product source with its lap points on.
// Built into a full library next to the product's other objects: devtools/microbench/build_sell_trace.sh -> librbgnn_selltrace.so,
// loaded through RBGNN_LIB (devtools/r05_trace.py).
#define RBG_SELL_TRACE 1
#include "../../recbole-gnn_amd/csrc/sell.hip"

extern "C" int mb_sell_trace_set(unsigned long long *trace) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(rbg::g_sell_trace), &trace, sizeof(trace));
}

extern "C" int mb_sell_debug_set(int bits) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(rbg::g_sell_debug), &bits, sizeof(bits)); }
