// Per-wave phase clock of score_kernel: builds the product source with its lap points enabled.
#define RBG_SCORE_TRACE 1
#include "../../recbole-gnn_amd/csrc/score.hip"

extern "C" int mb_score_trace(const float *U, const float *I, float *S, int64_t B, int64_t n, int d, unsigned long long *trace, void *stream) {
    (void)hipMemcpyToSymbol(HIP_SYMBOL(rbg::g_score_trace), &trace, sizeof(trace));
    return rbg_score_f32(U, d, I, d, S, B, n, d, stream);
}

extern "C" int mb_score_debug_set(int bits) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(rbg::g_score_debug), &bits, sizeof(bits)); }
