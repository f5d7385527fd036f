// Per-wave clock trace of bignn_dense_pipe_kernel: builds the product source with its stamps enabled.
#define RBG_BIGNN_TRACE 1
#include "../../recbole-gnn_amd/csrc/bignn.hip"

extern "C" int mb_bignn_trace(const float *P, const float *X, const float *W1, const float *b1, const float *W2, const float *b2,
                              float *Y, int64_t n_rows, int leaky, unsigned long long *trace, void *stream) {
    hipMemcpyToSymbol(HIP_SYMBOL(rbg::g_bignn_trace), &trace, sizeof(trace));
    rbg::BignnParams p{};
    p.P = P, p.X = X, p.ldx = 64, p.W1 = W1, p.b1 = b1, p.W2 = W2, p.b2 = b2, p.Y = Y, p.ldy = 64, p.n_rows = n_rows;
    p.d_in = 64, p.d_out = 64, p.leaky_norm = leaky, p.slope = 0.2f;
    return rbg::launch_dense_pipe<4>(p, (hipStream_t)stream);
}
