/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
 *
 * Plain-C restatement of the operation ORDER of the two dense torch ops on the reference's hot path
 * (/root/reference/ultra/layers.py:233-240: nn.Linear, nn.LayerNorm), i.e. of what torch's CPU kernels compute for
 * the shapes ULTRA uses.  The reference delegates these to torch (third-party, absent from /root/reference:
 * torch 2.10.0 here, ATen native/cpu/moments_utils.h + layer_norm_kernel.cpp and MKL sgemm behind addmm); the order
 * below was established empirically -- tests/test_torch_math.py checks these functions BIT FOR BIT against
 * torch.nn.functional.linear / layer_norm on random inputs, on whatever host runs the tests -- and the GPU kernels
 * (ultra_amd/csrc/torch_math.hpp and the MFMA feeds) follow the same order.
 *
 *   linear:     y[m][f] = fl( chain + b[f] ),  chain = fmaf(x[K-1], W[f][K-1], ... fmaf(x[0], W[f][0], 0) ...)
 *   layer norm: 8 Welford accumulators (accumulator i takes features i, 8 + i, ...), merged sequentially (Chan),
 *               var = m2 / N, rstd = 1 / sqrt(var + eps), y = fmaf((x - mean) * rstd, gamma, beta)       (N % 8 == 0)
 */
#include <math.h>

void oracle_linear_seq_f32(const float *x, const float *w, const float *b, float *out, long M, long K, long N) {
#pragma omp parallel for
    for (long m = 0; m < M; ++m)
        for (long n = 0; n < N; ++n) {
            float acc = 0.f;
            for (long k = 0; k < K; ++k) acc = fmaf(x[m * K + k], w[n * K + k], acc);
            out[m * N + n] = b ? acc + b[n] : acc;
        }
}

void oracle_layer_norm_f32(const float *X, const float *gamma, const float *beta, float *out, long M, int N, float eps) {
    const int V = 8, n = N / V;
#pragma omp parallel for
    for (long r = 0; r < M; ++r) {
        const float *x = X + r * N;
        float m1v[8], m2v[8];
        for (int i = 0; i < V; ++i) m1v[i] = m2v[i] = 0.f;
        for (int j = 0; j < n; ++j) {
            const float c = 1.0f / (float)(j + 1);
            for (int i = 0; i < V; ++i) {
                const float xv = x[j * V + i];
                const float delta = xv - m1v[i];
                m1v[i] = fmaf(delta, c, m1v[i]);
                m2v[i] = fmaf(delta, xv - m1v[i], m2v[i]);
            }
        }
        long m0 = 0;
        float m1 = 0.f, m2 = 0.f;
        for (int i = 0; i < V; ++i) {
            const long nn = m0 + n;
            const float c = (float)n / (float)nn;
            const float delta = m1v[i] - m1;
            m1 = fmaf(c, delta, m1);
            m2 = m2 + fmaf(delta * delta * c, (float)m0, m2v[i]);
            m0 = nn;
        }
        const float rstd = 1.0f / sqrtf(m2 / (float)N + eps);
        for (int k = 0; k < N; ++k) {
            const float g = gamma ? gamma[k] : 1.f, bt = beta ? beta[k] : 0.f;
            out[r * N + k] = fmaf((x[k] - m1) * rstd, g, bt);
        }
    }
}
