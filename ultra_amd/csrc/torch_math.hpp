// The reference's dense arithmetic, restated operation for operation.
//
// ULTRA's layers are torch ops on the reference's CPU path (layers.py:233-240): nn.Linear and nn.LayerNorm.  Their
// results are deterministic functions of the operation ORDER, and scores only reproduce the reference's bit patterns
// -- hence its rankings at near-ties -- if the GPU follows the same order:
//
//   nn.Linear (fp32, K <= a few hundred; torch 2.x CPU = MKL sgemm):  y[f] = fl( chain + bias[f] ),
//       chain = fma(x[K-1], W[f][K-1], ... fma(x[1], W[f][1], fma(x[0], W[f][0], 0)) ...)   -- one accumulator per
//       output, k ascending, fused multiply-add, bias added last.  v_mfma_f32_*_f32 is exactly such a k-ordered fmaf
//       chain, so the matrix kernels only have to feed k in ascending order.
//   nn.LayerNorm over 64 features (ATen RowwiseMoments, 8-float vectors): eight Welford accumulators, accumulator i
//       taking features i, 8 + i, ..., 56 + i; the eight are merged sequentially (Chan's update); var = m2 / 64;
//       y = fma((x - mean) * rstd, gamma, beta), rstd = 1 / sqrt(var + eps).
//
// Both restatements are pinned against torch itself on the CPU (tests/test_torch_math.py runs the C twin of this
// header, oracle/torch_math_oracle.c, against torch.nn.functional on random inputs: bit-equal).
#pragma once

#include <hip/hip_runtime.h>

#pragma clang fp contract(off)

namespace ultra {

// One Welford accumulator over the 8 values it owns (features i, 8 + i, ..., 56 + i in that order).
struct Moments {
    float m1, m2;
};

__device__ __forceinline__ Moments welford8(const float (&x)[8]) {
    // c[j] = fl(1 / (j + 1))
    const float c[8] = {1.f, 0.5f, 1.f / 3.f, 0.25f, 0.2f, 1.f / 6.f, 1.f / 7.f, 0.125f};
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float delta = x[j] - m1;
        m1 = __builtin_fmaf(delta, c[j], m1);
        m2 = __builtin_fmaf(delta, x[j] - m1, m2);
    }
    return Moments{m1, m2};
}

// Sequential merge of the eight accumulators (each holds 8 values) -> (mean, rstd) of the 64 features.
__device__ __forceinline__ void merge8(const Moments (&w)[8], float eps, float &mean, float &rstd) {
    // c[i] = fl(8 / (8 (i + 1)))
    const float c[8] = {1.f, 0.5f, 8.f / 24.f, 0.25f, 8.f / 40.f, 8.f / 48.f, 8.f / 56.f, 0.125f};
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float m0 = (float)(8 * i);
        const float delta = w[i].m1 - m1;
        m1 = __builtin_fmaf(c[i], delta, m1);
        m2 = m2 + __builtin_fmaf(delta * delta * c[i], m0, w[i].m2);
    }
    mean = m1;
    rstd = 1.f / sqrtf(m2 / 64.f + eps);
}

__device__ __forceinline__ float ln_apply(float x, float mean, float rstd, float gamma, float beta) {
    return __builtin_fmaf((x - mean) * rstd, gamma, beta);
}

}  // namespace ultra
