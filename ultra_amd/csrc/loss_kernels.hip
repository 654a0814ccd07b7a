// The fine-tuning step's loss (script/run.py:66-77) and its gradient in one launch.
//
//     target      = [1, 0, ..., 0]                                        (column 0 is the positive)
//     l[b, i]     = binary_cross_entropy_with_logits(pred[b, i], target[i])
//     w[b, 0]     = 1,   w[b, 1:] = softmax(pred[b, 1:] / T)   (T > 0; constant: the reference takes it under no_grad)
//                                   or 1 / num_negative        (T == 0)
//     loss        = mean_b ( sum_i l[b, i] w[b, i] / sum_i w[b, i] )
//     d loss / d pred[b, i] = w[b, i] (sigmoid(pred[b, i]) - target[i]) / (sum_i w[b, i] * batch)
//
// The reference's op chain is ~ 25 elementwise / reduction launches forward and backward on a (8, 257) tensor; here one
// workgroup does both.  A wave owns a row at a time (rows wave, wave + 4, ...), sums run over lanes in a fixed butterfly and over
// rows in row order: run-to-run reproducible.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "../../include/ultra_nbfnet.h"
#include "../../include/ultra_rspmm.h"
#include "plan.hpp"
#include "device_scope.hpp"

namespace ultra {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

constexpr int LOSS_MAX_ROWS = 4096;

__global__ void __launch_bounds__(256) ranking_loss_kernel(const float *__restrict__ pred, int rows, int n, float temperature,
                                                           float uniform_w, float *__restrict__ loss, float *__restrict__ grad) {
    __shared__ float row_loss[LOSS_MAX_ROWS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv_rows = 1.f / (float)rows;
    for (int b = wave; b < rows; b += 4) {
        const float *p = pred + (long long)b * n;
        float *g = grad + (long long)b * n;
        float mx = -3.402823466e38f, z = 1.f;
        if (temperature > 0.f) {
            for (int i = 1 + lane; i < n; i += 64) mx = fmaxf(mx, p[i] / temperature);
            mx = wave_max(mx);
            float s = 0.f;
            for (int i = 1 + lane; i < n; i += 64) s += expf(p[i] / temperature - mx);
            z = wave_sum(s);
        }
        float lw = 0.f, wsum = 0.f;
        for (int i = lane; i < n; i += 64) {
            const float x = p[i], y = i == 0 ? 1.f : 0.f;
            const float w = i == 0 ? 1.f : (temperature > 0.f ? expf(x / temperature - mx) / z : uniform_w);
            // (1 - y) x - log_sigmoid(x),  log_sigmoid(x) = min(x, 0) - log1p(exp(-|x|))
            const float l = (1.f - y) * x - (fminf(x, 0.f) - log1pf(expf(-fabsf(x))));
            lw += l * w;
            wsum += w;
        }
        lw = wave_sum(lw);
        wsum = wave_sum(wsum);
        for (int i = lane; i < n; i += 64) {
            const float x = p[i], y = i == 0 ? 1.f : 0.f;
            const float w = i == 0 ? 1.f : (temperature > 0.f ? expf(x / temperature - mx) / z : uniform_w);
            const float sig = 1.f / (1.f + expf(-x));
            g[i] = w * (sig - y) / wsum * inv_rows;
        }
        if (lane == 0) row_loss[b] = lw / wsum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < rows; ++b) s += row_loss[b];
        *loss = s * inv_rows;
    }
}

}  // namespace ultra

using namespace ultra;

extern "C" int32_t ultra_ranking_loss(const void *pred, int64_t rows, int64_t n, float temperature, float uniform_weight,
                                      void *loss, void *grad, void *stream) {
    ULTRA_DEVICE_SCOPE(stream, pred);
    if (!pred || !loss || !grad || rows < 1 || rows > LOSS_MAX_ROWS || n < 2) {
        set_error("ultra_ranking_loss: NULL operand, rows outside [1, 4096] or fewer than two columns");
        return ULTRA_ERR_INVALID;
    }
    (void)hipGetLastError();
    hipLaunchKernelGGL(ranking_loss_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), (const float *)pred,
                       (int)rows, (int)n, temperature, uniform_weight, (float *)loss, (float *)grad);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error(std::string("ranking_loss_kernel launch: ") + hipGetErrorString(e));
        return ULTRA_ERR_HIP;
    }
    return ULTRA_OK;
}
