// The readout of a training step (models.py:202-207: score = mlp(cat[hidden, query]) on the candidates' rows) and its backward.
//
//     f[b, j]   = [hid[b, j] (64) ; query[b] (64)]                      hid (batch, n, 64): the last layer evaluated on the candidates' rows
//     h[b, j]   = relu(f[b, j] W1^T + b1)          W1 (128, 128) = mlp.0.weight
//     s[b, j]   = h[b, j] . w2 + b2                w2 (128)      = mlp.2.weight
//     gh        = (g (x) w2) * [h > 0]
//     ghid      = (gh W1)[:, :64]          gquery[b] = sum_j (gh W1)[b, j, 64:]
//     gW1       = gh^T f      gb1 = sum_rows gh      gw2 = sum_rows g h      gb2 = sum_rows g
//
// torch ran this as cat + two hipBLASLt products + relu + copies forward (7 launches, ~ 45 us) and a dozen launches backward (four
// products with M or K = 2,056, three reductions, fills, copies: ~ 95 us of a 2.9-ms step at FB15k237's size; profiles/r6_17).  Here:
// one launch forward, two backward.  A workgroup owns 32 candidates of ONE sample (a sample's last tile is partly empty), so the
// query half of f and the sum over a sample's candidates stay inside workgroups; W1 sits in LDS.  All products are plain fp32 FMAs
// in registers (33.7 MFLOP a pass: the launches are latency, not arithmetic); every sum has a fixed order -- a k-ascending chain per
// output, lanes combined by a fixed butterfly, tiles added in ascending order by the reduce kernel: run-to-run reproducible.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "../../include/ultra_nbfnet.h"
#include "../../include/ultra_rspmm.h"
#include "plan.hpp"
#include "device_scope.hpp"

namespace ultra {

constexpr int RT_TILE = 32;            // candidates per workgroup
constexpr int RT_F = 128;              // feature width = hidden width of the MLP
constexpr int RT_STRIDE = 132;         // LDS row stride (floats): 16-byte aligned rows, consecutive rows 4 banks apart
constexpr int RT_PART = RT_F * RT_F + 3 * RT_F + 64;      // floats of one tile's partial sums: gW1 | gb1 | gw2 | gb2 (1, padded) | gquery

struct ReadoutTrainParams {
    const float *hid, *query, *w1, *b1, *w2, *b2;
    float *h;                           // (batch * n, 128) post-ReLU activations (saved for the backward)
    float *score;                       // (batch, n)
    const float *grad;                  // (batch, n)
    float *ghid;                        // (batch, n, 64)
    float *part;                        // (batch * tiles, RT_PART)
    int batch, n, tiles;                // tiles per sample
};

// f tile of workgroup (sample b, tile t) -> LDS; rows past the sample's end are zero.  w1 -> LDS.
__device__ __forceinline__ void rt_stage(const ReadoutTrainParams &p, int b, int row0, float *f_lds, float *w_lds) {
    const int tid = threadIdx.x;
    for (int i = tid; i < RT_F * (RT_F / 4); i += 256) {
        const int c = i >> 5, k4 = i & 31;
        *reinterpret_cast<float4 *>(w_lds + c * RT_STRIDE + 4 * k4) = reinterpret_cast<const float4 *>(p.w1)[i];
    }
    for (int i = tid; i < RT_TILE * (RT_F / 4); i += 256) {
        const int r = i >> 5, k4 = i & 31;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + r < p.n)
            v = k4 < 16 ? reinterpret_cast<const float4 *>(p.hid + ((long long)b * p.n + row0 + r) * 64)[k4]
                        : reinterpret_cast<const float4 *>(p.query + (long long)b * 64)[k4 - 16];
        *reinterpret_cast<float4 *>(f_lds + r * RT_STRIDE + 4 * k4) = v;
    }
}

// acc[i][j] = sum_k a[(4 rb + i)][k] * w[(cb + 32 j)][k], k ascending: thread (rb, cb) of 8 x 32
__device__ __forceinline__ void rt_rows_times_wt(const float *a_lds, const float *w_lds, int rb, int cb, float (&acc)[4][4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k = 0; k < RT_F; k += 4) {
        float4 a[4], w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const float4 *>(a_lds + (4 * rb + i) * RT_STRIDE + k);
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const float4 *>(w_lds + (cb + 32 * j) * RT_STRIDE + k);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = fmaf(a[i].x, w[j].x, acc[i][j]);
                acc[i][j] = fmaf(a[i].y, w[j].y, acc[i][j]);
                acc[i][j] = fmaf(a[i].z, w[j].z, acc[i][j]);
                acc[i][j] = fmaf(a[i].w, w[j].w, acc[i][j]);
            }
    }
}

__global__ void __launch_bounds__(256) readout_train_fwd_kernel(const ReadoutTrainParams p) {
    __shared__ __attribute__((aligned(16))) float w_lds[RT_F * RT_STRIDE];
    __shared__ __attribute__((aligned(16))) float f_lds[RT_TILE * RT_STRIDE];
    const int tid = threadIdx.x, rb = tid >> 5, cb = tid & 31;
    const int b = blockIdx.x / p.tiles, row0 = (blockIdx.x % p.tiles) * RT_TILE;
    rt_stage(p, b, row0, f_lds, w_lds);
    __syncthreads();
    float acc[4][4];
    rt_rows_times_wt(f_lds, w_lds, rb, cb, acc);
    float b1[4], w2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        b1[j] = p.b1[cb + 32 * j];
        w2[j] = p.w2[cb + 32 * j];
    }
    const float b2 = p.b2[0];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = row0 + 4 * rb + i;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float h = fmaxf(acc[i][j] + b1[j], 0.f);
            if (row < p.n) p.h[((long long)b * p.n + row) * RT_F + cb + 32 * j] = h;
            s = fmaf(h, w2[j], s);
        }
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) s += __shfl_xor(s, off);      // (the 32 lanes of one rb: a half wave)
        if (cb == 0 && row < p.n) p.score[(long long)b * p.n + row] = s + b2;
    }
}

__global__ void __launch_bounds__(256) readout_train_bwd_kernel(const ReadoutTrainParams p) {
    __shared__ __attribute__((aligned(16))) float w_lds[RT_F * RT_STRIDE];
    __shared__ __attribute__((aligned(16))) float f_lds[RT_TILE * RT_STRIDE];
    __shared__ __attribute__((aligned(16))) float gh_lds[RT_TILE * RT_STRIDE];
    __shared__ float fold[8 * 64];
    __shared__ float g_lds[RT_TILE];
    const int tid = threadIdx.x, rb = tid >> 5, cb = tid & 31;
    const int b = blockIdx.x / p.tiles, row0 = (blockIdx.x % p.tiles) * RT_TILE;
    float *part = p.part + (long long)blockIdx.x * RT_PART;
    rt_stage(p, b, row0, f_lds, w_lds);
    if (tid < RT_TILE) g_lds[tid] = row0 + tid < p.n ? p.grad[(long long)b * p.n + row0 + tid] : 0.f;
    __syncthreads();
    // gh tile; the column sums gb1 / gw2 over the tile's rows (thread = column x half of the rows, rows ascending) and gb2
    {
        const int c = tid & 127, half = tid >> 7;
        const float w2 = p.w2[c];
        float sb1 = 0.f, sw2 = 0.f;
        for (int r = 16 * half; r < 16 * half + 16; ++r) {
            const float g = g_lds[r];
            const float h = row0 + r < p.n ? p.h[((long long)b * p.n + row0 + r) * RT_F + c] : 0.f;
            const float gh = h > 0.f ? g * w2 : 0.f;
            gh_lds[r * RT_STRIDE + c] = gh;
            sb1 += gh;
            sw2 = fmaf(g, h, sw2);
        }
        fold[tid] = sb1;
        fold[256 + tid] = sw2;
    }
    __syncthreads();
    if (tid < 128) {
        part[RT_F * RT_F + tid] = fold[tid] + fold[128 + tid];
        part[RT_F * RT_F + RT_F + tid] = fold[256 + tid] + fold[256 + 128 + tid];
    }
    if (tid == 0) {
        float s = 0.f;
        for (int r = 0; r < RT_TILE; ++r) s += g_lds[r];
        part[RT_F * RT_F + 2 * RT_F] = s;
    }
    __syncthreads();
    // gf[r][k] = sum_c gh[r][c] W1[c][k], c ascending: thread (rb, kb = cb) owns rows 4 rb .. + 3, k = kb + 32 j
    {
        float acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
        for (int c = 0; c < RT_F; c += 4) {
            float4 a[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const float4 *>(gh_lds + (4 * rb + i) * RT_STRIDE + c);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = w_lds[(c + u) * RT_STRIDE + cb + 32 * j];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float av = u == 0 ? a[i].x : u == 1 ? a[i].y : u == 2 ? a[i].z : a[i].w;
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av, w[j], acc[i][j]);
                }
            }
        }
        float q0 = 0.f, q1 = 0.f;              // the query half, summed over this thread's rows (ascending)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = row0 + 4 * rb + i;
            if (row < p.n) {
                p.ghid[((long long)b * p.n + row) * 64 + cb] = acc[i][0];
                p.ghid[((long long)b * p.n + row) * 64 + cb + 32] = acc[i][1];
            }
            q0 += acc[i][2];                    // (rows past the sample's end have gh = 0)
            q1 += acc[i][3];
        }
        __syncthreads();                         // (fold is free again)
        fold[rb * 64 + cb] = q0;
        fold[rb * 64 + 32 + cb] = q1;
        __syncthreads();
        if (tid < 64) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) s += fold[r * 64 + tid];
            part[RT_F * RT_F + 3 * RT_F + tid] = s;
        }
    }
    // gW1[c][k] partial = sum_r gh[r][c] f[r][k], r ascending: thread (ci = rb, kb = cb) owns c = 16 ci .. + 15, k = kb + 32 j
    {
        float acc[16][4];
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
        for (int r = 0; r < RT_TILE; ++r) {
            float4 a[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const float4 *>(gh_lds + r * RT_STRIDE + 16 * rb + 4 * i);
            float f[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) f[j] = f_lds[r * RT_STRIDE + cb + 32 * j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[4 * i + 0][j] = fmaf(a[i].x, f[j], acc[4 * i + 0][j]);
                    acc[4 * i + 1][j] = fmaf(a[i].y, f[j], acc[4 * i + 1][j]);
                    acc[4 * i + 2][j] = fmaf(a[i].z, f[j], acc[4 * i + 2][j]);
                    acc[4 * i + 3][j] = fmaf(a[i].w, f[j], acc[4 * i + 3][j]);
                }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) part[(16 * rb + i) * RT_F + cb + 32 * j] = acc[i][j];
    }
}

// tiles added in ascending order: gW1 (128 x 128), gb1, gw2 (128 each), gb2, and per sample gquery (64) over the sample's tiles
__global__ void __launch_bounds__(256) readout_train_reduce_kernel(const float *__restrict__ part, int batch, int tiles, float *gw1,
                                                                  float *gb1, float *gw2, float *gb2, float *gquery) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int shared = RT_F * RT_F + 2 * RT_F + 1;
    if (i < shared) {
        float s = 0.f;
        const int total = batch * tiles;
        int t = 0;
        for (; t + 8 <= total; t += 8) {          // (eight loads in flight, added in tile order)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(long long)(t + u) * RT_PART + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; t < total; ++t) s += part[(long long)t * RT_PART + i];
        if (i < RT_F * RT_F)
            gw1[i] = s;
        else if (i < RT_F * RT_F + RT_F)
            gb1[i - RT_F * RT_F] = s;
        else if (i < RT_F * RT_F + 2 * RT_F)
            gw2[i - RT_F * RT_F - RT_F] = s;
        else
            gb2[0] = s;
    } else if (i < shared + batch * 64) {
        const int b = (i - shared) / 64, k = (i - shared) % 64;
        float s = 0.f;
        for (int t = 0; t < tiles; ++t) s += part[(long long)(b * tiles + t) * RT_PART + RT_F * RT_F + 3 * RT_F + k];
        gquery[b * 64 + k] = s;
    }
}

}  // namespace ultra

static bool readout_train_shape_ok(int64_t batch, int64_t n) { return batch > 0 && n > 0 && batch <= 4096 && n <= (1 << 20); }

extern "C" int32_t ultra_readout_train_forward(const void *hid, const void *query, const void *w1, const void *b1, const void *w2,
                                               const void *b2, void *h, void *score, int64_t batch, int64_t n, void *stream) {
    ULTRA_DEVICE_SCOPE(stream, score);
    if (!hid || !query || !w1 || !b1 || !w2 || !b2 || !h || !score || !readout_train_shape_ok(batch, n)) {
        ultra::set_error("ultra_readout_train_forward: NULL operand or empty batch");
        return ULTRA_ERR_INVALID;
    }
    ultra::ReadoutTrainParams p{};
    p.hid = (const float *)hid, p.query = (const float *)query, p.w1 = (const float *)w1, p.b1 = (const float *)b1;
    p.w2 = (const float *)w2, p.b2 = (const float *)b2, p.h = (float *)h, p.score = (float *)score;
    p.batch = (int)batch, p.n = (int)n, p.tiles = (int)((n + ultra::RT_TILE - 1) / ultra::RT_TILE);
    (void)hipGetLastError();   // drop any stale error left by other users of the runtime
    hipLaunchKernelGGL(ultra::readout_train_fwd_kernel, dim3((unsigned)(p.batch * p.tiles)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), p);
    if (hipGetLastError() != hipSuccess) {
        ultra::set_error("readout_train_fwd_kernel launch failed");
        return ULTRA_ERR_HIP;
    }
    return ULTRA_OK;
}

extern "C" int64_t ultra_readout_train_backward_workspace(int64_t batch, int64_t n) {
    if (!readout_train_shape_ok(batch, n)) return 0;
    return batch * ((n + ultra::RT_TILE - 1) / ultra::RT_TILE) * (int64_t)ultra::RT_PART * 4;
}

extern "C" int32_t ultra_readout_train_backward(const void *grad_score, const void *h, const void *hid, const void *query,
                                                const void *w1, const void *w2, void *grad_hid, void *grad_query, void *grad_w1,
                                                void *grad_b1, void *grad_w2, void *grad_b2, void *work, int64_t work_bytes,
                                                int64_t batch, int64_t n, void *stream) {
    ULTRA_DEVICE_SCOPE(stream, grad_hid);
    if (!grad_score || !h || !hid || !query || !w1 || !w2 || !grad_hid || !grad_query || !grad_w1 || !grad_b1 || !grad_w2 ||
        !grad_b2 || !work || !readout_train_shape_ok(batch, n)) {
        ultra::set_error("ultra_readout_train_backward: NULL operand or empty batch");
        return ULTRA_ERR_INVALID;
    }
    if (work_bytes < ultra_readout_train_backward_workspace(batch, n)) {
        ultra::set_error("ultra_readout_train_backward: workspace smaller than ultra_readout_train_backward_workspace()");
        return ULTRA_ERR_INVALID;
    }
    ultra::ReadoutTrainParams p{};
    p.hid = (const float *)hid, p.query = (const float *)query, p.w1 = (const float *)w1, p.w2 = (const float *)w2;
    p.h = (float *)const_cast<void *>(h), p.grad = (const float *)grad_score, p.ghid = (float *)grad_hid, p.part = (float *)work;
    p.batch = (int)batch, p.n = (int)n, p.tiles = (int)((n + ultra::RT_TILE - 1) / ultra::RT_TILE);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    (void)hipGetLastError();   // drop any stale error left by other users of the runtime
    hipLaunchKernelGGL(ultra::readout_train_bwd_kernel, dim3((unsigned)(p.batch * p.tiles)), dim3(256), 0, s, p);
    const int outputs = ultra::RT_F * ultra::RT_F + 2 * ultra::RT_F + 1 + p.batch * 64;
    hipLaunchKernelGGL(ultra::readout_train_reduce_kernel, dim3((unsigned)((outputs + 255) / 256)), dim3(256), 0, s,
                       (const float *)work, p.batch, p.tiles, (float *)grad_w1, (float *)grad_b1, (float *)grad_w2, (float *)grad_b2,
                       (float *)grad_query);
    if (hipGetLastError() != hipSuccess) {
        ultra::set_error("readout_train_bwd_kernel launch failed");
        return ULTRA_ERR_HIP;
    }
    return ULTRA_OK;
}
