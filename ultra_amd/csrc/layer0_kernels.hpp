// Layer 0 of an NBFNet on its own boundary condition, computed where it is non-trivial.
//
// The first GeneralizedRelationalConv of RelNBFNet / EntityNBFNet reads the boundary condition as its input
// (models.py:59-66 / 135-141 build it as zeros + one row per sample; models.py:72-80 / 150-163 feed it to layer 0):
//     x0[b, n]  = q[b] if n == s[b] else 0
//     agg[b, n] = sum_{e: row_e == n, col_e == s[b]} w_e * rel[b, type_e] * q[b]   +   x0[b, n]        (layers.py:183-207, sum/distmult)
//     out[b, n] = [x0[b, n] +] relu( LayerNorm( W . [x0[b, n] ; agg[b, n]] + bias ) )                  (layers.py:233-240, models.py:158-160)
// Every node that is neither s[b] nor the target of an edge leaving s[b] has x0 = agg = 0, hence the SAME output
// row c0 = relu(LayerNorm(bias)).  So: one streaming fill with c0, then one 16-lane group per "special" row
// (the distinct targets of s[b]'s out-edges in the transposed plan, and s[b] itself) does the 128-term update by
// hand.  Exact arithmetic of the dense formulation (products of exact zeros dropped), no (batch, N, d) boundary or
// aggregate tensor is ever materialised.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "rspmm_kernels.hpp"

namespace ultra {

enum { L0_LN = 1, L0_RELU = 2, L0_RESIDUAL = 4 };

struct Layer0Params {
    const int32_t *trow_ptr;   // transposed plan: row = gathered source
    const int32_t *tcol;       // = aggregation target
    const int32_t *ttype;
    const int32_t *tperm;
    const float *w;            // edge weights in original edge order, or NULL
    const int64_t *src;        // [n_outer] source row s[b]
    const float *q;            // [n_outer][64] boundary value of the source row, or NULL = ones (RelNBFNet)
    MatArg rel;                // (n_outer, num_rel, 64)
    const float *weight;       // (64, 128) row-major = linear.weight
    const float *bias, *ln_w, *ln_b;   // (64); bias may be NULL
    float *out;
    long long out_so, out_sr;
    long long num_node;
    int32_t n_outer;
    float eps;
    int32_t flags;
};

// LayerNorm / ReLU of one 64-feature row held by a 16-lane group (4 features per lane), layers.py:235-238
__device__ __forceinline__ void l0_finish(float (&y)[4], const Layer0Params &p, int l16) {
    if (p.flags & L0_LN) {
        float s = (y[0] + y[1]) + (y[2] + y[3]);
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 16);
        const float mean = s * (1.f / 64.f);
        float qv = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = y[e] - mean;
            qv += d * d;
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) qv += __shfl_xor(qv, off, 16);
        const float rstd = 1.f / sqrtf(qv * (1.f / 64.f) + p.eps);
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = (y[e] - mean) * rstd * p.ln_w[4 * l16 + e] + p.ln_b[4 * l16 + e];
    }
    if (p.flags & L0_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], 0.f);
    }
}

// out[b, n, :] = c0 = relu(LayerNorm(bias)) for every row: what the layer makes of x0 = agg = 0.
__global__ void __launch_bounds__(256) nbf_layer0_fill_kernel(const Layer0Params p) {
    const int l16 = threadIdx.x & 15;
    float c0[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) c0[e] = p.bias ? p.bias[4 * l16 + e] : 0.f;
    l0_finish(c0, p, l16);
    const float4 v = make_float4(c0[0], c0[1], c0[2], c0[3]);
    const long long total = (long long)p.n_outer * p.num_node;
    const long long g0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4, ng = ((long long)gridDim.x * blockDim.x) >> 4;
    for (long long r = g0; r < total; r += ng) {
        const long long b = r / p.num_node, n = r - b * p.num_node;
        *reinterpret_cast<float4 *>(p.out + b * p.out_so + n * p.out_sr + 4 * l16) = v;
    }
}

// The special rows.  grid = (blocks per sample, n_outer); one 16-lane group per run of equal targets in s[b]'s
// out-edge list (sorted by target in the transposed plan), group 0 of block 0 also covers s[b] itself when no edge
// leads back to it.  lds_wt[k][f] = W[f][k]: a lane reads the 4 weights of its features with one 16-byte LDS load.
__global__ void __launch_bounds__(256) nbf_layer0_rows_kernel(const Layer0Params p) {
    __shared__ __attribute__((aligned(16))) float lds_wt[128 * 64];
    const int outer = blockIdx.y;
    const int l16 = threadIdx.x & 15;
    const int grp = blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);
    const int ngrp = gridDim.x * (blockDim.x >> 4);
    long long s = p.src[outer];
    s = s < 0 ? 0 : (s >= p.num_node ? p.num_node - 1 : s);   // (an out-of-range id reads a valid row instead of faulting)
    const int k0 = p.trow_ptr[s], k1 = p.trow_ptr[s + 1];
    if (blockIdx.x > 0 && k0 + (int)blockIdx.x * (int)(blockDim.x >> 4) >= k1) return;   // no run can start in this block
    {   // transpose W[f][k] -> [k][f]: lane = f (64 distinct LDS banks per store), 16 bytes of one weight row per load
        const int f = threadIdx.x & 63;
        for (int kc = 4 * (threadIdx.x >> 6); kc < 128; kc += 4 * (blockDim.x >> 6)) {
            const float4 wv = *reinterpret_cast<const float4 *>(p.weight + f * 128 + kc);
            lds_wt[(kc + 0) * 64 + f] = wv.x;
            lds_wt[(kc + 1) * 64 + f] = wv.y;
            lds_wt[(kc + 2) * 64 + f] = wv.z;
            lds_wt[(kc + 3) * 64 + f] = wv.w;
        }
    }
    __syncthreads();
    float qv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) qv[e] = p.q ? p.q[(long long)outer * 64 + 4 * l16 + e] : 1.f;
    const float *relb = reinterpret_cast<const float *>(p.rel.ptr) + outer * p.rel.stride_outer;
    float *outb = p.out + outer * p.out_so;

    // y = bias + W[:, :64] . x + W[:, 64:] . agg for this group's row; x = q on the source row, 0 elsewhere
    const auto update_row = [&](const float (&agg)[4], bool is_src, long long row) {
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = p.bias ? p.bias[4 * l16 + e] : 0.f;
        if (is_src) {
            for (int k = 0; k < 64; ++k) {
                const float xk = __shfl(qv[k & 3], k >> 2, 16);
                const float4 wv = *reinterpret_cast<const float4 *>(lds_wt + k * 64 + 4 * l16);
                y[0] += wv.x * xk;
                y[1] += wv.y * xk;
                y[2] += wv.z * xk;
                y[3] += wv.w * xk;
            }
        }
        for (int k = 0; k < 64; ++k) {
            const float ak = __shfl(agg[k & 3], k >> 2, 16);
            const float4 wv = *reinterpret_cast<const float4 *>(lds_wt + (64 + k) * 64 + 4 * l16);
            y[0] += wv.x * ak;
            y[1] += wv.y * ak;
            y[2] += wv.z * ak;
            y[3] += wv.w * ak;
        }
        l0_finish(y, p, l16);
        if (is_src && (p.flags & L0_RESIDUAL)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] += qv[e];
        }
        *reinterpret_cast<float4 *>(outb + row * p.out_sr + 4 * l16) = make_float4(y[0], y[1], y[2], y[3]);
    };

    for (int k = k0 + grp; k < k1; k += ngrp) {
        const int target = p.tcol[k];
        if (k > k0 && p.tcol[k - 1] == target) continue;
        int kend = k + 1;
        while (kend < k1 && p.tcol[kend] == target) ++kend;
        float agg[4] = {0.f, 0.f, 0.f, 0.f};
        for (int kk = k; kk < kend; ++kk) {   // edge order of the run (deterministic)
            const float4 rv = *reinterpret_cast<const float4 *>(relb + (long long)p.ttype[kk] * p.rel.stride_row + 4 * l16);
            const float r4[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float m = r4[e] * qv[e];
                if (p.w) m = p.w[p.tperm[kk]] * m;
                agg[e] += m;
            }
        }
        const bool is_src = target == (int)s;
        if (is_src) {   // update + boundary (layers.py:200)
#pragma unroll
            for (int e = 0; e < 4; ++e) agg[e] += qv[e];
        }
        update_row(agg, is_src, target);
    }
    if (grp == 0) {
        // no edge leads back to the source row: its aggregate is the boundary value alone
        int lo = k0, hi = k1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (p.tcol[mid] < (int)s) lo = mid + 1; else hi = mid;
        }
        if (!(lo < k1 && p.tcol[lo] == (int)s)) update_row(qv, true, s);
    }
}

}  // namespace ultra
