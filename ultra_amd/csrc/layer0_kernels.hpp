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
#include "torch_math.hpp"

namespace ultra {

// L0_MAX: the max aggregate (layers.py:206-207).  The messages of the zero rows are exact zeros and the boundary tensor is
// zero off the source row, so a reached row aggregates max(0, messages from s[b]) and every other row 0 -- the same constant
// output row as under the sum; the source row meets its boundary value q, the messages of its self loops, and a zero only
// if some OTHER node has an edge onto it.
// L0_ONLY_FILL / L0_SKIP_FILL: the constant fill depends on the layer's parameters only -- a caller may launch it ahead of
// time (beside the relation model, whose output the special rows need) and the special rows later.
enum { L0_LN = 1, L0_RELU = 2, L0_RESIDUAL = 4, L0_MAX = 8, L0_ONLY_FILL = 16, L0_SKIP_FILL = 32 };

struct Layer0Params {
    const int32_t *trow_ptr;   // transposed plan: row = gathered source
    const int32_t *tcol;       // = aggregation target
    const int32_t *ttype;
    const int32_t *tperm;
    const uint8_t *self_loop;  // [num_node] bit 0: the node has an edge onto itself, bit 1: an in-edge from another node
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
struct L0Vec {
    float bias[4], ln_w[4], ln_b[4];
};

__device__ __forceinline__ L0Vec l0_load_vectors(const Layer0Params &p, int l16) {
    L0Vec v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v.bias[e] = p.bias ? p.bias[4 * l16 + e] : 0.f;
        v.ln_w[e] = (p.flags & L0_LN) ? p.ln_w[4 * l16 + e] : 1.f;
        v.ln_b[e] = (p.flags & L0_LN) ? p.ln_b[4 * l16 + e] : 0.f;
    }
    return v;
}

// LayerNorm (the reference's operation order, torch_math.hpp) and ReLU of a row held by a 16-lane group, feature
// 4 l16 + e in register e.  `slot` = 64 + 16 floats of LDS private to the group: the row is parked there, lanes 0..7 run
// the Welford accumulator i = l16 over features 8 j + i, every lane merges the eight (LDS operations of one wave
// execute in order: no barrier between the group's writes and reads).
__device__ __forceinline__ void l0_finish(float (&y)[4], const Layer0Params &p, const L0Vec &vec, float *slot, const int l16) {
    if (p.flags & L0_LN) {
        *reinterpret_cast<float4 *>(slot + 4 * l16) = make_float4(y[0], y[1], y[2], y[3]);
        float xv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = slot[8 * j + (l16 & 7)];
        const Moments w = welford8(xv);
        if (l16 < 8) {
            slot[64 + 2 * l16] = w.m1;
            slot[64 + 2 * l16 + 1] = w.m2;
        }
        Moments all[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) all[i] = Moments{slot[64 + 2 * i], slot[64 + 2 * i + 1]};
        float mean, rstd;
        merge8(all, p.eps, mean, rstd);
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = ln_apply(y[e], mean, rstd, vec.ln_w[e], vec.ln_b[e]);
    }
    if (p.flags & L0_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], 0.f);
    }
}

// out[b, n, :] = c0 = relu(LayerNorm(bias)) for every row: what the layer makes of x0 = agg = 0.
__global__ void __launch_bounds__(256) nbf_layer0_fill_kernel(const Layer0Params p) {
    __shared__ __attribute__((aligned(16))) float lds_slot[16 * 80];   // one (64 + 16)-float slot per 16-lane group
    const int l16 = threadIdx.x & 15;
    const L0Vec vec = l0_load_vectors(p, l16);
    float c0[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) c0[e] = vec.bias[e];   // W . 0 + bias
    l0_finish(c0, p, vec, lds_slot + (threadIdx.x >> 4) * 80, l16);
    const float4 v = make_float4(c0[0], c0[1], c0[2], c0[3]);
    if (p.out_sr == 64 && p.out_so == p.num_node * 64) {
        // contiguous output: a plain 16-byte stream; the stride is a multiple of 16 chunks, so a thread always writes
        // the same four features
        float4 *o4 = reinterpret_cast<float4 *>(p.out);
        const long long n16 = (long long)p.n_outer * p.num_node * 16;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) o4[i] = v;
        return;
    }
    const long long total = (long long)p.n_outer * p.num_node;
    const long long g0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4, ng = ((long long)gridDim.x * blockDim.x) >> 4;
    for (long long r = g0; r < total; r += ng) {
        const long long b = r / p.num_node, n = r - b * p.num_node;
        *reinterpret_cast<float4 *>(p.out + b * p.out_so + n * p.out_sr + 4 * l16) = v;
    }
}

// The special rows.  grid = (blocks per sample, n_outer); one 16-lane group per run of equal targets in s[b]'s
// out-edge list (sorted by target in the transposed plan), group 0 of block 0 also covers s[b] itself when no edge
// leads back to it.  W is staged transposed in LDS (lds_wt[k][f] = W[f][k]: a lane reads
// the 4 weights of its features with one 16-byte load); the group's 64-vector goes through a 256-byte LDS slot and
// is read back as 16-byte broadcasts, so the 64-term dot products are plain unrolled FMAs on pipelined LDS reads.
__global__ void __launch_bounds__(1024) nbf_layer0_rows_kernel(const Layer0Params p) {
    __shared__ __attribute__((aligned(16))) float lds_wt[128 * 64];
    __shared__ __attribute__((aligned(16))) float lds_vec[64 * 80];   // one 64-vector (+ 16 LayerNorm moments) per group
    const int outer = blockIdx.y;
    const int l16 = threadIdx.x & 15;
    const int grp = blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);
    const int ngrp = gridDim.x * (blockDim.x >> 4);
    long long s = p.src[outer];
    s = s < 0 ? 0 : (s >= p.num_node ? p.num_node - 1 : s);   // (an out-of-range id reads a valid row instead of faulting)
    const int k0 = p.trow_ptr[s], k1 = p.trow_ptr[s + 1];
    if (blockIdx.x > 0 && k0 + (int)blockIdx.x * (int)(blockDim.x >> 4) >= k1) return;   // no run can start in this block
    // small vectors first: their latency hides under the weight staging instead of ending the dependency chain
    const L0Vec vec = l0_load_vectors(p, l16);
    float qv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) qv[e] = p.q ? p.q[(long long)outer * 64 + 4 * l16 + e] : 1.f;
    {   // transpose W[f][k] -> [k][f]: lane = f (64 distinct LDS banks per store), 16 bytes of one weight row per load
        const int f = threadIdx.x & 63;
#pragma unroll
        for (int kc = 4 * (threadIdx.x >> 6); kc < 128; kc += 4 * (blockDim.x >> 6)) {
            const float4 wv = *reinterpret_cast<const float4 *>(p.weight + f * 128 + kc);
            lds_wt[(kc + 0) * 64 + f] = wv.x;
            lds_wt[(kc + 1) * 64 + f] = wv.y;
            lds_wt[(kc + 2) * 64 + f] = wv.z;
            lds_wt[(kc + 3) * 64 + f] = wv.w;
        }
    }
    __syncthreads();
    const float *relb = reinterpret_cast<const float *>(p.rel.ptr) + outer * p.rel.stride_outer;
    float *outb = p.out + outer * p.out_so;
    float *slot = lds_vec + (threadIdx.x >> 4) * 80;

    // y = (W[:, :64] . x + W[:, 64:] . agg) + bias for this group's row; x = q on the source row, 0 elsewhere.  One fmaf
    // chain per output over k = 0..127, bias added last: the reference's nn.Linear order (torch_math.hpp); the terms of
    // an all-zero x are exact zeros and are skipped.
    const auto update_row = [&](const float (&agg)[4], bool is_src, long long row) {
        float y[4] = {0.f, 0.f, 0.f, 0.f};
        if (is_src) {   // once per sample: the input half of W
            *reinterpret_cast<float4 *>(slot + 4 * l16) = make_float4(qv[0], qv[1], qv[2], qv[3]);
#pragma unroll
            for (int k4 = 0; k4 < 16; ++k4) {
                const float4 x4 = *reinterpret_cast<const float4 *>(slot + 4 * k4);
                const float xs[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float4 wv = *reinterpret_cast<const float4 *>(lds_wt + (4 * k4 + e) * 64 + 4 * l16);
                    y[0] = __builtin_fmaf(xs[e], wv.x, y[0]);
                    y[1] = __builtin_fmaf(xs[e], wv.y, y[1]);
                    y[2] = __builtin_fmaf(xs[e], wv.z, y[2]);
                    y[3] = __builtin_fmaf(xs[e], wv.w, y[3]);
                }
            }
        }
        *reinterpret_cast<float4 *>(slot + 4 * l16) = make_float4(agg[0], agg[1], agg[2], agg[3]);
#pragma unroll
        for (int k4 = 0; k4 < 16; ++k4) {
            const float4 a4 = *reinterpret_cast<const float4 *>(slot + 4 * k4);   // same address in the group: broadcast
            const float as[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float4 wv = *reinterpret_cast<const float4 *>(lds_wt + (64 + 4 * k4 + e) * 64 + 4 * l16);
                y[0] = __builtin_fmaf(as[e], wv.x, y[0]);
                y[1] = __builtin_fmaf(as[e], wv.y, y[1]);
                y[2] = __builtin_fmaf(as[e], wv.z, y[2]);
                y[3] = __builtin_fmaf(as[e], wv.w, y[3]);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] += vec.bias[e];
        l0_finish(y, p, vec, slot, l16);
        if (is_src && (p.flags & L0_RESIDUAL)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] += qv[e];
        }
        *reinterpret_cast<float4 *>(outb + row * p.out_sr + 4 * l16) = make_float4(y[0], y[1], y[2], y[3]);
    };

    for (int k = k0 + grp; k < k1; k += ngrp) {
        // run head test and the first four entries of the run in one round of loads (runs are short: one edge per
        // relation type at most in a simple graph)
        const int last = k1 - 1;
        const int prev = p.tcol[k > k0 ? k - 1 : k0];
        int tc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) tc[u] = p.tcol[min(k + u, last)];
        const int target = tc[0];
        if (k > k0 && prev == target) continue;
        int kend = k + 1;
        if (kend < k1 && tc[1] == target) ++kend;
        if (kend == k + 2 && kend < k1 && tc[2] == target) ++kend;
        if (kend == k + 3 && kend < k1 && tc[3] == target) ++kend;
        if (kend == k + 4)
            while (kend < k1 && p.tcol[kend] == target) ++kend;
        const bool is_src = target == (int)s;
        const bool use_max = (p.flags & L0_MAX) != 0;
        // (max: the zeros a row meets anyway -- messages of zero rows, the boundary tensor off the source row; on the source
        // row only the former, and only if another node has an edge onto it)
        const float init = (use_max && is_src && !(p.self_loop[s] & 2)) ? -3.402823466e+38f : 0.f;
        float agg[4] = {init, init, init, init};
        for (int kb = k; kb < kend; kb += 4) {   // edge order of the run (deterministic)
            float4 rv[4];
            float wv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kk = min(kb + u, kend - 1);
                const int ty = p.ttype[kk];
                rv[u] = *reinterpret_cast<const float4 *>(relb + (long long)ty * p.rel.stride_row + 4 * l16);
                wv[u] = p.w ? p.w[p.tperm[kk]] : 1.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (kb + u < kend) {
                    const float r4[4] = {rv[u].x, rv[u].y, rv[u].z, rv[u].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float m = r4[e] * qv[e];
                        if (p.w) m = wv[u] * m;
                        agg[e] = use_max ? fmaxf(agg[e], m) : agg[e] + m;
                    }
                }
            }
        }
        if (is_src) {   // update + boundary (layers.py:200), max(update, boundary) (layers.py:207)
#pragma unroll
            for (int e = 0; e < 4; ++e) agg[e] = use_max ? fmaxf(agg[e], qv[e]) : agg[e] + qv[e];
        }
        update_row(agg, is_src, target);
    }
    // no edge leads back to the source row: its aggregate is the boundary value alone (under max: against the zero
    // messages of its other in-edges, if it has any)
    if (grp == 0 && !(p.self_loop[s] & 1)) {
        float a0[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) a0[e] = ((p.flags & L0_MAX) && (p.self_loop[s] & 2)) ? fmaxf(qv[e], 0.f) : qv[e];
        update_row(a0, true, s);
    }
}

}  // namespace ultra
