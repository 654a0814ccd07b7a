// rspmm restricted to a LIST of output rows (fine-tuning path, fp32, sum aggregate).
//
// A training step reads the last entity layer's output at its 1 + num_negative candidate rows only (models.py:202-207), so
// that layer needs, per sample, the aggregate of those rows alone -- the in-edges of ~ 2,000 rows instead of a walk over every
// edge of the graph (36 k against 17 M edge slices at YAGO3-10's size) -- and its backward only what flows back through them:
//
//   forward   agg[o, j]           = sum_{e : row_e == rows[o, j]} w_e * BINARY(rel[o, type_e], x[o, col_e])   (+ boundary[o, rows[o, j]])
//   backward  input_grad[o, col_e]   += w_e * d BINARY / d x   * agg_grad[o, j]          for the same edges   (rspmm.cpp:110-112)
//             relation_grad[o, type_e] += w_e * d BINARY / d rel * agg_grad[o, j]                             (rspmm.cpp:106-108)
//
// One workgroup per (sample, listed row): its 64 sixteen-lane groups take every 64th in-edge of the row from the plan's sorted
// edge list (row_ptr / col / type / perm), 16 bytes of the 64-element span per lane.  The backward is a scatter: several listed rows
// share sources and relation types, so it ADDS with float atomics into caller-zeroed (or caller-prefilled) gradients -- like
// the reference's own GPU backward (atomicAdd, rspmm.cu:153-214), the last bits of these two sums vary run to run.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "rspmm_kernels.hpp"

namespace ultra {

struct RowsParams {
    const int32_t *row_ptr, *col, *type, *perm;
    const float *w;                 // per-edge weights in ORIGINAL edge order (NULL: ones)
    MatArg rel, x, bnd;             // bnd.ptr == NULL: no boundary tensor
    const long long *rows;          // [n_outer][n_list]
    const long long *point_rows;    // point boundary: point_vals[o] is added where rows[o, j] == point_rows[o] (NULL: none)
    const float *point_vals;        // [n_outer][row_len]
    float *agg;                     // forward: [n_outer][n_list][row_len] out; backward: the same shape, its gradient (in)
    float *xgrad, *rgrad;           // backward: added into (strides of x / rel)
    long long xgrad_so, xgrad_sr, rgrad_so, rgrad_sr;
    int32_t n_outer, n_list, row_len, spans, mul_add;   // mul_add: 0 = rel * x (DistMult), 1 = rel + x (TransE)
};

// One workgroup of 16 waves (64 lane groups) per (sample, span, listed row): group G takes the row's in-edges G, G + 64, ...,
// four at a time -- a hub row of thousands of edges (the positives of a batch are degree-biased) is a few dozen rounds.
constexpr int ROWS_GROUPS = 64;
template <bool BACKWARD>
__global__ void __launch_bounds__(1024) rspmm_rows_kernel(const RowsParams p) {
    __shared__ __attribute__((aligned(16))) float part[16][64];
    const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, wave = tid >> 6, G = tid >> 4;
    const long long item = blockIdx.x;                                          // (outer, span, listed row)
    const int j = (int)(item % p.n_list);
    const int span = (int)((item / p.n_list) % p.spans);
    const int o = (int)(item / ((long long)p.n_list * p.spans));
    const int d0 = span * 64 + 4 * l16;
    const long long row = p.rows[(long long)o * p.n_list + j];
    const int beg = p.row_ptr[row], end = p.row_ptr[row + 1];
    const float *xb = reinterpret_cast<const float *>(p.x.ptr) + o * p.x.stride_outer + d0;
    const float *rb = reinterpret_cast<const float *>(p.rel.ptr) + o * p.rel.stride_outer + d0;
    float *cell = p.agg + ((long long)o * p.n_list + j) * p.row_len + d0;
    if (!BACKWARD) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // four edges in flight per group (a hub row of 16 k in-edges is 64 rounds of the workgroup's 256 instead of 128 of 128)
        for (int e0 = beg + G; e0 < end; e0 += 4 * ROWS_GROUPS) {
            int c[4], t[4];
            float w[4];
            bool live[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * ROWS_GROUPS;
                live[u] = e < end;
                const int ec = live[u] ? e : e0;
                c[u] = p.col[ec], t[u] = p.type[ec];
                w[u] = (p.w && live[u]) ? p.w[p.perm[ec]] : 1.f;
            }
            float4 xv[4], rv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xv[u] = *reinterpret_cast<const float4 *>(xb + (long long)c[u] * p.x.stride_row);
                rv[u] = *reinterpret_cast<const float4 *>(rb + (long long)t[u] * p.rel.stride_row);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!live[u]) continue;
                if (p.mul_add)
                    acc.x += w[u] * (rv[u].x + xv[u].x), acc.y += w[u] * (rv[u].y + xv[u].y), acc.z += w[u] * (rv[u].z + xv[u].z),
                        acc.w += w[u] * (rv[u].w + xv[u].w);
                else
                    acc.x += w[u] * (rv[u].x * xv[u].x), acc.y += w[u] * (rv[u].y * xv[u].y), acc.z += w[u] * (rv[u].z * xv[u].z),
                        acc.w += w[u] * (rv[u].w * xv[u].w);
            }
        }
        // a wave's four groups: 0 + 1, 2 + 3, then the two halves; the sixteen waves through LDS, in wave order
        acc.x += __shfl_xor(acc.x, 16), acc.y += __shfl_xor(acc.y, 16), acc.z += __shfl_xor(acc.z, 16), acc.w += __shfl_xor(acc.w, 16);
        acc.x += __shfl_xor(acc.x, 32), acc.y += __shfl_xor(acc.y, 32), acc.z += __shfl_xor(acc.z, 32), acc.w += __shfl_xor(acc.w, 32);
        if (lane < 16) *reinterpret_cast<float4 *>(&part[wave][4 * l16]) = acc;
        __syncthreads();
        if (tid >= 16) return;
        float4 s = *reinterpret_cast<const float4 *>(&part[0][4 * l16]);
#pragma unroll
        for (int wv = 1; wv < 16; ++wv) {
            const float4 a = *reinterpret_cast<const float4 *>(&part[wv][4 * l16]);
            s.x += a.x, s.y += a.y, s.z += a.z, s.w += a.w;
        }
        if (p.bnd.ptr) {
            const float4 b = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(p.bnd.ptr) + o * p.bnd.stride_outer +
                                                               row * p.bnd.stride_row + d0);
            s.x += b.x, s.y += b.y, s.z += b.z, s.w += b.w;
        }
        if (p.point_rows && p.point_rows[o] == row) {
            const float4 b = *reinterpret_cast<const float4 *>(p.point_vals + (long long)o * p.row_len + d0);
            s.x += b.x, s.y += b.y, s.z += b.z, s.w += b.w;
        }
        *reinterpret_cast<float4 *>(cell) = s;
    } else {
        const float4 g = *reinterpret_cast<const float4 *>(cell);
        float *xg = p.xgrad + o * p.xgrad_so + d0;
        float *rg = p.rgrad + o * p.rgrad_so + d0;
        for (int e = beg + G; e < end; e += ROWS_GROUPS) {
            const int c = p.col[e], t = p.type[e];
            const float w = p.w ? p.w[p.perm[e]] : 1.f;
            const float4 v = make_float4(w * g.x, w * g.y, w * g.z, w * g.w);
            float4 dx = v, dr = v;                              // TransE: both partial derivatives are 1
            if (!p.mul_add) {
                const float4 xv = *reinterpret_cast<const float4 *>(xb + (long long)c * p.x.stride_row);
                const float4 rv = *reinterpret_cast<const float4 *>(rb + (long long)t * p.rel.stride_row);
                dx = make_float4(rv.x * v.x, rv.y * v.y, rv.z * v.z, rv.w * v.w);
                dr = make_float4(xv.x * v.x, xv.y * v.y, xv.z * v.z, xv.w * v.w);
            }
            float *xd = xg + (long long)c * p.xgrad_sr, *rd = rg + (long long)t * p.rgrad_sr;
            atomicAdd(xd + 0, dx.x), atomicAdd(xd + 1, dx.y), atomicAdd(xd + 2, dx.z), atomicAdd(xd + 3, dx.w);
            atomicAdd(rd + 0, dr.x), atomicAdd(rd + 1, dr.y), atomicAdd(rd + 2, dr.z), atomicAdd(rd + 3, dr.w);
        }
    }
}

}  // namespace ultra
