// Backward of the rspmm on a LIST of output rows (rspmm_rows_kernels.hpp) as two GATHERS -- no atomics, a fixed summation order.
//
//   input_grad[o, c]    = base[o, c] + sum_{e : col_e == c, row_e listed for o} w_e * dBINARY/dx   * agg_grad[o, j(row_e)]   (rspmm.cpp:110-112)
//   relation_grad[o, t] =              sum_{e : type_e == t, row_e listed for o} w_e * dBINARY/drel * agg_grad[o, j(row_e)]   (rspmm.cpp:106-108)
//
// Round 5 scattered these sums from the listed rows with float atomics.  Listed rows share sources (hub nodes) and, far more,
// relation types (Zipf: the commonest type carries a sixth of the edges): thousands of adds per sample land on the same 64
// addresses, issued from eight XCDs, and the memory side retires same-address atomics one after the other -- 309 to 697 us per
// step at FB15k237's size for 76 k edges (profiles/r5_finetune_kernel_stats.csv, r6_01_timeline_eager.txt).
//
// Here every destination row has ONE owner that walks its own edges -- the edges leaving source c (the graph's edge list grouped
// by source), the edges of type t (grouped by type) -- and asks, per edge, whether the edge's aggregation row is listed:
//
//   slot[row][o % 8] = 1 + the first list position j of `row` in sample o (0: not listed)       uint16, 16 bytes per node
//
// built per step by rows_prepare_kernel, which also folds repeated list entries into their first position (agg_grad and the
// update's share `base` are summed over the repeats in list order).  One 16-byte load per edge answers for eight samples at once;
// ~ 2 % of the (edge, sample) pairs hit and gather two 256-byte rows.  A wave takes a segment of at most ROWS_BWD_SEG (sources) / ROWS_BWD_SEG_TYPE (types) edges of one
// owner; owners with one segment are written directly, the others leave partial
// rows that rows_bwd_combine_kernel adds in segment order.  The walk reads every edge record once (16 bytes each): 0.5 M edges
// at FB15k237's size, 2.2 M at YAGO3-10's -- the same for any number of listed rows.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "rspmm_kernels.hpp"

namespace ultra {

constexpr int ROWS_BWD_SEG = 128;          // edges per segment (one 16-lane group) of the input gradient: most sources have fewer, and are then written directly
constexpr int ROWS_BWD_SEG_TYPE = 256;     // ... of the relation gradient: few owners with long lists, every segment leaves a partial row
constexpr int ROWS_BWD_MAX_LIST = 1024;    // listed rows per sample served by this route (the prepare kernel compares them pairwise)

struct RowsPrepParams {
    const long long *rows;        // [n_outer][n_list]
    const float *agg_grad;        // [n_outer][n_list][64]
    const float *upd_grad;        // [n_outer][n_list][64] or NULL
    const long long *point_rows;  // [n_outer] or NULL
    uint16_t *slot;               // [n_chunk][num_row][8], zeroed by the caller
    float *gc, *uc;               // combined copies of agg_grad / upd_grad (rows of repeated entries other than the first: untouched)
    float *values_grad;           // [n_outer][64] or NULL: sum of agg_grad over the entries that list point_rows[o]
    long long num_row;
    int n_list;
};

// One workgroup per sample.  Repeated entries are rare (negatives are drawn with replacement from ~ N candidates), so the pairwise
// comparison runs without early exits (the compiler pipelines the LDS reads) and only entries that HAVE a repeat are summed by a scan.
__global__ void __launch_bounds__(1024) rows_prepare_kernel(const RowsPrepParams p) {
    __shared__ int lrow[ROWS_BWD_MAX_LIST];
    __shared__ int canon[ROWS_BWD_MAX_LIST];
    __shared__ int has_dup[ROWS_BWD_MAX_LIST];
    __shared__ int n_match;
    const int o = blockIdx.x, tid = threadIdx.x, n = p.n_list;
    const int pr = p.point_rows ? (int)p.point_rows[o] : -1;
    if (tid == 0) n_match = 0;
    for (int j = tid; j < n; j += blockDim.x) {
        lrow[j] = (int)p.rows[(long long)o * n + j];      // (node ids fit 31 bits: the plan's arrays are int32)
        has_dup[j] = 0;
    }
    __syncthreads();
    for (int j = tid; j < n; j += blockDim.x) {
        const int r = lrow[j];
        int first = j;
        for (int k = 0; k < j; ++k) first = (lrow[k] == r && k < first) ? k : first;
        canon[j] = first;
        if (first == j)
            p.slot[((long long)(o >> 3) * p.num_row + r) * 8 + (o & 7)] = (uint16_t)(j + 1);
        else
            has_dup[first] = 1;
        if (r == pr) atomicAdd(&n_match, 1);
    }
    __syncthreads();
    const int G = tid >> 4, l16 = tid & 15, ngroup = blockDim.x >> 4;
    for (int j = G; j < n; j += ngroup) {
        if (canon[j] != j) continue;
        const long long cell = ((long long)o * n + j) * 64 + 4 * l16;
        float4 g = *reinterpret_cast<const float4 *>(p.agg_grad + cell);
        float4 u = p.upd_grad ? *reinterpret_cast<const float4 *>(p.upd_grad + cell) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_dup[j]) {      // (rare: the sixteen lanes compare sixteen entries at a time, the matches are added in list order)
            const int shift = 16 * ((tid >> 4) & 3);
            for (int k0 = j + 1; k0 < n; k0 += 16) {
                const bool mine = k0 + l16 < n && canon[k0 + l16] == j;
                uint32_t found = (uint32_t)(__ballot(mine) >> shift) & 0xffffu;
                while (found) {
                    const int k = k0 + __builtin_ctz(found);
                    found &= found - 1;
                    const long long other = ((long long)o * n + k) * 64 + 4 * l16;
                    const float4 a = *reinterpret_cast<const float4 *>(p.agg_grad + other);
                    g.x += a.x, g.y += a.y, g.z += a.z, g.w += a.w;
                    if (p.upd_grad) {
                        const float4 b = *reinterpret_cast<const float4 *>(p.upd_grad + other);
                        u.x += b.x, u.y += b.y, u.z += b.z, u.w += b.w;
                    }
                }
            }
        }
        *reinterpret_cast<float4 *>(p.gc + cell) = g;
        if (p.upd_grad) *reinterpret_cast<float4 *>(p.uc + cell) = u;
    }
    if (p.values_grad && G == 0) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n_match > 0) {
            for (int k = 0; k < n; ++k) {
                if (lrow[k] != pr) continue;
                const float4 a = *reinterpret_cast<const float4 *>(p.agg_grad + ((long long)o * n + k) * 64 + 4 * l16);
                s.x += a.x, s.y += a.y, s.z += a.z, s.w += a.w;
            }
        }
        *reinterpret_cast<float4 *>(p.values_grad + (long long)o * 64 + 4 * l16) = s;
    }
}

struct RowsGatherParams {
    const int4 *rec;         // per edge, grouped by owner: {aggregation row, index of the OTHER operand's row, original edge id, 0}
    const int4 *seg;         // per segment: {owner, first edge, end edge, partial row or -1 (the owner's only segment: final write)}
    int n_seg;
    const uint16_t *slot;    // this chunk's [num_row][8]
    const float *keep;       // per-edge weights in original edge order, or NULL
    const float *gc;         // [n_outer][n_list][64] combined aggregate gradient
    const float *uc;         // combined update share (input gradient only), or NULL
    MatArg other;            // relation (input gradient) / input (relation gradient), fp32
    float *out;              // final destination, rows indexed by owner
    long long out_so, out_sr;
    float *partial;          // [n_partial][8][64]
    int n_list, o0, nb, mul_add;
};

// One 16-LANE GROUP per segment (four segments a wave, sixteen a workgroup).  The group looks 16 edges up at a time -- one record,
// one slot vector and one weight per lane: two dependent loads for the batch instead of two per edge; the next batch's are
// requested before this batch's hits are worked through.  Every (hit edge, sample) pair is then ONE step of a flat loop: request
// the pair's two 256-byte rows, add the PREVIOUS pair's product (its rows were requested a step earlier) into the group's
// accumulator rows in LDS (8 samples x 256 bytes: indexed by the sample, which registers could not be), move on.  The group's
// sixteen lanes hold the owner's whole 64-element row and write it; no cross-group reduction.
// (History, profiles/r6_experiments.txt: an edge per group and iteration -- 62 + 56 us at FB15k237's size; a wave per segment with
// 64-edge batches -- 35 + 103 us, 171 + 55 at YAGO3-10's; a group per segment with register accumulators and a branch per sample
// -- 56 + 106, 91 + 61: the four groups of a wave hit different samples, so a wave ran up to eight load-wait-add bodies per hit.)
template <bool INPUT_GRAD>
__global__ void __launch_bounds__(256) rows_bwd_gather_kernel(const RowsGatherParams p) {
    __shared__ float4 lacc[16][8][16];      // [group of the workgroup][sample][lane of the group]
    const int lane = threadIdx.x & 63, l16 = lane & 15, G = lane >> 4, gw = threadIdx.x >> 4;
    const int s = blockIdx.x * 16 + gw;
    const bool live = s < p.n_seg;
    const int4 sg = live ? p.seg[s] : make_int4(0, 0, 0, -1);
#pragma unroll
    for (int b = 0; b < 8; ++b) lacc[gw][b][l16] = make_float4(0.f, 0.f, 0.f, 0.f);      // (a lane only ever touches its own column)
    const float *ob = reinterpret_cast<const float *>(p.other.ptr) + (long long)p.o0 * p.other.stride_outer + 4 * l16;
    const float *gb = p.gc + (long long)p.o0 * p.n_list * 64 + 4 * l16;
    const auto fetch = [&](int base, int4 &r, uint4 &sv, float &w) {
        const int e = base + l16;
        r = make_int4(0, 0, 0, 0), sv = make_uint4(0u, 0u, 0u, 0u), w = 0.f;
        if (e < sg.z) {
            r = p.rec[e];
            sv = *reinterpret_cast<const uint4 *>(p.slot + (long long)r.x * 8);
            w = p.keep ? p.keep[r.z] : 1.f;
        }
    };
    // the pair whose rows are in flight
    bool have = false;
    float4 pg = make_float4(0.f, 0.f, 0.f, 0.f), pov = pg;
    float pww = 0.f;
    int pb = 0;
    const auto flush = [&]() {
        if (!have) return;
        const float4 v = make_float4(pww * pg.x, pww * pg.y, pww * pg.z, pww * pg.w);
        float4 a = lacc[gw][pb][l16];
        if (p.mul_add)          // TransE: both partial derivatives are 1
            a.x += v.x, a.y += v.y, a.z += v.z, a.w += v.w;
        else
            a.x += pov.x * v.x, a.y += pov.y * v.y, a.z += pov.z * v.z, a.w += pov.w * v.w;
        lacc[gw][pb][l16] = a;
        have = false;
    };
    int4 r, rn;
    uint4 sv, svn;
    float w, wn;
    fetch(sg.y, r, sv, w);
    // (trip counts differ between the four groups of a wave: the shuffles only ever read lanes of the own group, and every lane
    // of a group is in a loop while the group is)
    for (int base = sg.y; base < sg.z; base += 16) {
        fetch(base + 16, rn, svn, wn);
        const unsigned long long all = __ballot((sv.x | sv.y | sv.z | sv.w) != 0u && w != 0.f);
        uint32_t hits = (uint32_t)(all >> (16 * G)) & 0xffffu;
        while (hits) {
            const int from = 16 * G + __builtin_ctz(hits);
            hits &= hits - 1;
            const unsigned long long lo = (unsigned long long)(uint32_t)__shfl((int)sv.x, from) |
                                          ((unsigned long long)(uint32_t)__shfl((int)sv.y, from) << 32);
            const unsigned long long hi = (unsigned long long)(uint32_t)__shfl((int)sv.z, from) |
                                          ((unsigned long long)(uint32_t)__shfl((int)sv.w, from) << 32);
            const int other_row = __shfl(r.y, from);
            const float ww = __shfl(w, from);
            uint32_t samples = 0;
#pragma unroll
            for (int b = 0; b < 8; ++b) samples |= (((b < 4 ? lo : hi) >> (16 * (b & 3))) & 0xffffull) ? (1u << b) : 0u;
            while (samples) {
                const int b = __builtin_ctz(samples);
                samples &= samples - 1;
                const uint32_t j1 = (uint32_t)(((b < 4 ? lo : hi) >> (16 * (b & 3))) & 0xffffull);
                const float4 ng = *reinterpret_cast<const float4 *>(gb + ((long long)b * p.n_list + (j1 - 1)) * 64);
                float4 nov = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!p.mul_add)
                    nov = *reinterpret_cast<const float4 *>(ob + b * p.other.stride_outer + (long long)other_row * p.other.stride_row);
                flush();          // the previous pair: its rows were requested one step ago
                pg = ng, pov = nov, pww = ww, pb = b, have = true;
            }
        }
        r = rn, sv = svn, w = wn;
    }
    flush();
    if (!live) return;
    if (sg.w >= 0) {
#pragma unroll
        for (int b = 0; b < 8; ++b)
            if (b < p.nb) *reinterpret_cast<float4 *>(p.partial + ((long long)sg.w * 8 + b) * 64 + 4 * l16) = lacc[gw][b][l16];
        return;
    }
    uint4 own = make_uint4(0u, 0u, 0u, 0u);
    if (INPUT_GRAD && p.uc) own = *reinterpret_cast<const uint4 *>(p.slot + (long long)sg.x * 8);
    const uint32_t oword[4] = {own.x, own.y, own.z, own.w};
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        if (b >= p.nb) continue;
        float4 y = lacc[gw][b][l16];
        const uint32_t j1 = (oword[b >> 1] >> (16 * (b & 1))) & 0xffffu;
        if (j1) {
            const float4 u = *reinterpret_cast<const float4 *>(p.uc + ((long long)(p.o0 + b) * p.n_list + (j1 - 1)) * 64 + 4 * l16);
            y.x += u.x, y.y += u.y, y.z += u.z, y.w += u.w;
        }
        *reinterpret_cast<float4 *>(p.out + (long long)(p.o0 + b) * p.out_so + (long long)sg.x * p.out_sr + 4 * l16) = y;
    }
}

struct RowsCombineParams {
    const int4 *multi;       // per owner with several segments: {owner, first partial row, number of partial rows, 0}
    int n_multi;
    const float *partial;
    const uint16_t *slot;
    const float *uc;         // or NULL
    float *out;
    long long out_so, out_sr;
    int n_list, o0, nb;
};

// One workgroup per owner: 8 slices x (8 samples x 16 lanes).  Slice q adds the owner's partial rows q, q + 8, ... (four loads in
// flight), the slices are folded 0 + 1 + ... + 7 through LDS, then the update's share: a fixed order.  (A commonest relation type
// owns hundreds of partial rows: one serial chain over them took 91 us at FB15k237's size, 291 at YAGO3-10's.)
__global__ void __launch_bounds__(1024) rows_bwd_combine_kernel(const RowsCombineParams p) {
    __shared__ __attribute__((aligned(16))) float fold[8][8 * 64];
    const int q = threadIdx.x >> 7, b = (threadIdx.x >> 4) & 7, l16 = threadIdx.x & 15;
    const int4 m = p.multi[blockIdx.x];
    float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
    if (b < p.nb) {
        const float *src = p.partial + ((long long)m.y * 8 + b) * 64 + 4 * l16;
        int k = q;
        for (; k + 24 < m.z; k += 32) {
            const float4 a0 = *reinterpret_cast<const float4 *>(src + (long long)k * 512);
            const float4 a1 = *reinterpret_cast<const float4 *>(src + (long long)(k + 8) * 512);
            const float4 a2 = *reinterpret_cast<const float4 *>(src + (long long)(k + 16) * 512);
            const float4 a3 = *reinterpret_cast<const float4 *>(src + (long long)(k + 24) * 512);
            y.x = (((y.x + a0.x) + a1.x) + a2.x) + a3.x, y.y = (((y.y + a0.y) + a1.y) + a2.y) + a3.y;
            y.z = (((y.z + a0.z) + a1.z) + a2.z) + a3.z, y.w = (((y.w + a0.w) + a1.w) + a2.w) + a3.w;
        }
        for (; k < m.z; k += 8) {
            const float4 a = *reinterpret_cast<const float4 *>(src + (long long)k * 512);
            y.x += a.x, y.y += a.y, y.z += a.z, y.w += a.w;
        }
    }
    *reinterpret_cast<float4 *>(&fold[q][b * 64 + 4 * l16]) = y;
    __syncthreads();
    if (q != 0 || b >= p.nb) return;
#pragma unroll
    for (int s = 1; s < 8; ++s) {
        const float4 a = *reinterpret_cast<const float4 *>(&fold[s][b * 64 + 4 * l16]);
        y.x += a.x, y.y += a.y, y.z += a.z, y.w += a.w;
    }
    if (p.uc) {
        const uint32_t j1 = p.slot[(long long)m.x * 8 + b];
        if (j1) {
            const float4 u = *reinterpret_cast<const float4 *>(p.uc + ((long long)(p.o0 + b) * p.n_list + (j1 - 1)) * 64 + 4 * l16);
            y.x += u.x, y.y += u.y, y.z += u.z, y.w += u.w;
        }
    }
    *reinterpret_cast<float4 *>(p.out + (long long)(p.o0 + b) * p.out_so + (long long)m.x * p.out_sr + 4 * l16) = y;
}

__global__ void __launch_bounds__(256) zero_words_kernel(uint4 *dst, long long n16) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256)
        dst[i] = make_uint4(0u, 0u, 0u, 0u);
}

}  // namespace ultra
