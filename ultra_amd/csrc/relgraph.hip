// Relation graph of a knowledge graph on the GPU (reference: /root/reference/ultra/tasks.py:144-199).
//
// Nodes of the relation graph are the relation ids; an edge r1 -> r2 of type hh / tt / ht / th exists iff some entity is
// a head (h) or tail (t) of an r1 edge and of an r2 edge respectively.  The reference forms four sparse products
// E^T E over (entity, relation) incidence matrices and keeps only their sparsity pattern (tasks.py:186-189) -- 5.9 s on
// the CPU at FB15k237 shape.  Here the pattern is built as bit matrices:
//   1. incidence_bits_kernel : H[n] / T[n] = bit set of the relations entity n is head / tail of        (one atomicOr per edge)
//   2. pair_mark_kernel      : for every entity, every relation r1 in H[n]: A_hh[r1] |= H[n], A_ht[r1] |= T[n];
//                              every r1 in T[n]: A_tt[r1] |= T[n], A_th[r1] |= H[n]                       (word-wide ORs, skipped
//                              when the row already holds the bits)
//   3. row_count_kernel / emit_edges_kernel : popcounts per (type, row), then the edge list in the reference's order --
//      the hh, tt, ht, th blocks one after the other, each sorted by (row, col) -- so that the result equals the
//      reference's relation_graph.edge_index / edge_type element for element;
//   4. dense_order_adjacency_kernel : the same bit matrices as the byte adjacency of the reference-order layer kernel
//      (dense_order_layer.hip, plan.hpp `a_ex`): plan format straight from the device, no edge list in between.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "../../include/ultra_nbfnet.h"
#include "plan.hpp"
#include "device_scope.hpp"

namespace ultra {

__global__ void __launch_bounds__(256) incidence_bits_kernel(const int64_t *__restrict__ edge_index, const int64_t *__restrict__ edge_type,
                                                             long long num_edge, int words, uint32_t *__restrict__ hbits,
                                                             uint32_t *__restrict__ tbits) {
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < num_edge; e += (long long)gridDim.x * blockDim.x) {
        const long long h = edge_index[e], t = edge_index[num_edge + e], r = edge_type[e];
        const uint32_t bit = 1u << (r & 31);
        uint32_t *hw = hbits + h * words + (r >> 5), *tw = tbits + t * words + (r >> 5);
        if (!(*hw & bit)) atomicOr(hw, bit);
        if (!(*tw & bit)) atomicOr(tw, bit);
    }
}

// one wave per entity; lane w owns word w (+ 64, ...) of the relation bit sets
__global__ void __launch_bounds__(256) pair_mark_kernel(const uint32_t *__restrict__ hbits, const uint32_t *__restrict__ tbits,
                                                        long long num_node, int num_rel, int words, uint32_t *__restrict__ adj) {
    const int lane = threadIdx.x & 63;
    const long long wave = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6, nwave = ((long long)gridDim.x * blockDim.x) >> 6;
    const size_t mat = (size_t)num_rel * words;    // words per type matrix; adj = [hh | tt | ht | th]
    for (long long n = wave; n < num_node; n += nwave) {
        for (int w0 = 0; w0 < words; w0 += 64) {           // the words this lane ORs into the rows
            const int w = w0 + lane;
            const uint32_t hv = w < words ? hbits[n * words + w] : 0u, tv = w < words ? tbits[n * words + w] : 0u;
            for (int s = 0; s < words; ++s) {              // the relations (rows) of this entity, word by word (wave-uniform)
                uint32_t hs = hbits[n * words + s], ts = tbits[n * words + s];
                while (hs) {
                    const int r1 = 32 * s + __builtin_ctz(hs);
                    hs &= hs - 1;
                    if (w < words) {
                        uint32_t *a = adj + 0 * mat + (size_t)r1 * words + w, *b = adj + 2 * mat + (size_t)r1 * words + w;
                        if (hv & ~*a) atomicOr(a, hv);     // hh
                        if (tv & ~*b) atomicOr(b, tv);     // ht
                    }
                }
                while (ts) {
                    const int r1 = 32 * s + __builtin_ctz(ts);
                    ts &= ts - 1;
                    if (w < words) {
                        uint32_t *a = adj + 1 * mat + (size_t)r1 * words + w, *b = adj + 3 * mat + (size_t)r1 * words + w;
                        if (tv & ~*a) atomicOr(a, tv);     // tt
                        if (hv & ~*b) atomicOr(b, hv);     // th
                    }
                }
            }
        }
    }
}

__global__ void __launch_bounds__(256) row_count_kernel(const uint32_t *__restrict__ adj, int rows_total, int words,
                                                        int64_t *__restrict__ counts) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;     // (type, row) flattened
    if (r >= rows_total) return;
    long long c = 0;
    for (int w = 0; w < words; ++w) c += __builtin_popcount(adj[(size_t)r * words + w]);
    counts[r] = c;
}

// one 64-lane wave per (type, row): columns in ascending order
__global__ void __launch_bounds__(256) emit_edges_kernel(const uint32_t *__restrict__ adj, const int64_t *__restrict__ offsets,
                                                         int num_rel, int words, long long total, int64_t *__restrict__ edge_index,
                                                         int64_t *__restrict__ edge_type) {
    const int lane = threadIdx.x & 63;
    const int row_flat = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (row_flat >= 4 * num_rel) return;
    const int type = row_flat / num_rel, r1 = row_flat - type * num_rel;
    long long out = offsets[row_flat];
    for (int w0 = 0; w0 < words; w0 += 64) {
        const int w = w0 + lane;
        uint32_t v = w < words ? adj[(size_t)row_flat * words + w] : 0u;
        int c = __builtin_popcount(v), pre = c;     // inclusive prefix over the wave's 64 words
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(pre, off);
            if (lane >= off) pre += up;
        }
        long long pos = out + pre - c;
        while (v) {
            const int r2 = 32 * w + __builtin_ctz(v);
            v &= v - 1;
            edge_index[pos] = r1;
            edge_index[total + pos] = r2;
            edge_type[pos] = type;
            ++pos;
        }
        out += __shfl(pre, 63);
    }
}

// a_ex[row_tile16][col_chunk16][lane = (row % 16) + 16 type][col % 16]  (plan.hpp), one thread per 16-byte lane record
__global__ void __launch_bounds__(256) dense_order_adjacency_kernel(const uint32_t *__restrict__ adj, int num_rel, int words,
                                                                    uint4 *__restrict__ a_ex) {
    const int njc = (num_rel + 15) / 16, nrt = (num_rel + 15) / 16;
    const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (idx >= (long long)nrt * njc * 64) return;
    const int lane = (int)(idx & 63), jc = (int)((idx >> 6) % njc), rt = (int)((idx >> 6) / njc);
    const int row = 16 * rt + (lane & 15), type = lane >> 4;
    uint32_t half = 0;     // bits of columns 16 jc .. 16 jc + 15
    if (row < num_rel) half = (adj[((size_t)type * num_rel + row) * words + (jc >> 1)] >> (16 * (jc & 1))) & 0xffffu;
    uint32_t out[4];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        uint32_t wv = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) wv |= ((half >> (4 * q4 + b)) & 1u) << (8 * b);
        out[q4] = wv;
    }
    a_ex[idx] = make_uint4(out[0], out[1], out[2], out[3]);
}

}  // namespace ultra

using namespace ultra;

extern "C" {

int32_t ultra_relation_graph_bits(const int64_t *edge_index_dev, const int64_t *edge_type_dev, int64_t num_edge, int64_t num_node,
                                  int64_t num_relation, void *hbits_dev, void *tbits_dev, void *adj_dev, int64_t *row_counts_dev,
                                  void *stream) {
    ULTRA_DEVICE_SCOPE(stream, edge_index_dev);
    if (num_edge < 0 || num_node < 0 || num_relation <= 0 || (num_edge > 0 && (!edge_index_dev || !edge_type_dev)) || !hbits_dev ||
        !tbits_dev || !adj_dev || !row_counts_dev) {
        set_error("ultra_relation_graph_bits: bad argument");
        return ULTRA_ERR_INVALID;
    }
    if (num_relation > (1 << 20)) {
        set_error("ultra_relation_graph_bits: more than 2^20 relations are not supported");
        return ULTRA_ERR_UNSUPPORTED;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int words = (int)((num_relation + 31) / 32);
    (void)hipGetLastError();
    if (num_edge > 0)
        hipLaunchKernelGGL(incidence_bits_kernel, dim3((unsigned)std::min<int64_t>((num_edge + 255) / 256, 4096)), dim3(256), 0, s,
                           edge_index_dev, edge_type_dev, (long long)num_edge, words, (uint32_t *)hbits_dev, (uint32_t *)tbits_dev);
    if (num_node > 0)
        hipLaunchKernelGGL(pair_mark_kernel, dim3((unsigned)std::min<int64_t>((num_node + 3) / 4, 8192)), dim3(256), 0, s,
                           (const uint32_t *)hbits_dev, (const uint32_t *)tbits_dev, (long long)num_node, (int)num_relation, words,
                           (uint32_t *)adj_dev);
    const int rows_total = (int)(4 * num_relation);
    hipLaunchKernelGGL(row_count_kernel, dim3((rows_total + 255) / 256), dim3(256), 0, s, (const uint32_t *)adj_dev, rows_total, words,
                       row_counts_dev);
    if (hipGetLastError() != hipSuccess) {
        set_error("relation-graph kernels: launch failed");
        return ULTRA_ERR_HIP;
    }
    return ULTRA_OK;
}

int32_t ultra_relation_graph_emit(const void *adj_dev, const int64_t *row_offsets_dev, int64_t num_relation, int64_t total_edges,
                                  int64_t *edge_index_out_dev, int64_t *edge_type_out_dev, void *stream) {
    ULTRA_DEVICE_SCOPE(stream, adj_dev);
    if (!adj_dev || !row_offsets_dev || num_relation <= 0 || total_edges < 0 || (total_edges > 0 && (!edge_index_out_dev || !edge_type_out_dev))) {
        set_error("ultra_relation_graph_emit: bad argument");
        return ULTRA_ERR_INVALID;
    }
    if (total_edges == 0) return ULTRA_OK;
    const int words = (int)((num_relation + 31) / 32);
    const long long waves = 4 * num_relation;
    (void)hipGetLastError();
    hipLaunchKernelGGL(emit_edges_kernel, dim3((unsigned)((waves * 64 + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       (const uint32_t *)adj_dev, row_offsets_dev, (int)num_relation, words, (long long)total_edges, edge_index_out_dev,
                       edge_type_out_dev);
    if (hipGetLastError() != hipSuccess) {
        set_error("emit_edges_kernel: launch failed");
        return ULTRA_ERR_HIP;
    }
    return ULTRA_OK;
}

int32_t ultra_relation_graph_dense_adjacency(const void *adj_dev, int64_t num_relation, void *a_ex_out_dev, void *stream) {
    ULTRA_DEVICE_SCOPE(stream, adj_dev);
    if (!adj_dev || !a_ex_out_dev || num_relation <= 0) {
        set_error("ultra_relation_graph_dense_adjacency: bad argument");
        return ULTRA_ERR_INVALID;
    }
    const int words = (int)((num_relation + 31) / 32);
    const long long nt = (num_relation + 15) / 16;
    const long long total = nt * nt * 64;
    (void)hipGetLastError();
    hipLaunchKernelGGL(dense_order_adjacency_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), (const uint32_t *)adj_dev, (int)num_relation, words, (uint4 *)a_ex_out_dev);
    if (hipGetLastError() != hipSuccess) {
        set_error("dense_order_adjacency_kernel: launch failed");
        return ULTRA_ERR_HIP;
    }
    return ULTRA_OK;
}

}  // extern "C"
