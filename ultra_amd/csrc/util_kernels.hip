// Measurement helper: a plain 16-byte-per-lane streaming copy used to (a) measure the achievable HBM
// copy ceiling next to the rspmm numbers and (b) calibrate the FETCH_SIZE / WRITE_SIZE counters on a
// known byte count (MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced reads on gfx950).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "../../include/ultra_rspmm.h"
#include "plan.hpp"
#include "device_scope.hpp"

namespace ultra {

// Streaming copy, 16 B per lane: eight loads in flight per lane before the first store (one wave keeps 8 KiB moving), the
// grid strides in 8-chunk blocks so that a wave's eight requests are eight consecutive 1-KiB lines.
__global__ void __launch_bounds__(256) stream_copy_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst,
                                                          long long n16) {
    constexpr int U = 8;
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 *s4 = reinterpret_cast<const f4 *>(src);
    f4 *d4 = reinterpret_cast<f4 *>(dst);
    const long long stride = (long long)gridDim.x * blockDim.x * U;
    long long base = ((long long)blockIdx.x * blockDim.x + (threadIdx.x & ~63)) * U + (threadIdx.x & 63);
    for (; base + (U - 1) * 64 < n16; base += stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(s4 + base + u * 64);
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u], d4 + base + u * 64);
    }
    for (int u = 0; u < U; ++u)       // ragged tail of the last block
        if (base + u * 64 < n16) dst[base + u * 64] = src[base + u * 64];
}

// out[b, n, :] = (n == rows[b]) ? values[b, :] (or 1 when values == NULL) : 0 -- the NBFNet boundary condition
// (models.py:59-66, 135-141: zeros + scatter_add of one row per sample) in a single pass, no memset node.
// With `table`: values[b] = table[b, pick[b], :] (the query relation's representation, models.py:131-133), also written
// to values_out[b] for the readout.
__global__ void __launch_bounds__(256) onehot_rows_kernel(float4 *__restrict__ out, const int64_t *__restrict__ rows,
                                                          const float4 *__restrict__ values, const float4 *__restrict__ table,
                                                          const int64_t *__restrict__ pick, long long table_rows,
                                                          float4 *__restrict__ values_out, long long num_node, int dim4,
                                                          long long total4, const float *__restrict__ w1,
                                                          const float *__restrict__ b1, float *__restrict__ qbias_out,
                                                          int batch) {
    if (!out && table) {
        // gather-only launch (one block per sample): query row and the readout's per-sample bias
        const int b = blockIdx.x, dim = 4 * dim4;
        long long t = pick[b];
        t = t < 0 ? 0 : (t >= table_rows ? table_rows - 1 : t);
        const float4 *qr = table + ((long long)b * table_rows + t) * dim4;
        for (int d = threadIdx.x; d < dim4; d += blockDim.x) values_out[b * dim4 + d] = qr[d];
        if (qbias_out) {
            for (int f = threadIdx.x; f < 2 * dim; f += blockDim.x) {
                const float4 *wr = reinterpret_cast<const float4 *>(w1 + (long long)f * 2 * dim + dim);
                float acc = b1[f];
                for (int k = 0; k < dim4; ++k) {
                    const float4 w = wr[k], q = qr[k];
                    acc += w.x * q.x;
                    acc += w.y * q.y;
                    acc += w.z * q.z;
                    acc += w.w * q.w;
                }
                qbias_out[b * 2 * dim + f] = acc;
            }
        }
        return;
    }
    if (table && blockIdx.x == gridDim.x - 1) {
        // the gathered query rows themselves (read back by the layers and the readout)
        for (int idx = threadIdx.x; idx < batch * dim4; idx += blockDim.x) {
            const int b = idx / dim4, d = idx % dim4;
            long long t = pick[b];   // (an out-of-range relation id reads a valid row instead of faulting)
            t = t < 0 ? 0 : (t >= table_rows ? table_rows - 1 : t);
            values_out[idx] = table[((long long)b * table_rows + t) * dim4 + d];
        }
    }
    if (qbias_out && blockIdx.x == gridDim.x - 1) {
        // readout preamble riding on this launch (models.py:166-170 concatenates the query to every node feature; its half
        // of mlp.0 is a per-sample constant): qbias[b, f] = b1[f] + sum_k w1[f, dim + k] * query[b, k], dim = 4 dim4 = 64
        const int dim = 4 * dim4;
        for (int idx = threadIdx.x; idx < batch * 2 * dim; idx += blockDim.x) {
            const int b = idx / (2 * dim), f = idx % (2 * dim);
            long long t = pick[b];
            t = t < 0 ? 0 : (t >= table_rows ? table_rows - 1 : t);
            const float4 *qr = table + ((long long)b * table_rows + t) * dim4;
            const float4 *wr = reinterpret_cast<const float4 *>(w1 + (long long)f * 2 * dim + dim);
            float acc = b1[f];
            for (int k = 0; k < dim4; ++k) {
                const float4 w = wr[k], q = qr[k];
                acc += w.x * q.x;
                acc += w.y * q.y;
                acc += w.z * q.z;
                acc += w.w * q.w;
            }
            qbias_out[idx] = acc;
        }
    }
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const int d = (int)(i % dim4);
        const long long r = i / dim4;
        const long long n = r % num_node, b = r / num_node;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n == rows[b]) {
            if (table) {
                long long t = pick[b];
                t = t < 0 ? 0 : (t >= table_rows ? table_rows - 1 : t);
                v = table[(b * table_rows + t) * dim4 + d];
            } else {
                v = values ? values[b * dim4 + d] : make_float4(1.f, 1.f, 1.f, 1.f);
            }
        }
        out[i] = v;
    }
}

// keep[e] = 0 where edge e = (head, tail[, type]) equals one of the listed (easy) edges, else 1: the 0/1 vector that
// replaces the graph copy of base_nbfnet.py:54-77 (edge_match + boolean indexing) -- the graph, and with it the plan,
// stays static across training batches.  Edges are compared through the mixed-radix key (head * N + tail) * R + type the
// reference's edge_match uses (tasks.py:7-39); the easy keys arrive sorted and sit in LDS, one binary search per edge.
constexpr int EASY_MAX = 8192;
__global__ void __launch_bounds__(256) edge_keep_mask_kernel(const long long *head, const long long *tail, const long long *type,
                                                             long long num_edge, const long long *easy_key, int n_easy,
                                                             long long num_node, long long num_rel, float *keep) {
    __shared__ long long lds_key[EASY_MAX];
    for (int i = threadIdx.x; i < n_easy; i += blockDim.x) lds_key[i] = easy_key[i];
    __syncthreads();
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < num_edge; e += (long long)gridDim.x * blockDim.x) {
        long long key = head[e] * num_node + tail[e];
        if (type) key = key * num_rel + type[e];
        int lo = 0, hi = n_easy;          // first position with lds_key[pos] >= key
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (lds_key[mid] < key)
                lo = mid + 1;
            else
                hi = mid;
        }
        keep[e] = (lo < n_easy && lds_key[lo] == key) ? 0.f : 1.f;
    }
}

// edge_keep_mask_kernel with the list built in place: every workgroup hashes the batch's 2 n_triple keys ((h, t, r) and
// (t, h, r + inverse_offset), base_nbfnet.py:57-59) into an open-addressing table in LDS (16,384 slots for <= 8,192 keys) and
// probes it once per edge -- no concatenations, no key arithmetic in torch, no sort: one launch instead of ~ 20.
constexpr int EASY_SLOTS = 16384;
constexpr int EASY_THREADS = 1024;
__device__ __forceinline__ unsigned easy_slot(const unsigned long long key) {
    return (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 50);      // top 14 bits
}
__global__ void __launch_bounds__(EASY_THREADS) easy_edge_keep_kernel(const long long *head, const long long *tail, const long long *type,
                                                                     long long num_edge, const long long *qh, const long long *qt,
                                                                     const long long *qr, int n_triple, long long stride,
                                                                     long long num_node, long long num_rel, long long inverse_offset,
                                                                     float *keep) {
    __shared__ unsigned long long table[EASY_SLOTS];
    constexpr unsigned long long EMPTY = ~0ull;
    for (int i = threadIdx.x; i < EASY_SLOTS; i += EASY_THREADS) table[i] = EMPTY;
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * n_triple; i += EASY_THREADS) {
        const int j = i < n_triple ? i : i - n_triple;
        long long a = qh[j * stride], b = qt[j * stride];
        long long key;
        if (i >= n_triple) {
            const long long s = a;
            a = b;
            b = s;
        }
        key = a * num_node + b;
        if (type) key = key * num_rel + qr[j * stride] + (i >= n_triple ? inverse_offset : 0);
        if (key < 0) continue;               // (no edge of a valid graph has a negative key)
        unsigned slot = easy_slot((unsigned long long)key);
        while (true) {
            const unsigned long long old = atomicCAS(&table[slot], EMPTY, (unsigned long long)key);
            if (old == EMPTY || old == (unsigned long long)key) break;
            slot = (slot + 1) & (EASY_SLOTS - 1);
        }
    }
    __syncthreads();
    for (long long e = blockIdx.x * (long long)EASY_THREADS + threadIdx.x; e < num_edge; e += (long long)gridDim.x * EASY_THREADS) {
        long long key = head[e] * num_node + tail[e];
        if (type) key = key * num_rel + type[e];
        unsigned slot = easy_slot((unsigned long long)key);
        float k = 1.f;
        while (true) {
            const unsigned long long seen = table[slot];
            if (seen == EMPTY) break;
            if (seen == (unsigned long long)key) {
                k = 0.f;
                break;
            }
            slot = (slot + 1) & (EASY_SLOTS - 1);
        }
        keep[e] = k;
    }
}

}  // namespace ultra

extern "C" int32_t ultra_easy_edge_keep(const int64_t *head, const int64_t *tail, const int64_t *type, int64_t num_edge,
                                        const int64_t *h, const int64_t *t, const int64_t *r, int64_t n_triple, int64_t stride,
                                        int64_t num_node, int64_t num_rel, int64_t inverse_offset, void *keep, void *stream) {
    ULTRA_DEVICE_SCOPE(stream, keep);
    if (!head || !tail || !keep || num_edge < 0 || n_triple < 0 || (n_triple > 0 && (!h || !t)) || stride <= 0 || num_node <= 0 ||
        (type && (num_rel <= 0 || !r))) {
        ultra::set_error("ultra_easy_edge_keep: NULL operand or empty key space");
        return ULTRA_ERR_INVALID;
    }
    if (2 * n_triple > ultra::EASY_MAX) {
        ultra::set_error("ultra_easy_edge_keep: more than 8192 edges to match");
        return ULTRA_ERR_UNSUPPORTED;
    }
    if (num_edge == 0) return ULTRA_OK;
    (void)hipGetLastError();   // drop any stale error left by other users of the runtime
    const long long want = (num_edge + ultra::EASY_THREADS - 1) / ultra::EASY_THREADS;
    const unsigned blocks = (unsigned)(want < 256 ? want : 256);
    hipLaunchKernelGGL(ultra::easy_edge_keep_kernel, dim3(blocks), dim3(ultra::EASY_THREADS), 0, reinterpret_cast<hipStream_t>(stream),
                       (const long long *)head, (const long long *)tail, (const long long *)type, (long long)num_edge,
                       (const long long *)h, (const long long *)t, (const long long *)r, (int)n_triple, (long long)stride,
                       (long long)num_node, (long long)num_rel, (long long)inverse_offset, (float *)keep);
    if (hipGetLastError() != hipSuccess) {
        ultra::set_error("easy_edge_keep_kernel launch failed");
        return ULTRA_ERR_HIP;
    }
    return ULTRA_OK;
}

extern "C" int32_t ultra_onehot_rows(void *out, const int64_t *rows, const void *values, int64_t batch, int64_t num_node,
                                     int64_t dim, void *stream) {
    ULTRA_DEVICE_SCOPE(stream, out);
    if (!out || !rows || batch < 0 || num_node < 0 || dim <= 0 || (dim & 3)) {
        ultra::set_error("ultra_onehot_rows: NULL operand or dim not a multiple of 4");
        return ULTRA_ERR_INVALID;
    }
    const long long total4 = (long long)batch * num_node * (dim / 4);
    if (total4 == 0) return ULTRA_OK;
    const int grid = (int)((total4 + 255) / 256 < 2048 ? (total4 + 255) / 256 : 2048);
    (void)hipGetLastError();   // drop any stale error left by other users of the runtime
    hipLaunchKernelGGL(ultra::onehot_rows_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       (float4 *)out, rows, (const float4 *)values, (const float4 *)nullptr, (const int64_t *)nullptr, 0ll,
                       (float4 *)nullptr, (long long)num_node, (int)(dim / 4), total4, (const float *)nullptr,
                       (const float *)nullptr, (float *)nullptr, 0);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ultra::set_error(std::string("onehot_rows_kernel launch: ") + hipGetErrorString(e));
        return ULTRA_ERR_HIP;
    }
    return ULTRA_OK;
}

extern "C" int32_t ultra_query_boundary(void *out, void *query_out, const int64_t *rows, const void *table,
                                        const int64_t *pick, int64_t batch, int64_t num_node, int64_t table_rows, int64_t dim,
                                        const void *w1, const void *b1, void *qbias_out, void *stream) {
    ULTRA_DEVICE_SCOPE(stream, out);
    if (!query_out || !rows || !table || !pick || batch < 0 || num_node <= 0 || table_rows <= 0 || dim <= 0 || (dim & 3)) {
        ultra::set_error("ultra_query_boundary: NULL operand, empty graph or dim not a multiple of 4");
        return ULTRA_ERR_INVALID;
    }
    if (batch == 0) return ULTRA_OK;
    // out == NULL: only the gathers (query_out, qbias_out) -- the boundary stays in closed form (ultra_rspmm_forward_point)
    const long long total4 = out ? (long long)batch * num_node * (dim / 4) : 0;
    const int grid = out ? (int)((total4 + 255) / 256 < 2048 ? (total4 + 255) / 256 : 2048) : (int)batch;
    (void)hipGetLastError();   // drop any stale error left by other users of the runtime
    hipLaunchKernelGGL(ultra::onehot_rows_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       (float4 *)out, rows, (const float4 *)nullptr, (const float4 *)table, pick, (long long)table_rows,
                       (float4 *)query_out, (long long)num_node, (int)(dim / 4), total4, (const float *)w1, (const float *)b1,
                       (float *)((w1 && b1) ? qbias_out : nullptr), (int)batch);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ultra::set_error(std::string("onehot_rows_kernel launch: ") + hipGetErrorString(e));
        return ULTRA_ERR_HIP;
    }
    return ULTRA_OK;
}

extern "C" int32_t ultra_stream_copy(void *dst, const void *src, int64_t bytes, void *stream) {
    ULTRA_DEVICE_SCOPE(stream, dst);
    if (!dst || !src || bytes < 0 || (bytes & 15)) {
        ultra::set_error("ultra_stream_copy: NULL pointer or size not a multiple of 16");
        return ULTRA_ERR_INVALID;
    }
    if (bytes == 0) return ULTRA_OK;
    (void)hipGetLastError();   // drop any stale error left by other users of the runtime
    const long long n16 = bytes / 16;
    const unsigned blocks = (unsigned)((n16 + 255) / 256 < 2048 ? (n16 + 255) / 256 : 2048);
    hipLaunchKernelGGL(ultra::stream_copy_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       (const float4 *)src, (float4 *)dst, n16);
    if (hipGetLastError() != hipSuccess) {
        ultra::set_error("stream_copy_kernel launch failed");
        return ULTRA_ERR_HIP;
    }
    return ULTRA_OK;
}

extern "C" int32_t ultra_edge_keep_mask(const int64_t *head, const int64_t *tail, const int64_t *type, int64_t num_edge,
                                        const int64_t *easy_key_sorted, int64_t n_easy, int64_t num_node, int64_t num_rel,
                                        void *keep, void *stream) {
    ULTRA_DEVICE_SCOPE(stream, keep);
    if (!head || !tail || !keep || num_edge < 0 || n_easy < 0 || (n_easy > 0 && !easy_key_sorted) || num_node <= 0 ||
        (type && num_rel <= 0)) {
        ultra::set_error("ultra_edge_keep_mask: NULL operand or empty key space");
        return ULTRA_ERR_INVALID;
    }
    if (n_easy > ultra::EASY_MAX) {
        ultra::set_error("ultra_edge_keep_mask: more than 8192 edges to match");
        return ULTRA_ERR_UNSUPPORTED;
    }
    if (num_edge == 0) return ULTRA_OK;
    (void)hipGetLastError();   // drop any stale error left by other users of the runtime
    const unsigned blocks = (unsigned)((num_edge + 255) / 256 < 2048 ? (num_edge + 255) / 256 : 2048);
    hipLaunchKernelGGL(ultra::edge_keep_mask_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       (const long long *)head, (const long long *)tail, (const long long *)type, (long long)num_edge,
                       (const long long *)easy_key_sorted, (int)n_easy, (long long)num_node, (long long)num_rel, (float *)keep);
    if (hipGetLastError() != hipSuccess) {
        ultra::set_error("edge_keep_mask_kernel launch failed");
        return ULTRA_ERR_HIP;
    }
    return ULTRA_OK;
}
