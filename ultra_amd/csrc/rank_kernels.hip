// Filtered ranking without the (batch, N) boolean mask (reference: ultra/tasks.py:94-141).
//
//   rank[q]    = 1 + #{t : mask[q, t] and score[q, pos_q] <= score[q, t]}            (tasks.py:133-141)
//   mask[q, t] = t is not a known true answer of query q and t != pos_q              (tasks.py:94-130)
//
// computed as  1 + #{t : pos <= score[t]}  -  #{t in known(q) : pos <= score[t]}  where known(q) is the
// de-duplicated list of true answers INCLUDING the positive itself (a ragged int64 list per query).
// Integer counting: bit-exact against the reference for any score tensor (ties count against the positive).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "../../include/ultra_rspmm.h"
#include "plan.hpp"
#include "device_scope.hpp"

namespace ultra {

constexpr int RANK_CHUNK = 4096;

__global__ void __launch_bounds__(256) filtered_rank_kernel(const float *__restrict__ score, const int64_t *__restrict__ pos,
                                                            const int64_t *__restrict__ known_ptr,
                                                            const int64_t *__restrict__ known_index, long long n_cand,
                                                            unsigned long long *rank, long long *num_negative) {
    const int q = blockIdx.y;
    const float *row = score + (long long)q * n_cand;
    const float pos_score = row[pos[q]];
    const long long lo = (long long)blockIdx.x * RANK_CHUNK;
    const long long hi = lo + RANK_CHUNK < n_cand ? lo + RANK_CHUNK : n_cand;
    long long count = 0;
    for (long long t = lo + threadIdx.x; t < hi; t += blockDim.x) count += (pos_score <= row[t]) ? 1 : 0;
    if (blockIdx.x == 0) {
        // the known answers (and the positive itself) do not count; the leading "+ 1" of the rank lives here too
        const long long k0 = known_ptr[q], k1 = known_ptr[q + 1];
        for (long long k = k0 + threadIdx.x; k < k1; k += blockDim.x) count -= (pos_score <= row[known_index[k]]) ? 1 : 0;
        if (threadIdx.x == 0) {
            count += 1;
            num_negative[q] = n_cand - (k1 - k0);
        }
    }
    // wave reduce, then one atomic per wave (integer: order independent)
    for (int off = 32; off > 0; off >>= 1) count += __shfl_down(count, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(rank + q, (unsigned long long)count);
}

// Batch prologue of EntityNBFNet.forward (models.py:190-197 + base_nbfnet.py:79-86) in one pass over the
// (batch, n_cand, 3) int64 triples [h, t, r]:
//   side[b]  = 1 if every candidate of row b shares the head (a tail-prediction row), else 0 (head-prediction row:
//              the propagation then starts from the shared TAIL with the inverse relation r + num_direct_rel);
//   h0[b], r0[b] = source node and query relation of the row after that conversion;
//   valid[b] = the row really shares its source node and its relation (the reference's two asserts), per row:
//              no initialisation pass / memset node is needed (hipGraph friendly).
// PROLOGUE_SPLIT workgroups scan one row of the batch each (a single workgroup per row left 248 CUs idle: 7-11 us for
// 2.8 MB).  They meet in `scratch` (4 ints per row: violations of h / t / r uniformity, arrival ticket): the last
// workgroup to arrive writes the row's results and puts the four ints back to zero, so the buffer needs zeroing once at
// allocation and never inside a captured graph.
constexpr int PROLOGUE_SPLIT = 16;
__global__ void __launch_bounds__(256) batch_prologue_kernel(const int64_t *__restrict__ batch, long long n_cand,
                                                            long long num_direct_rel, int64_t *h0, int64_t *r0,
                                                            int32_t *side, int32_t *valid, int32_t *scratch) {
    const int b = blockIdx.y, part = blockIdx.x;
    const int64_t *row = batch + (long long)b * n_cand * 3;
    const int64_t fh = row[0], ft = row[1], fr = row[2];
    int same_h = 1, same_t = 1, same_r = 1;
    const long long per = (n_cand + PROLOGUE_SPLIT - 1) / PROLOGUE_SPLIT;
    const long long lo = part * per, hi = lo + per < n_cand ? lo + per : n_cand;
    // 4 candidates (12 independent loads) in flight per thread
    long long i = lo + threadIdx.x;
    for (; i + 3 * (long long)blockDim.x < hi; i += 4 * (long long)blockDim.x) {
        int64_t v[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < 3; ++c) v[u][c] = row[3 * (i + u * (long long)blockDim.x) + c];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            same_h &= (v[u][0] == fh);
            same_t &= (v[u][1] == ft);
            same_r &= (v[u][2] == fr);
        }
    }
    for (; i < hi; i += blockDim.x) {
        same_h &= (row[3 * i] == fh);
        same_t &= (row[3 * i + 1] == ft);
        same_r &= (row[3 * i + 2] == fr);
    }
    int32_t *sc = scratch + 4 * b;
    const bool wave_h = __all(same_h), wave_t = __all(same_t), wave_r = __all(same_r);
    if ((threadIdx.x & 63) == 0) {
        if (!wave_h) atomicAdd(&sc[0], 1);
        if (!wave_t) atomicAdd(&sc[1], 1);
        if (!wave_r) atomicAdd(&sc[2], 1);
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(&sc[3], 1) == PROLOGUE_SPLIT - 1) {
        const int bad_h = atomicAdd(&sc[0], 0), bad_t = atomicAdd(&sc[1], 0), bad_r = atomicAdd(&sc[2], 0);
        const int tail_row = bad_h == 0;     // base_nbfnet.py:82 is_t_neg
        side[b] = tail_row;
        h0[b] = tail_row ? fh : ft;
        r0[b] = tail_row ? fr : fr + num_direct_rel;
        valid[b] = ((bad_h == 0 || bad_t == 0) && bad_r == 0) ? 1 : 0;
        sc[0] = sc[1] = sc[2] = sc[3] = 0;      // ready for the next launch (or graph replay)
    }
}

}  // namespace ultra

extern "C" int32_t ultra_batch_prologue(const int64_t *batch, int64_t batch_size, int64_t n_cand, int64_t num_direct_rel,
                                        int64_t *h0, int64_t *r0, int32_t *side, int32_t *valid, int32_t *scratch,
                                        void *stream) {
    ULTRA_DEVICE_SCOPE(stream, batch);
    if (!batch || !h0 || !r0 || !side || !valid || !scratch || batch_size < 0 || n_cand <= 0) {
        ultra::set_error("ultra_batch_prologue: NULL operand or empty candidate set");
        return ULTRA_ERR_INVALID;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (batch_size == 0) return ULTRA_OK;
    (void)hipGetLastError();   // drop any stale error left by other users of the runtime
    hipLaunchKernelGGL(ultra::batch_prologue_kernel, dim3(ultra::PROLOGUE_SPLIT, (unsigned)batch_size), dim3(256), 0, s, batch, (long long)n_cand,
                       (long long)num_direct_rel, h0, r0, side, valid, scratch);
    if (hipGetLastError() != hipSuccess) {
        ultra::set_error("batch_prologue_kernel launch failed");
        return ULTRA_ERR_HIP;
    }
    return ULTRA_OK;
}

extern "C" int32_t ultra_filtered_rank(const void *score, const int64_t *pos_index, const int64_t *known_ptr,
                                       const int64_t *known_index, int64_t batch, int64_t n_cand, int64_t *rank_out,
                                       int64_t *num_negative_out, void *stream) {
    ULTRA_DEVICE_SCOPE(stream, score);
    if (!score || !pos_index || !known_ptr || !rank_out || !num_negative_out || batch < 0 || n_cand <= 0) {
        ultra::set_error("ultra_filtered_rank: NULL operand or empty candidate set");
        return ULTRA_ERR_INVALID;
    }
    if (batch == 0) return ULTRA_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (hipMemsetAsync(rank_out, 0, sizeof(int64_t) * (size_t)batch, s) != hipSuccess) {
        ultra::set_error("ultra_filtered_rank: hipMemsetAsync failed");
        return ULTRA_ERR_HIP;
    }
    const dim3 grid((unsigned)((n_cand + ultra::RANK_CHUNK - 1) / ultra::RANK_CHUNK), (unsigned)batch);
    (void)hipGetLastError();   // drop any stale error left by other users of the runtime
    hipLaunchKernelGGL(ultra::filtered_rank_kernel, grid, dim3(256), 0, s, (const float *)score, pos_index, known_ptr,
                       known_index, (long long)n_cand, reinterpret_cast<unsigned long long *>(rank_out),
                       reinterpret_cast<long long *>(num_negative_out));
    if (hipGetLastError() != hipSuccess) {
        ultra::set_error("filtered_rank_kernel launch failed");
        return ULTRA_ERR_HIP;
    }
    return ULTRA_OK;
}
