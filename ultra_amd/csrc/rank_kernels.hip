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

}  // namespace ultra

extern "C" int32_t ultra_filtered_rank(const void *score, const int64_t *pos_index, const int64_t *known_ptr,
                                       const int64_t *known_index, int64_t batch, int64_t n_cand, int64_t *rank_out,
                                       int64_t *num_negative_out, void *stream) {
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
    hipLaunchKernelGGL(ultra::filtered_rank_kernel, grid, dim3(256), 0, s, (const float *)score, pos_index, known_ptr,
                       known_index, (long long)n_cand, reinterpret_cast<unsigned long long *>(rank_out),
                       reinterpret_cast<long long *>(num_negative_out));
    if (hipGetLastError() != hipSuccess) {
        ultra::set_error("filtered_rank_kernel launch failed");
        return ULTRA_ERR_HIP;
    }
    return ULTRA_OK;
}
