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
// PROLOGUE_SPLIT workgroups scan one row of the batch each (a single workgroup per row left 248 CUs idle).  They meet in
// `scratch` (4 ints per row: violations of h / t / r uniformity, arrival ticket): the last workgroup to arrive writes the
// row's results and puts the four ints back to zero, so the buffer needs zeroing once at allocation and never inside a
// captured graph.  The row is read as a flat int64 array with 16-byte loads (two consecutive elements per lane, fully
// coalesced; an element's column is its index mod 3) -- the first version read the three columns of a candidate with
// three 8-byte loads at a 24-byte lane stride: 15.6 us for the 2.8 MB of the benchmark batch.
constexpr int PROLOGUE_SPLIT = 64;
__global__ void __launch_bounds__(256) batch_prologue_kernel(const int64_t *__restrict__ batch, long long n_cand,
                                                            long long num_direct_rel, int64_t *h0, int64_t *r0,
                                                            int32_t *side, int32_t *valid, int32_t *scratch) {
    const int b = blockIdx.y, part = blockIdx.x;
    const long long L = 3 * n_cand;
    const int64_t *row = batch + (long long)b * L;
    const int64_t fh = row[0], ft = row[1], fr = row[2];
    int bad_h = 0, bad_t = 0, bad_r = 0;      // some element of the column differs from the row's first
    const auto check = [&](const unsigned c, const int64_t v) {     // (selects, no indexed private array)
        bad_h |= (c == 0u && v != fh);
        bad_t |= (c == 1u && v != ft);
        bad_r |= (c == 2u && v != fr);
    };
    const long long per = (((L + PROLOGUE_SPLIT - 1) / PROLOGUE_SPLIT) + 1) & ~1ll;
    long long lo = part * per, hi = lo + per < L ? lo + per : L;
    if (lo < hi) {
        if (((uintptr_t)(row + lo) & 15u) != 0) {     // (an odd row start: one element by itself, the rest in aligned pairs)
            if (threadIdx.x == 0) check((unsigned)(lo % 3), row[lo]);
            ++lo;
        }
        const long long n2 = (hi - lo) >> 1;
        const ulonglong2 *row2 = reinterpret_cast<const ulonglong2 *>(row + lo);
        const unsigned c_lo = (unsigned)(lo % 3);
        // 4 pairs (64 bytes) in flight per thread
        long long i = threadIdx.x;
        for (; i + 3 * (long long)blockDim.x < n2; i += 4 * (long long)blockDim.x) {
            ulonglong2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = row2[i + u * (long long)blockDim.x];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned c0 = (c_lo + 2u * (unsigned)((i + u * (long long)blockDim.x) % 3)) % 3u, c1 = c0 == 2u ? 0u : c0 + 1u;
                check(c0, (int64_t)v[u].x);
                check(c1, (int64_t)v[u].y);
            }
        }
        for (; i < n2; i += blockDim.x) {
            const ulonglong2 v = row2[i];
            const unsigned c0 = (c_lo + 2u * (unsigned)(i % 3)) % 3u, c1 = c0 == 2u ? 0u : c0 + 1u;
            check(c0, (int64_t)v.x);
            check(c1, (int64_t)v.y);
        }
        if (((hi - lo) & 1) && threadIdx.x == 0) check((unsigned)((hi - 1) % 3), row[hi - 1]);
    }
    const int same_h = !bad_h, same_t = !bad_t, same_r = !bad_r;
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
