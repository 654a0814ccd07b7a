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
// One 1024-thread workgroup per row of the batch, no meeting of workgroups: the row is read as a flat int64 array with
// 16-byte loads (two consecutive elements per lane, fully coalesced, eight in flight per thread; an element's column is
// its index mod 3) and the verdicts are combined inside the workgroup.  History, measured in the captured forward at the
// benchmark point (2.8 MB): three 8-byte loads per candidate at a 24-byte lane stride, 16 workgroups per row meeting
// through device-scope atomics -- 15.6 us; 64 workgroups per row with coalesced loads -- 38.7 us: the 64 arrival tickets of
// a row are atomics on ONE address issued from eight XCDs, which the memory side serialises at ~0.5 us apiece.
// rel_first[b] (optional) = the row's relation as given, triples[b, 0, 2]: the relation model's query (models.py:20).
// cand[b][i] (optional; the training step, where a row is a few hundred candidates) = candidate node i of the converted row:
// its tail in a tail-prediction row, its head otherwise (new_t_index of base_nbfnet.py:84) -- a second pass over the row the
// workgroup has just read.
constexpr int PROLOGUE_THREADS = 1024;
__global__ void __launch_bounds__(PROLOGUE_THREADS) batch_prologue_kernel(const int64_t *__restrict__ batch, long long n_cand,
                                                                         long long num_direct_rel, int64_t *h0, int64_t *r0,
                                                                         int32_t *side, int32_t *valid, int64_t *rel_first,
                                                                         int64_t *cand) {
    __shared__ int lds_bad[3];
    const int b = blockIdx.x;
    const long long L = 3 * n_cand;
    const int64_t *row = batch + (long long)b * L;
    const int64_t fh = row[0], ft = row[1], fr = row[2];
    if (threadIdx.x < 3) lds_bad[threadIdx.x] = 0;
    int bad_h = 0, bad_t = 0, bad_r = 0;      // some element of the column differs from the row's first
    const auto check = [&](const unsigned c, const int64_t v) {     // (selects, no indexed private array)
        bad_h |= (c == 0u && v != fh);
        bad_t |= (c == 1u && v != ft);
        bad_r |= (c == 2u && v != fr);
    };
    long long lo = 0;
    const long long hi = L;
    if (((uintptr_t)row & 15u) != 0) {     // (an odd row start: one element by itself, the rest in aligned pairs)
        if (threadIdx.x == 0) check(0u, row[0]);
        lo = 1;
    }
    const long long n2 = (hi - lo) >> 1;
    const ulonglong2 *row2 = reinterpret_cast<const ulonglong2 *>(row + lo);
    // column of the first element of pair i: (lo + 2 i) mod 3, stepped without divisions: PROLOGUE_THREADS = 1024 = 1 mod 3
    unsigned c = (unsigned)((lo + 2 * (long long)(threadIdx.x % 3)) % 3);   // pair i = threadIdx.x
    long long i = threadIdx.x;
    for (; i + 7 * (long long)PROLOGUE_THREADS < n2; i += 8 * (long long)PROLOGUE_THREADS) {
        ulonglong2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = row2[i + u * (long long)PROLOGUE_THREADS];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            check(c, (int64_t)v[u].x);
            check(c == 2u ? 0u : c + 1u, (int64_t)v[u].y);
            c = (c + 2u) % 3u;      // pair index + 1024: element index + 2048 = + 2 mod 3
        }
    }
    for (; i < n2; i += PROLOGUE_THREADS) {
        const ulonglong2 v = row2[i];
        check(c, (int64_t)v.x);
        check(c == 2u ? 0u : c + 1u, (int64_t)v.y);
        c = (c + 2u) % 3u;
    }
    if (((hi - lo) & 1) && threadIdx.x == 0) check((unsigned)((hi - 1) % 3), row[hi - 1]);
    __syncthreads();
    if (__any(bad_h) && (threadIdx.x & 63) == 0) atomicOr(&lds_bad[0], 1);
    if (__any(bad_t) && (threadIdx.x & 63) == 0) atomicOr(&lds_bad[1], 1);
    if (__any(bad_r) && (threadIdx.x & 63) == 0) atomicOr(&lds_bad[2], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int any_h = lds_bad[0], any_t = lds_bad[1], any_r = lds_bad[2];
        const int tail_row = any_h == 0;     // base_nbfnet.py:82 is_t_neg
        side[b] = tail_row;
        h0[b] = tail_row ? fh : ft;
        r0[b] = tail_row ? fr : fr + num_direct_rel;
        valid[b] = ((any_h == 0 || any_t == 0) && any_r == 0) ? 1 : 0;
        if (rel_first) rel_first[b] = fr;
    }
    if (cand) {
        const int col = lds_bad[0] == 0 ? 1 : 0;
        for (long long c2 = threadIdx.x; c2 < n_cand; c2 += PROLOGUE_THREADS) cand[(long long)b * n_cand + c2] = row[3 * c2 + col];
    }
}

}  // namespace ultra

extern "C" int32_t ultra_batch_prologue_rows(const int64_t *batch, int64_t batch_size, int64_t n_cand, int64_t num_direct_rel,
                                             int64_t *h0, int64_t *r0, int32_t *side, int32_t *valid, int64_t *rel_first,
                                             int64_t *cand, void *stream) {
    ULTRA_DEVICE_SCOPE(stream, batch);
    if (!batch || !h0 || !r0 || !side || !valid || batch_size < 0 || n_cand <= 0) {
        ultra::set_error("ultra_batch_prologue: NULL operand or empty candidate set");
        return ULTRA_ERR_INVALID;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (batch_size == 0) return ULTRA_OK;
    (void)hipGetLastError();   // drop any stale error left by other users of the runtime
    hipLaunchKernelGGL(ultra::batch_prologue_kernel, dim3((unsigned)batch_size), dim3(ultra::PROLOGUE_THREADS), 0, s, batch, (long long)n_cand,
                       (long long)num_direct_rel, h0, r0, side, valid, rel_first, cand);
    if (hipGetLastError() != hipSuccess) {
        ultra::set_error("batch_prologue_kernel launch failed");
        return ULTRA_ERR_HIP;
    }
    return ULTRA_OK;
}

extern "C" int32_t ultra_batch_prologue(const int64_t *batch, int64_t batch_size, int64_t n_cand, int64_t num_direct_rel,
                                        int64_t *h0, int64_t *r0, int32_t *side, int32_t *valid, int64_t *rel_first,
                                        void *stream) {
    return ultra_batch_prologue_rows(batch, batch_size, n_cand, num_direct_rel, h0, r0, side, valid, rel_first, nullptr, stream);
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
