// Strict negative sampling (tasks.py:42-76 with strict=True) without the (batch, N) masks and without a host round trip.
//
// The reference builds, per positive, the boolean mask of valid candidates (every entity that is not a known true answer of the
// query, tasks.py:94-130), lists the valid ids with nonzero() -- a host synchronisation, and a variable-size result -- and picks
// candidate[floor(rand * count)] for num_negative uniform draws.  The valid ids are ascending, so the pick is "the idx-th entity
// that is not excluded": with the query's known answers as a SORTED list (a slice of the static graph's sorted unique
// (anchor, relation, answer) keys) that is a bisection over the entity ids, no mask, no list of candidates.
//
// One workgroup per positive; every thread answers some of the draws.  The draws themselves come from torch.rand in the caller
// -- the same generator calls, in the same order and shapes, as the reference's -- so the sampled ids are the reference's.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "../../include/ultra_nbfnet.h"
#include "../../include/ultra_rspmm.h"
#include "plan.hpp"
#include "device_scope.hpp"

namespace ultra {

// first index in [lo, hi) whose key is >= v
__device__ __forceinline__ long long lower_bound_keys(const long long *keys, long long lo, long long hi, const long long v) {
    while (lo < hi) {
        const long long mid = lo + ((hi - lo) >> 1);
        if (keys[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ void __launch_bounds__(256) strict_negative_kernel(const long long *__restrict__ keys, const long long n_key,
                                                              const long long *__restrict__ anchor, const long long *__restrict__ rel,
                                                              const long long *__restrict__ positive, const float *__restrict__ rand,
                                                              const int n_draw, const long long num_node, const long long num_rel,
                                                              long long *__restrict__ out) {
    const int q = blockIdx.x;
    __shared__ long long s_lo, s_hi;
    __shared__ int s_pos_known;
    const long long base = (anchor[q] * num_rel + rel[q]) * num_node;
    const long long pos = positive[q];
    if (threadIdx.x == 0) {
        s_lo = lower_bound_keys(keys, 0, n_key, base);
        s_hi = lower_bound_keys(keys, s_lo, n_key, base + num_node);
        const long long at = lower_bound_keys(keys, s_lo, s_hi, base + pos);
        s_pos_known = (at < s_hi && keys[at] == base + pos) ? 1 : 0;
    }
    __syncthreads();
    const long long lo = s_lo, m = s_hi - s_lo;
    const int extra = s_pos_known ? 0 : 1;           // the positive is excluded too (tasks.py:110-111), known answer or not
    const long long count = num_node - m - extra;    // = mask.sum(dim=-1)
    // #excluded ids <= x
    const auto excluded_upto = [&](const long long x) {
        const long long k = lower_bound_keys(keys, lo, lo + m, base + x + 1) - lo;
        return k + ((extra && pos <= x) ? 1 : 0);
    };
    for (int d = threadIdx.x; d < n_draw; d += blockDim.x) {
        // (rand * count).long() in the reference: an fp32 product, truncated (tasks.py:60)
        long long idx = (long long)(rand[(long long)q * n_draw + d] * (float)count);
        if (idx > count - 1) idx = count - 1;
        if (idx < 0) idx = 0;
        // smallest x with (x + 1) - excluded_upto(x) >= idx + 1: the idx-th valid id
        long long a = idx, b = num_node - 1;           // (x >= idx: at most x + 1 valid ids up to x)
        while (a < b) {
            const long long mid = a + ((b - a) >> 1);
            if (mid + 1 - excluded_upto(mid) >= idx + 1) b = mid; else a = mid + 1;
        }
        out[(long long)q * n_draw + d] = a;
    }
}

}  // namespace ultra

using namespace ultra;

extern "C" {

int32_t ultra_strict_negatives(const int64_t *sorted_keys_dev, int64_t n_key, const int64_t *anchor_dev, const int64_t *relation_dev,
                               const int64_t *positive_dev, const void *rand_dev, int64_t n_query, int64_t n_draw, int64_t num_node,
                               int64_t num_relation, int64_t *out_dev, void *stream) {
    if (n_query == 0 || n_draw == 0) return ULTRA_OK;
    if (!sorted_keys_dev || !anchor_dev || !relation_dev || !positive_dev || !rand_dev || !out_dev || n_query < 0 || n_draw < 0 ||
        n_key < 0 || num_node <= 0 || num_relation <= 0 || n_draw > (1 << 30) || n_query > (1 << 30)) {
        set_error("ultra_strict_negatives: NULL operand or bad size");
        return ULTRA_ERR_INVALID;
    }
    ULTRA_DEVICE_SCOPE(stream, out_dev);
    (void)hipGetLastError();
    hipLaunchKernelGGL(strict_negative_kernel, dim3((unsigned)n_query), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const long long *>(sorted_keys_dev), (long long)n_key,
                       reinterpret_cast<const long long *>(anchor_dev), reinterpret_cast<const long long *>(relation_dev),
                       reinterpret_cast<const long long *>(positive_dev), static_cast<const float *>(rand_dev), (int)n_draw,
                       (long long)num_node, (long long)num_relation, reinterpret_cast<long long *>(out_dev));
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error(std::string("strict_negative_kernel launch: ") + hipGetErrorString(e));
        return ULTRA_ERR_HIP;
    }
    return ULTRA_OK;
}

}  // extern "C"
