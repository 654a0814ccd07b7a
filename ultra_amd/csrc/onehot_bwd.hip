// Backward of an NBFNet's FIRST layer rspmm (fine-tuning path): the input is the boundary condition -- values[b] at row
// src[b] of sample b, zero elsewhere (models.py:59-66, 135-141) -- so only the edges LEAVING src[b] carry a message:
//
//     out[b, row_e] += w_e rel[b, type_e] * values[b]      (col_e == src[b]);          out[b, src[b]] += values[b]
//
// and the gradients need nothing but those edges (rspmm.cpp:106-112 restricted to them):
//
//     S[b, t]              = sum_{e: col_e == src[b], type_e == t} w_e output_grad[b, row_e]
//     relation_grad[b, t]  = values[b] * S[b, t]
//     values_grad[b]       = sum_t rel[b, t] * S[b, t] + output_grad[b, src[b]]
//
// One workgroup per (sample, 64-column span).  The source's out-edges arrive sorted by type (the caller's CSR, keyed
// (source, type)); the 64 sixteen-lane groups take contiguous chunks of them and sum every run of one type in registers.
// A run that lies inside a chunk holds ALL edges of its type: its sum is S[t].  The first and last run of a chunk may
// continue in the neighbours: those partial sums go to a side table and one group folds them in chunk order -- no atomics,
// the same bits run to run.  Hubs (thousands of out-edges) are spread over all 64 groups.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "../../include/ultra_rspmm.h"
#include "plan.hpp"
#include "device_scope.hpp"

namespace ultra {

constexpr int OB_GROUPS = 64;                 // 16 waves x 4 sixteen-lane groups
constexpr int OB_SIDE_FLOATS = 68;            // one boundary partial: 64 sums + its type (+ padding: rows 16 bytes apart in banks)

struct OnehotBwdParams {
    const int64_t *out_ptr, *out_edge, *edge_target, *edge_type, *src_rows;
    const float *edge_weight;
    const float *rel;
    long long rel_stride_outer, rel_stride_row;
    const float *values;
    const float *og;
    long long og_stride_outer, og_stride_row;
    float *rel_grad, *values_grad;
    int32_t num_rel, row_len, spans;
};

__global__ void __launch_bounds__(1024) onehot_bwd_kernel(const OnehotBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *S = reinterpret_cast<float *>(smem);                                   // [num_rel][64]
    float *side = S + (size_t)p.num_rel * 64;                                     // [64 groups][2][OB_SIDE_FLOATS]
    const int tid = threadIdx.x, l16 = tid & 15, G = tid >> 4;
    const int b = blockIdx.x / p.spans, d0 = (blockIdx.x - b * p.spans) * 64 + 4 * l16;
    const long long src = p.src_rows[b];
    const long long start = p.out_ptr[src];
    const int deg = (int)(p.out_ptr[src + 1] - start);
    const float *og = p.og + b * p.og_stride_outer + d0;

    for (int i = tid; i < p.num_rel * 16; i += blockDim.x) reinterpret_cast<float4 *>(S)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (l16 < 2) side[(G * 2 + l16) * OB_SIDE_FLOATS + 64] = -1.f;                 // (type slot: empty)
    __syncthreads();

    const int chunk = (deg + OB_GROUPS - 1) / OB_GROUPS;
    const int j0 = min(deg, G * chunk), j1 = min(deg, j0 + chunk);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int cur_t = -1, runs = 0;
    const auto flush = [&](const bool last) {
        if (cur_t < 0) return;
        if (runs == 0 || last) {       // first or last run of the chunk: may continue next door
            float *dst = side + (G * 2 + (runs == 0 ? 0 : 1)) * OB_SIDE_FLOATS;
            *reinterpret_cast<float4 *>(dst + 4 * l16) = acc;
            if (l16 == 0) dst[64] = (float)cur_t;
        } else {
            *reinterpret_cast<float4 *>(S + (size_t)cur_t * 64 + 4 * l16) = acc;
        }
        ++runs;
    };
    for (int base = j0; base < j1; base += 16) {
        // sixteen edge records per group at a time, one per lane, then broadcast inside the group
        int my_t = 0;
        long long my_row = 0;
        float my_w = 0.f;
        if (base + l16 < j1) {
            const long long e = p.out_edge[start + base + l16];
            my_row = p.edge_target[e];
            my_t = (int)p.edge_type[e];
            my_w = p.edge_weight ? p.edge_weight[e] : 1.f;
        }
        const int n = min(16, j1 - base);
        for (int q0 = 0; q0 < n; q0 += 4) {
            float4 v[4];
            int t[4];
            float w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int srcl = (tid & 48) | ((q0 + u) & 15);
                t[u] = __shfl(my_t, srcl);
                w[u] = __shfl(my_w, srcl);
                const long long row = ((long long)__shfl((int)(my_row >> 32), srcl) << 32) | (uint32_t)__shfl((int)my_row, srcl);
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q0 + u < n) v[u] = *reinterpret_cast<const float4 *>(og + row * p.og_stride_row);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (q0 + u >= n) break;
                if (t[u] != cur_t) {
                    flush(false);
                    cur_t = t[u];
                    acc = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                acc.x += w[u] * v[u].x;
                acc.y += w[u] * v[u].y;
                acc.z += w[u] * v[u].z;
                acc.w += w[u] * v[u].w;
            }
        }
    }
    if (runs == 0) {
        flush(false);                  // a single run: the chunk's first (and last)
    } else {
        flush(true);
    }
    __syncthreads();
    // boundary partials, in chunk order: one group, sequential
    if (G == 0) {
        for (int i = 0; i < OB_GROUPS * 2; ++i) {
            const float *part = side + i * OB_SIDE_FLOATS;
            const int t = (int)part[64];
            if (t < 0) continue;
            float4 s = *reinterpret_cast<const float4 *>(S + (size_t)t * 64 + 4 * l16);
            const float4 a = *reinterpret_cast<const float4 *>(part + 4 * l16);
            s.x += a.x, s.y += a.y, s.z += a.z, s.w += a.w;
            *reinterpret_cast<float4 *>(S + (size_t)t * 64 + 4 * l16) = s;
        }
    }
    __syncthreads();
    // relation_grad = values * S;  values_grad = sum_t rel[t] * S[t] (+ the boundary's share)
    const float4 val = *reinterpret_cast<const float4 *>(p.values + (long long)b * p.row_len + d0);
    const float *rel = p.rel + b * p.rel_stride_outer + d0;
    float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = G; t < p.num_rel; t += OB_GROUPS) {
        const float4 s = *reinterpret_cast<const float4 *>(S + (size_t)t * 64 + 4 * l16);
        if (p.rel_grad)
            *reinterpret_cast<float4 *>(p.rel_grad + ((long long)b * p.num_rel + t) * p.row_len + d0) =
                make_float4(val.x * s.x, val.y * s.y, val.z * s.z, val.w * s.w);
        if (p.values_grad) {
            const float4 r = *reinterpret_cast<const float4 *>(rel + (long long)t * p.rel_stride_row);
            pv.x += r.x * s.x, pv.y += r.y * s.y, pv.z += r.z * s.z, pv.w += r.w * s.w;
        }
    }
    if (!p.values_grad) return;
    __syncthreads();                   // (the side table is free again)
    *reinterpret_cast<float4 *>(side + G * OB_SIDE_FLOATS + 4 * l16) = pv;
    __syncthreads();
    if (G == 0) {
        float4 s = *reinterpret_cast<const float4 *>(og + src * p.og_stride_row);
        for (int g = 0; g < OB_GROUPS; ++g) {
            const float4 a = *reinterpret_cast<const float4 *>(side + g * OB_SIDE_FLOATS + 4 * l16);
            s.x += a.x, s.y += a.y, s.z += a.z, s.w += a.w;
        }
        *reinterpret_cast<float4 *>(p.values_grad + (long long)b * p.row_len + d0) = s;
    }
}

}  // namespace ultra

using namespace ultra;

extern "C" {

int32_t ultra_rspmm_onehot_backward(const int64_t *out_ptr_dev, const int64_t *out_edge_dev, const int64_t *edge_target_dev,
                                    const int64_t *edge_type_dev, const void *edge_weight_dev, const ultra_mat *relation,
                                    const void *values_dev, const int64_t *src_rows_dev, const ultra_mat *output_grad,
                                    void *relation_grad_dev, void *values_grad_dev, void *stream) {
    if (!out_ptr_dev || !out_edge_dev || !edge_target_dev || !edge_type_dev || !relation || !relation->ptr || !values_dev ||
        !src_rows_dev || !output_grad || !output_grad->ptr) {
        set_error("ultra_rspmm_onehot_backward: NULL operand");
        return ULTRA_ERR_INVALID;
    }
    ULTRA_DEVICE_SCOPE(stream, output_grad->ptr);
    const int64_t bs = output_grad->n_outer, row_len = output_grad->row_len, num_rel = relation->n_row;
    if (bs <= 0 || row_len <= 0 || relation->n_outer != bs || relation->row_len != row_len) {
        set_error("ultra_rspmm_onehot_backward: relation and output_grad disagree on n_outer / row_len");
        return ULTRA_ERR_INVALID;
    }
    const auto vec_ok = [](const ultra_mat *m) {
        return (reinterpret_cast<uintptr_t>(m->ptr) % 16 == 0) && m->stride_row % 4 == 0 && m->stride_outer % 4 == 0;
    };
    const size_t lds = (size_t)num_rel * 64 * sizeof(float) + (size_t)OB_GROUPS * 2 * OB_SIDE_FLOATS * sizeof(float);
    const size_t lds_max = 160 * 1024;      // gfx950: 160 KB of LDS per workgroup (rspmm_api.hip device_info)
    if (row_len % 64 != 0 || !vec_ok(relation) || !vec_ok(output_grad) || reinterpret_cast<uintptr_t>(values_dev) % 16 != 0 ||
        num_rel <= 0 || lds > lds_max || bs * (row_len / 64) > (1 << 20)) {
        set_error("ultra_rspmm_onehot_backward: served for fp32 rows of whole 64-element spans, 16-byte aligned, with the "
                  "relation-gradient image of one span in LDS");
        return ULTRA_ERR_UNSUPPORTED;
    }
    OnehotBwdParams p;
    p.out_ptr = out_ptr_dev, p.out_edge = out_edge_dev, p.edge_target = edge_target_dev, p.edge_type = edge_type_dev;
    p.src_rows = src_rows_dev;
    p.edge_weight = static_cast<const float *>(edge_weight_dev);
    p.rel = static_cast<const float *>(relation->ptr);
    p.rel_stride_outer = relation->stride_outer, p.rel_stride_row = relation->stride_row;
    p.values = static_cast<const float *>(values_dev);
    p.og = static_cast<const float *>(output_grad->ptr);
    p.og_stride_outer = output_grad->stride_outer, p.og_stride_row = output_grad->stride_row;
    p.rel_grad = static_cast<float *>(relation_grad_dev);
    p.values_grad = static_cast<float *>(values_grad_dev);
    p.num_rel = (int32_t)num_rel, p.row_len = (int32_t)row_len, p.spans = (int32_t)(row_len / 64);
    (void)hipGetLastError();
    static size_t lds_opted_in = 0;
    if (lds > 48 * 1024 && lds > lds_opted_in) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(onehot_bwd_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error(std::string("ultra_rspmm_onehot_backward: LDS opt-in: ") + hipGetErrorString(e));
            return ULTRA_ERR_HIP;
        }
        lds_opted_in = lds;
    }
    hipLaunchKernelGGL(onehot_bwd_kernel, dim3((unsigned)(bs * p.spans)), dim3(1024), lds, reinterpret_cast<hipStream_t>(stream), p);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error(std::string("onehot_bwd_kernel launch: ") + hipGetErrorString(e));
        return ULTRA_ERR_HIP;
    }
    return ULTRA_OK;
}

}  // extern "C"
