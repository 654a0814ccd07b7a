// add_mul rspmm on a dense-format plan (ULTRA_PLAN_DENSE): out = sum_t rel[t] * (A_t . x) [+ boundary] on the matrix cores.
//
// ULTRA's relation graphs are a few hundred nodes with 4 edge types and most (row, type, col) cells occupied
// (rspmm.cpp:50-75 walks ~900 k edges for a 474-node graph); stored as R dense multiplicity matrices the same
// aggregation is a (num_out x R*num_in) . (R*num_in x row_len) product.  One workgroup owns a 32-row x 32-column
// output tile; its four waves split the source rows (k) into quarters and each keeps one accumulator tile per
// relation type (TC = 1 / 2 / 4 types at a time), so one B operand -- x[k .. k+1][32 columns], loaded straight from
// L2 in operand layout: two fully used 128-byte lines per wave -- feeds TC v_mfma_f32_32x32x2_f32.  The adjacency
// is stored as BYTES in A-operand fragment order: one load of 4 TC bytes per lane covers an 8-row step (4 TC matrix
// instructions, >= 1024 matrix-pipe cycles), converted with v_cvt_f32_ubyte.  Each wave scales its per-type tiles by
// rel[type] (one scalar per lane: a lane owns one output column), and the four k-quarters are added in wave order
// through LDS -- no atomics, deterministic.
// A_t holds small integers, so every product is exact and the result differs from the edge walk only in the order of
// the fp32 additions (same class of difference as the type-run plan; bounded in tests/helpers.assert_sum_close).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "plan.hpp"

namespace ultra {

using f32x16 = float __attribute__((ext_vector_type(16)));

struct DenseParams {
    const uint32_t *a_frag;   // [n_rt][n_tc][kg][64 lanes][tc] words of 4 multiplicity bytes
    const float *rel, *x, *bnd;
    const long long *bnd_rows;   // point boundary (one row per outer slice, bnd_sr = 0) or NULL
    float *out;
    long long rel_so, rel_sr, x_so, x_sr, bnd_so, bnd_sr, out_so, out_sr;
    int n_out, n_in, n_rel, kg, n_rt, n_tc, n_ct, row_len, has_bnd;
};

static constexpr int DENSE_PF = ULTRA_DENSE_KG_ALIGN / 4;   // 8-column steps in flight per wave (one step = 4 TC matrix instructions)

template <int TC>
struct DenseStage {
    uint32_t a[TC];
    float x[4];
};

// RG: the relation gradient of the same product instead of the product (fine-tuning, the relation graph's backward):
//     relation_grad[t][col] = sum_row og[row][col] * (A_t . x)[row][col]            (rspmm.cpp:106-108 for add_mul, unit weights)
// -- the per-type tiles this kernel holds anyway, weighed with the output gradient (p.bnd) instead of rel[t] and summed over the
// tile's rows instead of over the types.  Each workgroup leaves one partial row per type (p.out: [row tile][type][column]);
// dense_rgrad_combine_kernel adds the row tiles in order.  The edge walk over the relation-major plan took 71 + 12 us per
// relation-model layer for this (a 474-node graph with 0.9 M edges), five times a step.
template <int TC, bool RG = false>
__global__ void __launch_bounds__(256) rspmm_dense_kernel(const DenseParams p) {
    __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64];   // [wave][accumulator register][lane]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    // block -> (row tile, column tile): the row tiles (adjacency slices) are spread over the XCDs (block % 8)
    const int xcd = blockIdx.x & 7, qq = blockIdx.x >> 3;
    const int ct = qq % p.n_ct, rt = xcd + 8 * (qq / p.n_ct);
    if (rt >= p.n_rt) return;
    const int g0 = ct * 32, outer = g0 / p.row_len, d0 = g0 % p.row_len;

    // this wave's quarter of the source rows: 8-row steps [kb, ke); p.kg is a multiple of 4 DENSE_PF (zero padded), so
    // the software pipeline below runs whole blocks of DENSE_PF steps without a tail
    const int steps = p.kg / 4, kb = wave * steps, ke = kb + steps;
    // B operand: lane (h, j) holds x[8 kg + 2 q + h][d0 + j]; 32-bit byte offsets from a uniform base
    const char *xb = reinterpret_cast<const char *>(p.x + (long long)outer * p.x_so + d0 + j);
    const uint32_t x_row_bytes = (uint32_t)p.x_sr * 4u;
    const float *relb = p.rel + (long long)outer * p.rel_so + d0 + j;
    f32x16 total;
#pragma unroll
    for (int r = 0; r < 16; ++r) total[r] = 0.f;
    float ogv[16];      // RG: og[rt * 32 + (r & 3) + 8 (r >> 2) + 4 h][d0 + j], zero past the last row
    if constexpr (RG) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            ogv[r] = row < p.n_out ? p.bnd[(long long)outer * p.bnd_so + (long long)row * p.bnd_sr + d0 + j] : 0.f;
        }
    }

    for (int tc = 0; tc < p.n_tc; ++tc) {
        f32x16 acc[TC];
#pragma unroll
        for (int tl = 0; tl < TC; ++tl)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tl][r] = 0.f;
        const char *ap = reinterpret_cast<const char *>(p.a_frag + ((size_t)(rt * p.n_tc + tc) * p.kg * 64 + lane) * TC);
        const auto fetch = [&](int kg, DenseStage<TC> &st) {   // kg is wave-uniform
            const char *a = ap + (uint32_t)kg * (uint32_t)(64 * TC * 4);
            if constexpr (TC == 4) {
                const uint4 v = *reinterpret_cast<const uint4 *>(a);
                st.a[0] = v.x, st.a[1] = v.y, st.a[2] = v.z, st.a[3] = v.w;
            } else if constexpr (TC == 2) {
                const uint2 v = *reinterpret_cast<const uint2 *>(a);
                st.a[0] = v.x, st.a[1] = v.y;
            } else {
                st.a[0] = *reinterpret_cast<const uint32_t *>(a);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = 8 * kg + 2 * q + h;   // rows past n_in: read a valid row, zeroed where it is consumed
                st.x[q] = *reinterpret_cast<const float *>(xb + (uint32_t)min(k, p.n_in - 1) * x_row_bytes);
            }
        };
        // chunk-level double buffering: the loads of the next DENSE_PF steps are all issued before the matrix work of
        // the current DENSE_PF steps (>= 1024 matrix-pipe cycles each) and are only waited for after it
        DenseStage<TC> cur[DENSE_PF], nxt[DENSE_PF];
#pragma unroll
        for (int i = 0; i < DENSE_PF; ++i) fetch(kb + i, cur[i]);
        for (int base = kb; base < ke; base += DENSE_PF) {
            const int nb = min(base + DENSE_PF, ke - DENSE_PF);   // (the last block re-reads itself: harmless)
#pragma unroll
            for (int i = 0; i < DENSE_PF; ++i) fetch(nb + i, nxt[i]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < DENSE_PF; ++i) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // padding rows: multiplicity 0 times an exact 0 (never 0 * garbage)
                    const float xq = 8 * (base + i) + 2 * q + h < p.n_in ? cur[i].x[q] : 0.f;
#pragma unroll
                    for (int tl = 0; tl < TC; ++tl)
                        acc[tl] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)((cur[i].a[tl] >> (8 * q)) & 0xffu), xq, acc[tl], 0,
                                                                       0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < DENSE_PF; ++i) cur[i] = nxt[i];
        }
#pragma unroll
        for (int tl = 0; tl < TC; ++tl) {
            const int t = tc * TC + tl;
            if (t < p.n_rel) {
                if constexpr (RG) {
                    float sacc = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc += ogv[r] * acc[tl][r];
                    sacc += __shfl_xor(sacc, 32);
                    if (h == 0) red[(wave * p.n_rel + t) * 32 + j] = sacc;
                } else {
                    const float rv = relb[(long long)t * p.rel_sr];
#pragma unroll
                    for (int r = 0; r < 16; ++r) total[r] += rv * acc[tl][r];
                }
            }
        }
    }
    if constexpr (RG) {
        __syncthreads();
        for (int idx = tid; idx < p.n_rel * 32; idx += 256) {     // (type, column): the four source-row quarters in wave order
            const float sum = ((red[idx] + red[p.n_rel * 32 + idx]) + red[2 * p.n_rel * 32 + idx]) + red[3 * p.n_rel * 32 + idx];
            p.out[((long long)rt * p.n_rel + idx / 32) * ((long long)p.n_ct * 32) + g0 + (idx & 31)] = sum;
        }
        return;
    }

#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = total[r];
    __syncthreads();
    // thread -> (tile row, 4-column chunk); accumulator register r of lane (h, j) holds tile row (r & 3) + 8 (r >> 2) + 4 h
    const int r = tid >> 3, c4 = tid & 7;
    const int reg = (r >> 3) * 4 + (r & 3), ln = ((r >> 2) & 1) * 32 + c4 * 4;
    const float4 *red4 = reinterpret_cast<const float4 *>(red);
    float4 sum = red4[((0 * 16 + reg) * 64 + ln) >> 2];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const float4 v = red4[((w * 16 + reg) * 64 + ln) >> 2];
        sum.x += v.x;
        sum.y += v.y;
        sum.z += v.z;
        sum.w += v.w;
    }
    const int row = rt * 32 + r;
    if (row < p.n_out) {
        if (p.has_bnd && (!p.bnd_rows || p.bnd_rows[outer] == row)) {
            const float4 b = *reinterpret_cast<const float4 *>(p.bnd + (long long)outer * p.bnd_so + (long long)row * p.bnd_sr +
                                                                 d0 + c4 * 4);
            sum.x += b.x;
            sum.y += b.y;
            sum.z += b.z;
            sum.w += b.w;
        }
        *reinterpret_cast<float4 *>(p.out + (long long)outer * p.out_so + (long long)row * p.out_sr + d0 + c4 * 4) = sum;
    }
}

// relation_grad[outer][t][d] = sum over the row tiles of partial[tile][t][outer * row_len + d], tiles in ascending order
__global__ void __launch_bounds__(256) dense_rgrad_combine_kernel(const float *__restrict__ partial, int n_rt, int n_rel, int row_len,
                                                                  long long ncol, float *__restrict__ out, long long out_so,
                                                                  long long out_sr) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)n_rel * ncol) return;
    const int t = (int)(idx / ncol);
    const long long col = idx % ncol;
    float s = 0.f;
    for (int rt = 0; rt < n_rt; ++rt) s += partial[((long long)rt * n_rel + t) * ncol + col];
    out[(col / row_len) * out_so + (long long)t * out_sr + col % row_len] = s;
}

static bool ok16(const ultra_mat *m) {
    return (reinterpret_cast<uintptr_t>(m->ptr) & 15u) == 0 && m->stride_row % 4 == 0 && m->stride_outer % 4 == 0;
}

// Called by forward_impl for ULTRA_PLAN_DENSE plans (operands already shape-checked, plan uploaded).
int launch_dense_forward(ultra_plan *p, int sum, int mul, int dtype, const void *w, const ultra_mat *rel, const ultra_mat *x,
                         const ultra_mat *bnd, const int64_t *bnd_rows, const ultra_mat *out, hipStream_t stream) {
    if (sum != ULTRA_SUM_ADD || mul != ULTRA_MUL_MUL || dtype != ULTRA_F32 || w != nullptr) {
        set_error("a ULTRA_PLAN_DENSE plan serves fp32 add_mul with unit edge weights only");
        return ULTRA_ERR_UNSUPPORTED;
    }
    if ((uint64_t)p->num_in * (uint64_t)x->stride_row * 4u >= (1ull << 32)) {
        set_error("ULTRA_PLAN_DENSE: an input slice (rows * stride_row) exceeds 4 GiB; use the batch-major layout");
        return ULTRA_ERR_UNSUPPORTED;
    }
    if (out->row_len % 32 != 0 || !ok16(rel) || !ok16(x) || !ok16(out) || (bnd && !ok16(bnd))) {
        set_error("ULTRA_PLAN_DENSE: row_len must be a multiple of 32 and operands 16-byte aligned");
        return ULTRA_ERR_UNSUPPORTED;
    }
    DenseParams dp;
    dp.a_frag = reinterpret_cast<const uint32_t *>(p->d.a_frag);
    dp.rel = static_cast<const float *>(rel->ptr);
    dp.x = static_cast<const float *>(x->ptr);
    dp.bnd = bnd ? static_cast<const float *>(bnd->ptr) : nullptr;
    dp.out = static_cast<float *>(out->ptr);
    dp.rel_so = rel->stride_outer, dp.rel_sr = rel->stride_row;
    dp.x_so = x->stride_outer, dp.x_sr = x->stride_row;
    dp.bnd_so = bnd ? bnd->stride_outer : 0, dp.bnd_sr = (bnd && !bnd_rows) ? bnd->stride_row : 0;
    dp.bnd_rows = bnd ? reinterpret_cast<const long long *>(bnd_rows) : nullptr;
    dp.out_so = out->stride_outer, dp.out_sr = out->stride_row;
    dp.n_out = (int)p->num_out, dp.n_in = (int)p->num_in, dp.n_rel = (int)p->num_rel;
    dp.kg = p->dense_kg, dp.n_rt = p->dense_rt, dp.n_tc = p->dense_ntc;
    dp.n_ct = (int)(out->n_outer * out->row_len / 32);
    dp.row_len = (int)out->row_len;
    dp.has_bnd = bnd ? 1 : 0;
    const long long blocks = (long long)((p->dense_rt + 7) / 8) * 8 * dp.n_ct;
    const dim3 grid((unsigned)blocks), block(256);
    switch (p->dense_tc) {
        case 1: hipLaunchKernelGGL(rspmm_dense_kernel<1>, grid, block, 0, stream, dp); break;
        case 2: hipLaunchKernelGGL(rspmm_dense_kernel<2>, grid, block, 0, stream, dp); break;
        default: hipLaunchKernelGGL(rspmm_dense_kernel<4>, grid, block, 0, stream, dp); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error(std::string("rspmm_dense_kernel launch: ") + hipGetErrorString(e));
        return ULTRA_ERR_HIP;
    }
    return ULTRA_OK;
}

// relation_grad of add_mul with unit edge weights on a ULTRA_PLAN_DENSE plan (operands shape-checked by the caller).
// `scratch` holds dense_rt * num_rel * n_outer * row_len floats.
int launch_dense_relation_grad(ultra_plan *p, const ultra_mat *x, const ultra_mat *og, const ultra_mat *rgrad, float *scratch,
                               hipStream_t stream) {
    if (p->num_rel > 32 || og->row_len % 32 != 0 || !ok16(x) || !ok16(og) || !ok16(rgrad) ||
        (uint64_t)p->num_in * (uint64_t)x->stride_row * 4u >= (1ull << 32)) {
        set_error("ULTRA_PLAN_DENSE relation gradient: at most 32 relation types, row_len a multiple of 32, 16-byte aligned operands");
        return ULTRA_ERR_UNSUPPORTED;
    }
    DenseParams dp;
    dp.a_frag = reinterpret_cast<const uint32_t *>(p->d.a_frag);
    dp.rel = nullptr;
    dp.x = static_cast<const float *>(x->ptr);
    dp.bnd = static_cast<const float *>(og->ptr);
    dp.out = scratch;
    dp.rel_so = dp.rel_sr = 0;
    dp.x_so = x->stride_outer, dp.x_sr = x->stride_row;
    dp.bnd_so = og->stride_outer, dp.bnd_sr = og->stride_row;
    dp.bnd_rows = nullptr;
    dp.out_so = dp.out_sr = 0;
    dp.n_out = (int)p->num_out, dp.n_in = (int)p->num_in, dp.n_rel = (int)p->num_rel;
    dp.kg = p->dense_kg, dp.n_rt = p->dense_rt, dp.n_tc = p->dense_ntc;
    dp.n_ct = (int)(og->n_outer * og->row_len / 32);
    dp.row_len = (int)og->row_len;
    dp.has_bnd = 0;
    const long long blocks = (long long)((p->dense_rt + 7) / 8) * 8 * dp.n_ct;
    const dim3 grid((unsigned)blocks), block(256);
    switch (p->dense_tc) {
        case 1: hipLaunchKernelGGL((rspmm_dense_kernel<1, true>), grid, block, 0, stream, dp); break;
        case 2: hipLaunchKernelGGL((rspmm_dense_kernel<2, true>), grid, block, 0, stream, dp); break;
        default: hipLaunchKernelGGL((rspmm_dense_kernel<4, true>), grid, block, 0, stream, dp); break;
    }
    const long long ncol = (long long)dp.n_ct * 32, total = (long long)dp.n_rel * ncol;
    hipLaunchKernelGGL(dense_rgrad_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, scratch, dp.n_rt,
                       dp.n_rel, dp.row_len, ncol, static_cast<float *>(rgrad->ptr), (long long)rgrad->stride_outer,
                       (long long)rgrad->stride_row);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error(std::string("rspmm_dense_kernel (relation gradient) launch: ") + hipGetErrorString(e));
        return ULTRA_ERR_HIP;
    }
    return ULTRA_OK;
}

}  // namespace ultra
