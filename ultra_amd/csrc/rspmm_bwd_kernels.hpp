// Edge-parallel backward kernels (rspmm.cpp:77-119 restated for wave64).
//
// sum == add needs only weight_grad from here: input_grad and relation_grad are atomics-free
// re-runs of the forward kernel on the transposed / relation-major plans (rspmm_api.hip).
// sum == min / max routes everything through this kernel: the gradient reaches every edge whose
// message equals the forward output (operator.cuh:62-64, 75-77), accumulated with hardware
// float atomics like the reference's CUDA path (rspmm.cu:142-147, 208-209).
//
// One 16-lane group owns one sorted edge and sweeps its feature spans; weight_grad is reduced
// inside the group with four xor-shuffles and stored once (no atomics) in ORIGINAL edge order.
#pragma once

#include <hip/hip_runtime.h>

#include "rspmm_kernels.hpp"

#pragma clang fp contract(off)

namespace ultra {

struct EdgeParams {
    const int32_t *erow;
    const int32_t *col;
    const int32_t *type;
    const int32_t *perm;
    const void *w;  // original edge order, may be NULL (= ones)
    int64_t num_edge;
    int32_t num_out;
    MatArg rel, x, out, og;
    void *rgrad;
    long long rgrad_so, rgrad_sr;
    void *xgrad;
    long long xgrad_so, xgrad_sr;
    void *wgrad;  // original edge order, may be NULL
    int32_t n_outer, row_len;
};

template <typename T, int VEC, int SUM, int MUL, bool WANT_RI>
__global__ void __launch_bounds__(256) rspmm_edge_bwd_kernel(const EdgeParams p) {
    constexpr int SPAN = 16 * VEC;
    using P = Pack<T, VEC>;
    const int l16 = threadIdx.x & 15;
    const long long g0 = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 4;
    const long long ng = ((long long)gridDim.x * blockDim.x) >> 4;
    const int spans_per_outer = (p.row_len + SPAN - 1) / SPAN;
    for (long long k = g0; k < p.num_edge; k += ng) {
        const int row = p.erow[k], col = p.col[k], type = p.type[k];
        const int eid = p.perm[k];
        const T w = p.w ? reinterpret_cast<const T *>(p.w)[eid] : T(1);
        T wg = T(0);
        for (int outer = 0; outer < p.n_outer; ++outer) {
            for (int inner = 0; inner < spans_per_outer; ++inner) {
                const int d0 = inner * SPAN + l16 * VEC;
                if (d0 >= p.row_len) continue;
                const P r = *reinterpret_cast<const P *>(reinterpret_cast<const T *>(p.rel.ptr) +
                                                         outer * p.rel.stride_outer + (long long)type * p.rel.stride_row + d0);
                const P xi = *reinterpret_cast<const P *>(reinterpret_cast<const T *>(p.x.ptr) + outer * p.x.stride_outer +
                                                          (long long)col * p.x.stride_row + d0);
                const P g = *reinterpret_cast<const P *>(reinterpret_cast<const T *>(p.og.ptr) + outer * p.og.stride_outer +
                                                         (long long)row * p.og.stride_row + d0);
                P o;
                if (SUM != ULTRA_SUM_ADD)
                    o = *reinterpret_cast<const P *>(reinterpret_cast<const T *>(p.out.ptr) + outer * p.out.stride_outer +
                                                     (long long)row * p.out.stride_row + d0);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const T xb = binary<T, MUL>(r.v[e], xi.v[e]);
                    const T y = w * xb;
                    const T dout_dy = (SUM == ULTRA_SUM_ADD) ? T(1) : (o.v[e] == y ? T(1) : T(0));
                    const T t = g.v[e] * dout_dy;
                    wg += t * xb;
                    if (WANT_RI) {
                        const T tw = t * w;
                        const T dx_drel = (MUL == BIN_MUL) ? xi.v[e] : T(1);
                        const T dx_din = (MUL == BIN_MUL) ? r.v[e] : T(1);
                        T *rg = reinterpret_cast<T *>(p.rgrad) + outer * p.rgrad_so + (long long)type * p.rgrad_sr + d0 + e;
                        T *xg = reinterpret_cast<T *>(p.xgrad) + outer * p.xgrad_so + (long long)col * p.xgrad_sr + d0 + e;
                        if (tw != T(0)) {
                            unsafeAtomicAdd(rg, tw * dx_drel);
                            unsafeAtomicAdd(xg, tw * dx_din);
                        }
                    }
                }
            }
        }
        if (p.wgrad) {
            wg += __shfl_xor(wg, 8);
            wg += __shfl_xor(wg, 4);
            wg += __shfl_xor(wg, 2);
            wg += __shfl_xor(wg, 1);
            if (l16 == 0) reinterpret_cast<T *>(p.wgrad)[eid] = wg;
        }
    }
}

// One-hot (row-sparse) input, add_mul: x[outer] is zero except row src[outer].  Only the edges gathered FROM that
// row contribute, i.e. the src[outer]-th row of the TRANSPOSED plan (sorted by target, stable in the edge id):
//   out[outer, target, :] += w_e * rel[outer, type_e, :] * x[outer, src, :]
// out was zero-filled by the caller; runs of equal target are summed by one 16-lane group in edge order (no atomics),
// then boundary[outer, src] is added to the source row (the only non-zero boundary row of an NBFNet layer-0 input).
struct OneHotParams {
    const int32_t *trow_ptr;   // transposed plan: row = gathered source
    const int32_t *tcol;       // = aggregation target
    const int32_t *ttype;
    const int32_t *tperm;
    const void *w;             // original edge order or NULL
    const int64_t *src;        // [n_outer]
    MatArg rel, x, bnd;
    void *out;
    long long out_so, out_sr;
    int32_t n_outer, row_len, has_bnd;
};

template <typename T, int VEC>
__global__ void __launch_bounds__(256) rspmm_onehot_kernel(const OneHotParams p) {
    constexpr int SPAN = 16 * VEC;
    using P = Pack<T, VEC>;
    const int outer = blockIdx.y;
    const int l16 = threadIdx.x & 15;
    const int grp = blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);
    const int ngrp = gridDim.x * (blockDim.x >> 4);
    const long long s = p.src[outer];
    const int k0 = p.trow_ptr[s], k1 = p.trow_ptr[s + 1];
    const int spans = (p.row_len + SPAN - 1) / SPAN;
    const T *xrow = reinterpret_cast<const T *>(p.x.ptr) + outer * p.x.stride_outer + s * p.x.stride_row;
    const T *relb = reinterpret_cast<const T *>(p.rel.ptr) + outer * p.rel.stride_outer;
    const T *bndrow = p.has_bnd ? reinterpret_cast<const T *>(p.bnd.ptr) + outer * p.bnd.stride_outer + s * p.bnd.stride_row
                                : nullptr;
    T *outb = reinterpret_cast<T *>(p.out) + outer * p.out_so;
    // the source row's out-edges are sorted by target: each run of equal targets belongs to the group that owns
    // its first edge and is summed in edge order
    for (int k = k0 + grp; k < k1; k += ngrp) {
        const int target = p.tcol[k];
        if (k > k0 && p.tcol[k - 1] == target) continue;
        int kend = k + 1;
        while (kend < k1 && p.tcol[kend] == target) ++kend;
        for (int sp = 0; sp < spans; ++sp) {
            const int d0 = sp * SPAN + l16 * VEC;
            if (d0 >= p.row_len) continue;
            const P xv = *reinterpret_cast<const P *>(xrow + d0);
            P acc;
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc.v[e] = T(0);
            for (int kk = k; kk < kend; ++kk) {
                const P rv = *reinterpret_cast<const P *>(relb + (long long)p.ttype[kk] * p.rel.stride_row + d0);
                const T w = p.w ? reinterpret_cast<const T *>(p.w)[p.tperm[kk]] : T(1);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    T y = rv.v[e] * xv.v[e];
                    if (p.w) y = w * y;
                    acc.v[e] = acc.v[e] + y;
                }
            }
            if (bndrow && target == s) {   // update + boundary (layers.py:200): after the sum
                const P b = *reinterpret_cast<const P *>(bndrow + d0);
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc.v[e] = acc.v[e] + b.v[e];
            }
            *reinterpret_cast<P *>(outb + (long long)target * p.out_sr + d0) = acc;
        }
    }
    // no edge leads back to the source row itself: its output is 0 + boundary
    if (bndrow && grp == 0) {
        int lo = k0, hi = k1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (p.tcol[mid] < (int)s) lo = mid + 1; else hi = mid;
        }
        if (!(lo < k1 && p.tcol[lo] == (int)s)) {
            for (int d = l16; d < p.row_len; d += 16) outb[s * p.out_sr + d] = bndrow[d];
        }
    }
}

__global__ void __launch_bounds__(256) zero16_kernel(float4 *ptr, long long n16) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) ptr[i] = z;
}

template <typename T>
__global__ void __launch_bounds__(256) fill_zero_kernel(T *ptr, int n_outer, long long so, long long n_row, long long sr,
                                                        int row_len) {
    const long long total = (long long)n_outer * n_row * row_len;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int d = (int)(i % row_len);
        const long long r2 = i / row_len;
        const long long r = r2 % n_row;
        const int o = (int)(r2 / n_row);
        ptr[o * so + r * sr + d] = T(0);
    }
}

inline int launch_fill_zero(int dtype, const ultra_mat *m, int64_t rows, hipStream_t stream) {
    const long long total = (long long)m->n_outer * rows * m->row_len;
    if (total == 0) return ULTRA_OK;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
    if (dtype == ULTRA_F32)
        hipLaunchKernelGGL(fill_zero_kernel<float>, dim3(blocks), dim3(256), 0, stream, (float *)m->ptr, (int)m->n_outer,
                           (long long)m->stride_outer, (long long)rows, (long long)m->stride_row, (int)m->row_len);
    else
        hipLaunchKernelGGL(fill_zero_kernel<double>, dim3(blocks), dim3(256), 0, stream, (double *)m->ptr, (int)m->n_outer,
                           (long long)m->stride_outer, (long long)rows, (long long)m->stride_row, (int)m->row_len);
    if (hipGetLastError() != hipSuccess) {
        set_error("fill_zero_kernel launch failed");
        return ULTRA_ERR_HIP;
    }
    return ULTRA_OK;
}

// ---- min / max backward as a GATHER (no atomics, run-to-run deterministic) ----
// The reference scatters with atomicAdd (rspmm.cu:153-214): every edge whose message equals the forward output
// (operator.cuh:62-64, 75-77 -- every tying edge gets the full gradient) adds into input_grad[col] and
// relation_grad[type].  Here each destination row gathers instead, over the plan whose rows are that destination:
//   input_grad    : the transposed plan (rows = col).     self = x[col];    per edge: i = row, t = type
//   relation_grad : the relation-major plan (rows = type). self = rel[type]; per edge: j = col, i = row
//   contribution  = og[i] * [out[i] == w * (rel (x) x)] * w * d(rel (x) x)/d(destination)     -- the reference's factor order
// One 16-lane group walks one item (a run of at most seg_len sorted edges of one destination row) per 64-element span,
// four edges' operands in flight; rows cut into several items leave partial sums that rspmm_fixup_kernel adds in slot
// order, exactly like the forward of a re-associating plan.
struct GatherBwdParams {
    const Item *items;
    int32_t n_item;
    const int32_t *col, *type, *perm;   // of the destination-major plan: (a, b) per sorted edge, see above
    const void *w;                      // original edge order, may be NULL (= ones)
    MatArg rel, x, out, og;
    void *grad;                         // destination matrix
    long long grad_so, grad_sr;
    void *partial;
    int32_t n_outer, row_len, spans_per_outer, n_span, smod, nparts;
};

template <typename T, int SUM, int MUL, bool REL_GRAD>
__global__ void __launch_bounds__(256) rspmm_minmax_bwd_gather_kernel(const GatherBwdParams p) {
    constexpr int SPAN = 64;
    using P = Pack<T, 4>;
    const int l16 = threadIdx.x & 15;
    const int part = blockIdx.x / p.smod;
    if (part >= p.nparts) return;
    const int groups_per_block = blockDim.x >> 4;
    const T *wt = reinterpret_cast<const T *>(p.w);
    for (int span = blockIdx.x % p.smod; span < p.n_span; span += p.smod) {
        const int outer = span / p.spans_per_outer, inner = span - outer * p.spans_per_outer;
        const int d0 = inner * SPAN + l16 * 4;
        if (d0 >= p.row_len) continue;
        const T *relp = reinterpret_cast<const T *>(p.rel.ptr) + outer * p.rel.stride_outer + d0;
        const T *xp = reinterpret_cast<const T *>(p.x.ptr) + outer * p.x.stride_outer + d0;
        const T *outp = reinterpret_cast<const T *>(p.out.ptr) + outer * p.out.stride_outer + d0;
        const T *ogp = reinterpret_cast<const T *>(p.og.ptr) + outer * p.og.stride_outer + d0;
        for (int q = part * groups_per_block + (threadIdx.x >> 4); q < p.n_item; q += p.nparts * groups_per_block) {
            const Item it = p.items[q];
            const P self = REL_GRAD ? *reinterpret_cast<const P *>(relp + (long long)it.row * p.rel.stride_row)
                                    : *reinterpret_cast<const P *>(xp + (long long)it.row * p.x.stride_row);
            T acc[4] = {T(0), T(0), T(0), T(0)};
            const int end = it.begin + it.len;
            for (int k = it.begin; k < end; k += 4) {
                int a[4], b[4];
                T wv[4];
                P other[4], o[4], g[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int kk = k + u < end ? k + u : end - 1;
                    a[u] = p.col[kk], b[u] = p.type[kk];
                    wv[u] = wt ? wt[p.perm[kk]] : T(1);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = REL_GRAD ? b[u] : a[u];     // the aggregation row of the edge
                    other[u] = REL_GRAD ? *reinterpret_cast<const P *>(xp + (long long)a[u] * p.x.stride_row)
                                        : *reinterpret_cast<const P *>(relp + (long long)b[u] * p.rel.stride_row);
                    o[u] = *reinterpret_cast<const P *>(outp + (long long)i * p.out.stride_row);
                    g[u] = *reinterpret_cast<const P *>(ogp + (long long)i * p.og.stride_row);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (k + u < end) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const T r = REL_GRAD ? self.v[e] : other[u].v[e];
                            const T xi = REL_GRAD ? other[u].v[e] : self.v[e];
                            const T xb = binary<T, MUL>(r, xi);
                            const T y = wv[u] * xb;
                            const T t = g[u].v[e] * (o[u].v[e] == y ? T(1) : T(0));     // operator.cuh:62-64, 75-77
                            const T tw = t * wv[u];
                            const T d = (MUL == BIN_MUL) ? (REL_GRAD ? xi : r) : T(1);
                            acc[e] += tw * d;
                        }
                    }
                }
            }
            P res;
#pragma unroll
            for (int e = 0; e < 4; ++e) res.v[e] = acc[e];
            if (it.slot >= 0)
                *reinterpret_cast<P *>(reinterpret_cast<T *>(p.partial) + ((long long)it.slot * p.n_outer + outer) * p.row_len + d0) = res;
            else
                *reinterpret_cast<P *>(reinterpret_cast<T *>(p.grad) + outer * p.grad_so + (long long)it.row * p.grad_sr + d0) = res;
        }
    }
}

template <typename T>
inline hipError_t launch_gather_bwd_t(int sum, int mul, bool rel_grad, const GatherBwdParams &p, int grid, hipStream_t s) {
#define ULTRA_GB(S, M)                                                                                                     \
    if (sum == S && mul == M) {                                                                                            \
        if (rel_grad)                                                                                                      \
            hipLaunchKernelGGL((rspmm_minmax_bwd_gather_kernel<T, S, M, true>), dim3(grid), dim3(256), 0, s, p);           \
        else                                                                                                               \
            hipLaunchKernelGGL((rspmm_minmax_bwd_gather_kernel<T, S, M, false>), dim3(grid), dim3(256), 0, s, p);          \
        return hipGetLastError();                                                                                          \
    }
    ULTRA_GB(1, 0) ULTRA_GB(1, 1) ULTRA_GB(2, 0) ULTRA_GB(2, 1)
#undef ULTRA_GB
    return hipErrorInvalidValue;
}

template <typename T, int VEC>
inline hipError_t launch_edge_t(int sum, int mul, bool want_ri, const EdgeParams &p, hipStream_t s) {
    const long long groups = p.num_edge;
    const int blocks = (int)std::min<long long>((groups * 16 + 255) / 256, 16384);
#define ULTRA_EDGE(S, M)                                                                                              \
    if (sum == S && mul == M) {                                                                                       \
        if (want_ri)                                                                                                  \
            hipLaunchKernelGGL((rspmm_edge_bwd_kernel<T, VEC, S, M, true>), dim3(blocks), dim3(256), 0, s, p);        \
        else                                                                                                          \
            hipLaunchKernelGGL((rspmm_edge_bwd_kernel<T, VEC, S, M, false>), dim3(blocks), dim3(256), 0, s, p);       \
        return hipGetLastError();                                                                                     \
    }
    ULTRA_EDGE(0, 0) ULTRA_EDGE(0, 1) ULTRA_EDGE(1, 0) ULTRA_EDGE(1, 1) ULTRA_EDGE(2, 0) ULTRA_EDGE(2, 1)
#undef ULTRA_EDGE
    return hipErrorInvalidValue;
}

inline int launch_edge_kernel(int dtype, int vec, int sum, int mul, bool want_ri, const EdgeParams &p, hipStream_t s) {
    hipError_t e;
    if (dtype == ULTRA_F32)
        e = vec == 4 ? launch_edge_t<float, 4>(sum, mul, want_ri, p, s) : launch_edge_t<float, 1>(sum, mul, want_ri, p, s);
    else
        e = vec == 4 ? launch_edge_t<double, 4>(sum, mul, want_ri, p, s) : launch_edge_t<double, 1>(sum, mul, want_ri, p, s);
    if (e != hipSuccess) {
        set_error(std::string("rspmm_edge_bwd_kernel launch: ") + hipGetErrorString(e));
        return ULTRA_ERR_HIP;
    }
    return ULTRA_OK;
}

}  // namespace ultra
