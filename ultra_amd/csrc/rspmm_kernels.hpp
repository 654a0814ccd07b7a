// CDNA4 (gfx950, wave64) relational-SpMM kernels.  Device code + per-variant launchers.
//
//   out[row, :] = NARY_{e in row} w_e * BINARY(rel[type_e, :], x[col_e, :])      (rspmm.cpp:50-75)
//
// Mapping to the hardware
//   * The feature axis is cut into SPANs of 16*VEC elements (64 fp32 = 256 B = two 128-B lines).
//     A 16-lane group covers one span with one VEC-wide (16-B) load per lane, so every gather of a
//     source row is a fully coalesced 256-B segment; a wave64 carries four such groups.
//   * One workgroup (16 waves) works on ONE span at a time and stages that span's slice of the
//     relation table (R x 256 B) in LDS (MODE_REL_LDS); when the source matrix slice also fits
//     (N x 256 B <= LDS, the relation-graph regime: few hundred nodes, thousands of edges each) it
//     is staged too (MODE_ALL_LDS) and the whole gather runs out of LDS.  ds_read_b128 of 256-B
//     rows by 16-lane groups is bank-conflict free.
//   * span -> workgroup mapping is blockIdx % n_span, i.e. with 8 spans (batch 8 x dim 64) span s
//     runs on XCD s (block b is dispatched to XCD b % 8): each XCD's private 4-MiB L2 only ever
//     sees its own 1/8 slice of x / out.
//   * Work inside a span is the plan's static unit list (plan.hpp): a wave takes units
//     u = worker, worker + n_workers, ...  Group items are walked sequentially in sorted edge order
//     (bit-reproducible, oracle order); wave items stride their edges over the four groups and
//     combine with two cross-group shuffles.  Rows longer than seg_len were split by the plan and
//     are combined in slot order by rspmm_fixup_kernel: no atomics anywhere, run-to-run
//     deterministic.
//   * Edge records are fetched 16 per group with one coalesced load (next batch prefetched) and
//     broadcast inside the group with ds_bpermute; UNROLL source-row loads are in flight per lane.
//   * Arithmetic is compiled with fp-contract off: products and sums round separately exactly like
//     the reference's scalar CPU loop, so sequential (group) items match the oracle bit for bit.
#pragma once

#include <hip/hip_runtime.h>

#include <limits>
#include <type_traits>

#include "plan.hpp"

#pragma clang fp contract(off)

namespace ultra {

enum { MODE_GLOBAL = 0, MODE_REL_LDS = 1, MODE_ALL_LDS = 2 };
// BINARY variants: MUL/ADD are the reference's (operator.cuh:13-41); LHS/RHS pass one operand
// through and serve the backward passes (d/d input of rel+in is 1, etc.).
// BIN_MUL_TYPED: add_mul over a ULTRA_PLAN_TYPE_RUNS plan -- every item holds one relation, so the walk sums
// the sources and the relation vector is applied once per item (its own kernel, no extra code in BIN_MUL).
enum { BIN_MUL = 0, BIN_ADD = 1, BIN_LHS = 2, BIN_RHS = 3, BIN_MUL_TYPED = 4 };

#ifndef ULTRA_UNROLL
#define ULTRA_UNROLL 4
#endif
#ifndef ULTRA_UNROLL_LDS
#define ULTRA_UNROLL_LDS 4
#endif

struct MatArg {
    const void *ptr;
    long long stride_outer;
    long long stride_row;
};

struct FwdParams {
    const int32_t *col;
    const int32_t *type;
    const uint32_t *packed;
    const void *w_sorted;
    const Item *items;
    int32_t n_w, n_item, n_unit;
    MatArg rel, x, bnd;
    const long long *bnd_rows;   // point boundary: bnd holds ONE row per outer slice, added at row bnd_rows[outer] only
    void *out;
    long long out_stride_outer, out_stride_row;
    void *partial;
    int32_t n_outer, row_len, spans_per_outer, n_span;
    int32_t num_rel, num_in;
    int32_t type_bits;
    int32_t unit_w, packed_on, has_bnd;
    int32_t keep_mode;           // the weight stream is a 0/1 keep mask: a dropped edge is absent (matters for min / max)
    int32_t smod, nparts;
    uint32_t x_row_bytes, rel_row_bytes;   // row strides in bytes (each operand slice is < 4 GiB)
};

struct FixupParams {
    const int32_t *split_row;
    const int32_t *split_ptr;
    int32_t n_split;
    const void *partial;
    MatArg bnd;
    const long long *bnd_rows;
    void *out;
    long long out_stride_outer, out_stride_row;
    int32_t n_outer, row_len, has_bnd;
};

template <typename T, int VEC>
struct alignas(sizeof(T) * VEC > 16 ? 16 : sizeof(T) * VEC) Pack {
    T v[VEC];
};

template <typename T, int SUM>
__device__ __forceinline__ T nary_zero() {
    if (SUM == ULTRA_SUM_ADD) return T(0);
    if (SUM == ULTRA_SUM_MIN) return std::numeric_limits<T>::max();
    return std::numeric_limits<T>::lowest();
}

// min / max as ONE instruction.  The reference's `result < x ? result : x` (operator.cuh:58,71) compiles to v_cmp + v_cndmask (+ a
// wait state between them) -- three issue slots and twice the latency per link of a chain row's 9,000 dependent links: the chain
// phase of the max-aggregate kernels ran 40 % longer than the sum's (50 k against 35 k cycles, round 5).  v_min / v_max return the
// same value for every non-NaN pair but for the sign of a zero (max(-0, +0) = +0 where the ternary keeps its second operand), which
// torch.equal and every later product ignore -- the generated stream walk has used them since round 3.
__device__ __forceinline__ float hw_min(float a, float b) {
    float r;
    asm("v_min_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float hw_max(float a, float b) {
    float r;
    asm("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double hw_min(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double hw_max(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <typename T, int SUM>
__device__ __forceinline__ T nary(T result, T x) {  // operator.cuh:45,58,71
    if (SUM == ULTRA_SUM_ADD) return result + x;
    if (SUM == ULTRA_SUM_MIN) return hw_min(result, x);
    return hw_max(result, x);
}

template <typename T, int MUL>
__device__ __forceinline__ T binary(T rel, T x) {  // operator.cuh:15,29
    if (MUL == BIN_MUL) return rel * x;
    if (MUL == BIN_ADD) return rel + x;
    if (MUL == BIN_LHS) return rel;
    return x;
}

// Edge weight applied to a message (rspmm.cpp:68).  keep_mode: the weights are a 0/1 keep mask and a dropped edge must
// be ABSENT -- under min / max that is the identity of the reduction, not the value 0 a zero weight would produce
// (edge dropout of base_nbfnet.py:54-77 without rebuilding the graph); under add the two coincide.
template <typename T, int SUM, typename V>
__device__ __forceinline__ V weigh(const V y, const T w, const int keep_mode) {
    if (SUM != ULTRA_SUM_ADD && keep_mode) return w != T(0) ? y : V(nary_zero<T, SUM>());
    return y * V(w);
}

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Register-side view of a Pack: an ext vector, so that the elementwise math maps onto the packed
// fp32 VALU (v_pk_mul_f32 / v_pk_add_f32 on adjacent register pairs) without shuffling moves.
template <typename T, int VEC>
struct VecOf {
    using type = T __attribute__((ext_vector_type(VEC)));
};
template <typename T>
struct VecOf<T, 1> {
    using type = T;
};

template <typename T, int VEC>
__device__ __forceinline__ typename VecOf<T, VEC>::type to_vec(const Pack<T, VEC> &p) {
    typename VecOf<T, VEC>::type v;
    if constexpr (VEC == 1) {
        v = p.v[0];
    } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] = p.v[e];
    }
    return v;
}

template <typename T, int VEC>
__device__ __forceinline__ Pack<T, VEC> to_pack(const typename VecOf<T, VEC>::type &v) {
    Pack<T, VEC> p;
    if constexpr (VEC == 1) {
        p.v[0] = v;
    } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) p.v[e] = v[e];
    }
    return p;
}

template <typename V, int SUM>
__device__ __forceinline__ V nary_vec(V result, V x) {
    if constexpr (SUM == ULTRA_SUM_ADD) return result + x;
    else if constexpr (SUM == ULTRA_SUM_MIN) return __builtin_elementwise_min(result, x);
    else return __builtin_elementwise_max(result, x);
}

template <typename V, int MUL>
__device__ __forceinline__ V binary_vec(V rel, V x) {
    if constexpr (MUL == BIN_MUL) return rel * x;
    else if constexpr (MUL == BIN_MUL_TYPED) return x;
    else if constexpr (MUL == BIN_ADD) return rel + x;
    else if constexpr (MUL == BIN_LHS) return rel;
    else return x;
}


// Copies the span-wide column slice [inner * SPAN, +SPAN) of `rows` rows into LDS (row-major, SPAN
// elements per row).  Four 16-byte loads are in flight per thread before the first LDS write.
template <typename T, int VEC>
__device__ __forceinline__ void stage_slice(T *lds, const T *src, long long stride_row, int rows, int inner, int row_len,
                                            int tid, int nthreads) {
    constexpr int SPAN = 16 * VEC;
    using P = Pack<T, VEC>;
    const int total = rows * 16;
    for (int i0 = tid; i0 < total; i0 += 4 * nthreads) {
        P v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * nthreads;
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[u].v[e] = T(0);
            const int r = i >> 4, d = inner * SPAN + (i & 15) * VEC;
            if (i < total && d < row_len) v[u] = *reinterpret_cast<const P *>(src + (long long)r * stride_row + d);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * nthreads;
            if (i < total) *reinterpret_cast<P *>(lds + (i >> 4) * SPAN + (i & 15) * VEC) = v[u];
        }
    }
}

// Walks one group's edge stream {begin + k * stride : k < cnt} for nsteps (wave-uniform) steps and
// returns the group's accumulator.  PACKED: col/type share one word; UNITW: all edge weights are 1.
// Steps below nfull (wave-uniform) are valid for all four groups and run without per-lane predicates.
// Source rows are addressed as uniform base + 32-bit byte offset (one v_mad per edge, saddr loads).
template <typename T, int VEC, int SUM, int MUL, int MODE, bool PACKED, bool UNITW>
__device__ __forceinline__ Pack<T, VEC> walk_edges(const FwdParams &p, const int begin, const int cnt, const int stride,
                                                   const int nsteps, const int nfull, const int lane, const int l16,
                                                   const char *xbase, const char *relbase, const uint32_t lane_bytes,
                                                   const T *lds_x, const T *lds_rel) {
    constexpr int SPAN = 16 * VEC;
    constexpr bool TYPED = (MUL == BIN_MUL_TYPED);
    // edges per group and chunk: LDS-resident gathers have short latency and cheap registers -> deeper chunks
    constexpr int UNR = (MODE == MODE_ALL_LDS && sizeof(T) == 4) ? ULTRA_UNROLL_LDS : ULTRA_UNROLL;
    using P = Pack<T, VEC>;
    using V = typename VecOf<T, VEC>::type;
    V acc = V(nary_zero<T, SUM>());

    // record registers: this lane holds record (k0 + l16) of its group's stream
    uint32_t cur_c = 0, cur_t = 0;
    T cur_w = T(1);
    if (l16 < cnt) {
        const int idx = begin + l16 * stride;
        if (PACKED) {
            cur_c = p.packed[idx];
        } else {
            cur_c = (uint32_t)p.col[idx];
            cur_t = (uint32_t)p.type[idx];
        }
        if (!UNITW) cur_w = reinterpret_cast<const T *>(p.w_sorted)[idx];
    }
    const uint32_t tmask = (1u << p.type_bits) - 1u;
    const T *lds_x_lane = lds_x + l16 * VEC;
    const T *lds_rel_lane = lds_rel + l16 * VEC;

    // One chunk = ULTRA_UNROLL consecutive steps.  fetch() broadcasts the chunk's records inside each
    // group and issues its source-row loads; compute() consumes them.  The loop below keeps the loads
    // of chunk i + 1 in flight while chunk i is reduced (software pipelining: the compiler's vmcnt
    // bookkeeping only waits for the older chunk).
    struct Fetched {
        uint32_t t[UNR];
        T w[UNR];
        P xv[UNR];
    };
    auto fetch = [&](Fetched &f, const uint32_t rec_c, const uint32_t rec_t, const T rec_w, const int j) {
        uint32_t c[UNR];
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
            const int src = (lane & 48) | ((j + q) & 15);
            const uint32_t cc = (uint32_t)__shfl((int)rec_c, src);
            if (PACKED) {
                f.t[q] = cc & tmask;
                c[q] = cc >> p.type_bits;
            } else {
                c[q] = cc;
                f.t[q] = (uint32_t)__shfl((int)rec_t, src);
            }
            f.w[q] = UNITW ? T(1) : __shfl(rec_w, src);
        }
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
            if (MUL != BIN_LHS) {
                if (MODE == MODE_ALL_LDS)
                    f.xv[q] = *reinterpret_cast<const P *>(lds_x_lane + c[q] * SPAN);
                else
                    f.xv[q] = *reinterpret_cast<const P *>(xbase + (c[q] * p.x_row_bytes + lane_bytes));
            }
        }
    };
    auto compute = [&](auto pred_tag, const Fetched &f, const int kbase) {
        constexpr bool PRED = decltype(pred_tag)::value;
        P rv[UNR];
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
            if (MUL != BIN_RHS && !TYPED) {
                if (MODE >= MODE_REL_LDS)
                    rv[q] = *reinterpret_cast<const P *>(lds_rel_lane + f.t[q] * SPAN);
                else
                    rv[q] = *reinterpret_cast<const P *>(relbase + (f.t[q] * p.rel_row_bytes + lane_bytes));
            }
        }
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
            const V rr = (MUL != BIN_RHS && !TYPED) ? to_vec<T, VEC>(rv[q]) : V(T(0));
            const V xx = (MUL != BIN_LHS) ? to_vec<T, VEC>(f.xv[q]) : V(T(0));
            // TYPED items hold edges of ONE relation: sum the sources, multiply by rel[type] once at the end
            V y = TYPED ? xx : binary_vec<V, MUL>(rr, xx);
            if (!UNITW) y = weigh<T, SUM>(y, f.w[q], p.keep_mode);
            const V cand = nary_vec<V, SUM>(acc, y);
            if (PRED)
                acc = (kbase + q < cnt) ? cand : acc;
            else
                acc = cand;
        }
    };
    auto load_records = [&](uint32_t &rc, uint32_t &rt, T &rw, const int k0) {
        rc = 0;
        rt = 0;
        rw = T(1);
        const int k = k0 + l16;
        if (k0 < nsteps && k < cnt) {
            const int idx = begin + k * stride;
            if (PACKED) {
                rc = p.packed[idx];
            } else {
                rc = (uint32_t)p.col[idx];
                rt = (uint32_t)p.type[idx];
            }
            if (!UNITW) rw = reinterpret_cast<const T *>(p.w_sorted)[idx];
        }
    };

    constexpr int CHUNKS_PER_BATCH = 16 / UNR;
    const int nchunks = (nsteps + UNR - 1) / UNR;
    if (nchunks > 0) {
        uint32_t nxt_c, nxt_t;
        T nxt_w;
        load_records(nxt_c, nxt_t, nxt_w, 16);   // records of batch 1, needed CHUNKS_PER_BATCH chunks from now
        Fetched fa, fb;
        fetch(fa, cur_c, cur_t, cur_w, 0);
        for (int ci = 0; ci < nchunks; ci += 2) {
            // ---- even chunk: in fa; prefetch ci + 1 into fb ----
            {
                const int cn = ci + 1;
                if (cn < nchunks) {
                    if ((cn % CHUNKS_PER_BATCH) == 0) {
                        cur_c = nxt_c;
                        cur_t = nxt_t;
                        cur_w = nxt_w;
                        load_records(nxt_c, nxt_t, nxt_w, (cn / CHUNKS_PER_BATCH + 1) * 16);
                    }
                    fetch(fb, cur_c, cur_t, cur_w, (cn % CHUNKS_PER_BATCH) * UNR);
                }
                if ((ci + 1) * UNR <= nfull)
                    compute(std::false_type{}, fa, ci * UNR);
                else
                    compute(std::true_type{}, fa, ci * UNR);
            }
            // ---- odd chunk: in fb; prefetch ci + 2 into fa ----
            if (ci + 1 < nchunks) {
                const int cn = ci + 2;
                if (cn < nchunks) {
                    if ((cn % CHUNKS_PER_BATCH) == 0) {
                        cur_c = nxt_c;
                        cur_t = nxt_t;
                        cur_w = nxt_w;
                        load_records(nxt_c, nxt_t, nxt_w, (cn / CHUNKS_PER_BATCH + 1) * 16);
                    }
                    fetch(fa, cur_c, cur_t, cur_w, (cn % CHUNKS_PER_BATCH) * UNR);
                }
                if ((ci + 2) * UNR <= nfull)
                    compute(std::false_type{}, fb, (ci + 1) * UNR);
                else
                    compute(std::true_type{}, fb, (ci + 1) * UNR);
            }
        }
    }
    return to_pack<T, VEC>(acc);
}

template <typename T, int VEC, int SUM, int MUL, int MODE>
__global__ void __launch_bounds__(1024) rspmm_fwd_kernel(const FwdParams p) {
    constexpr int SPAN = 16 * VEC;
    using P = Pack<T, VEC>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *lds_rel = reinterpret_cast<T *>(smem);
    T *lds_x = lds_rel + (MUL == BIN_RHS ? 0 : (size_t)p.num_rel * SPAN);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = rfl(tid >> 6);
    const int nwave = blockDim.x >> 6;
    const int grp = lane >> 4;
    const int l16 = lane & 15;
    const int part = blockIdx.x / p.smod;
    if (part >= p.nparts) return;
    const int4 *items4 = reinterpret_cast<const int4 *>(p.items);

    for (int span = blockIdx.x % p.smod; span < p.n_span; span += p.smod) {
        const int outer = span / p.spans_per_outer;
        const int inner = span - outer * p.spans_per_outer;
        const int d0 = inner * SPAN + l16 * VEC;
        const bool dvalid = d0 < p.row_len;
        const int d0c = dvalid ? d0 : 0;
        const char *xbase = reinterpret_cast<const char *>(reinterpret_cast<const T *>(p.x.ptr) + outer * p.x.stride_outer);
        const char *relbase =
            reinterpret_cast<const char *>(reinterpret_cast<const T *>(p.rel.ptr) + outer * p.rel.stride_outer);
        const uint32_t lane_bytes = (uint32_t)d0c * (uint32_t)sizeof(T);
        const long long bnd_row = p.bnd_rows ? p.bnd_rows[outer] : -1;   // (stride_row of a point boundary is 0)

        if (MODE >= MODE_REL_LDS) {
            __syncthreads();  // readers of the previous span are done with the LDS image
            if (MUL != BIN_RHS)
                stage_slice<T, VEC>(lds_rel, reinterpret_cast<const T *>(p.rel.ptr) + outer * p.rel.stride_outer,
                                    p.rel.stride_row, p.num_rel, inner, p.row_len, tid, blockDim.x);
            if (MODE == MODE_ALL_LDS && MUL != BIN_LHS)
                stage_slice<T, VEC>(lds_x, reinterpret_cast<const T *>(p.x.ptr) + outer * p.x.stride_outer,
                                    p.x.stride_row, p.num_in, inner, p.row_len, tid, blockDim.x);
            __syncthreads();
        }

        for (int u = part * nwave + wave; u < p.n_unit; u += p.nparts * nwave) {
            const bool wmode = u < p.n_w;  // wave-uniform
            int begin, cnt, stride, row, slot, nsteps, nfull;
            if (wmode) {
                const int4 it = items4[u];
                row = rfl(it.x);
                const int ibegin = rfl(it.y), ilen = rfl(it.z);
                slot = rfl(it.w);
                begin = ibegin + grp;
                stride = 4;
                cnt = (ilen - grp + 3) >> 2;
                nsteps = (ilen + 3) >> 2;
                nfull = ilen >> 2;
            } else {
                const int q = p.n_w + 4 * (u - p.n_w) + grp;
                if (q < p.n_item) {
                    const int4 it = items4[q];
                    row = it.x;
                    begin = it.y;
                    cnt = it.z;
                    slot = it.w;
                } else {
                    row = -1;
                    begin = 0;
                    cnt = 0;
                    slot = -1;
                }
                stride = 1;
                const int m01 = max(__shfl(cnt, 0), __shfl(cnt, 16));
                const int m23 = max(__shfl(cnt, 32), __shfl(cnt, 48));
                nsteps = rfl(max(m01, m23));
                const int n01 = min(__shfl(cnt, 0), __shfl(cnt, 16));
                const int n23 = min(__shfl(cnt, 32), __shfl(cnt, 48));
                nfull = rfl(min(n01, n23));
            }

            P acc;
#define ULTRA_WALK(PK, UW)                                                                                       \
    acc = walk_edges<T, VEC, SUM, MUL, MODE, PK, UW>(p, begin, cnt, stride, nsteps, nfull, lane, l16, xbase, relbase, \
                                                     lane_bytes, lds_x, lds_rel)
            constexpr bool typed = (MUL == BIN_MUL_TYPED);
            if (p.packed_on) {
                if (p.unit_w) ULTRA_WALK(true, true); else ULTRA_WALK(true, false);
            } else {
                if (p.unit_w) ULTRA_WALK(false, true); else ULTRA_WALK(false, false);
            }
#undef ULTRA_WALK

            if (wmode) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    T v = acc.v[e];
                    v = nary<T, SUM>(v, __shfl_xor(v, 16));
                    v = nary<T, SUM>(v, __shfl_xor(v, 32));
                    acc.v[e] = v;
                }
            }
            const bool writer = wmode ? (grp == 0) : (row >= 0);
            if (typed && writer && dvalid) {
                // one relation per item: y = rel[type] (x) sum_e w_e x_e   (distributivity of mul over add)
                const int first = wmode ? (begin - grp) : begin;
                uint32_t t = 0;
                if (wmode || cnt > 0) t = p.packed_on ? (p.packed[first] & ((1u << p.type_bits) - 1u)) : (uint32_t)p.type[first];
                P r;
                if (MODE >= MODE_REL_LDS)
                    r = *reinterpret_cast<const P *>(lds_rel + t * SPAN + l16 * VEC);
                else
                    r = *reinterpret_cast<const P *>(relbase + (t * p.rel_row_bytes + lane_bytes));
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc.v[e] = r.v[e] * acc.v[e];
            }
            if (writer && dvalid) {
                if (slot >= 0) {
                    T *dst = reinterpret_cast<T *>(p.partial) + ((long long)slot * p.n_outer + outer) * p.row_len + d0;
                    *reinterpret_cast<P *>(dst) = acc;
                } else {
                    if (p.has_bnd && (bnd_row < 0 || bnd_row == row)) {
                        const P b = *reinterpret_cast<const P *>(reinterpret_cast<const T *>(p.bnd.ptr) +
                                                                 outer * p.bnd.stride_outer +
                                                                 (long long)row * p.bnd.stride_row + d0);
#pragma unroll
                        for (int e = 0; e < VEC; ++e) acc.v[e] = nary<T, SUM>(acc.v[e], b.v[e]);
                    }
                    T *dst = reinterpret_cast<T *>(p.out) + outer * p.out_stride_outer +
                             (long long)row * p.out_stride_row + d0;
                    *reinterpret_cast<P *>(dst) = acc;
                }
            }
        }
    }
}

// Combines the partial results of split rows and applies the boundary epilogue.  A workgroup handles 16 output vectors
// with 16 slot lanes each: slot lane s sums the row's partials s, s + 16, s + 32, ... in order, the 16 lane sums are folded
// 0, 1, ..., 15 -- a fixed association (deterministic, no atomics), wide enough for relation-major plans whose rows hold
// hundreds of slots (tests/helpers.py: emulate_plan_forward restates it).
constexpr int FIXUP_SLOT_LANES = 16;
template <typename T, int VEC, int SUM>
__global__ void __launch_bounds__(256) rspmm_fixup_kernel(const FixupParams p) {
    using P = Pack<T, VEC>;
    __shared__ P lds[FIXUP_SLOT_LANES][16];
    const int v = threadIdx.x & 15, s = threadIdx.x >> 4;
    const int vec_per_row = p.row_len / VEC;
    const long long total = (long long)p.n_split * p.n_outer * vec_per_row;
    for (long long base = blockIdx.x * 16ll; base < total; base += gridDim.x * 16ll) {
        const bool ok = base + v < total;
        const long long i = ok ? base + v : total - 1;
        const int dv = (int)(i % vec_per_row);
        const long long r2 = i / vec_per_row;
        const int outer = (int)(r2 % p.n_outer);
        const int k = (int)(r2 / p.n_outer);
        const int row = p.split_row[k];
        const int s0 = p.split_ptr[k], s1 = p.split_ptr[k + 1];
        const int d0 = dv * VEC;
        P acc;
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc.v[e] = nary_zero<T, SUM>();
        // four partials in flight per thread
        for (int sb = s0 + s; sb < s1; sb += 4 * FIXUP_SLOT_LANES) {
            P pv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int sl = sb + u * FIXUP_SLOT_LANES < s1 ? sb + u * FIXUP_SLOT_LANES : sb;
                pv[u] = *reinterpret_cast<const P *>(reinterpret_cast<const T *>(p.partial) +
                                                     ((long long)sl * p.n_outer + outer) * p.row_len + d0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (sb + u * FIXUP_SLOT_LANES < s1) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc.v[e] = nary<T, SUM>(acc.v[e], pv[u].v[e]);
                }
            }
        }
        lds[s][v] = acc;
        __syncthreads();
        if (s == 0 && ok) {
#pragma unroll
            for (int q = 1; q < FIXUP_SLOT_LANES; ++q) {
                const P o = lds[q][v];
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc.v[e] = nary<T, SUM>(acc.v[e], o.v[e]);
            }
            if (p.has_bnd && (!p.bnd_rows || p.bnd_rows[outer] == row)) {
                const P b = *reinterpret_cast<const P *>(reinterpret_cast<const T *>(p.bnd.ptr) +
                                                         outer * p.bnd.stride_outer + (long long)row * p.bnd.stride_row + d0);
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc.v[e] = nary<T, SUM>(acc.v[e], b.v[e]);
            }
            T *dst = reinterpret_cast<T *>(p.out) + outer * p.out_stride_outer + (long long)row * p.out_stride_row + d0;
            *reinterpret_cast<P *>(dst) = acc;
        }
        __syncthreads();
    }
}

// w_sorted[k] = w[perm[k]]: per-call edge weights arrive in original edge order.
template <typename T>
__global__ void __launch_bounds__(256) permute_weight_kernel(const T *w, const int32_t *perm, T *w_sorted, int64_t n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        w_sorted[i] = w[perm[i]];
}

// ---- per-variant launchers (explicitly instantiated in rspmm_variant_*.hip) ----
template <typename T, int VEC, int SUM, int MUL, int MODE>
inline hipError_t launch_one(const FwdParams &p, int grid, int threads, size_t lds, hipStream_t s) {
    auto kern = rspmm_fwd_kernel<T, VEC, SUM, MUL, MODE>;
    // LDS opt-in above 48 KiB: raised once per kernel to the largest size seen (not a stream operation,
    // and kept out of hipGraph capture: warm-up launches have already done it)
    static size_t lds_opted_in = 0;
    if (lds > 48 * 1024 && lds > lds_opted_in) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_opted_in = lds;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, s, p);
    return hipGetLastError();
}

template <typename T, int VEC, int MODE>
hipError_t launch_fwd_variant(int sum, int mul, const FwdParams &p, int grid, int threads, size_t lds, hipStream_t s);

#define ULTRA_CASE(S, M)                                                        \
    case (S) * 4 + (M):                                                         \
        return launch_one<T, VEC, S, M, MODE>(p, grid, threads, lds, s);

#define ULTRA_DEFINE_VARIANT(T_, VEC_, MODE_)                                                                     \
    template <>                                                                                                   \
    hipError_t launch_fwd_variant<T_, VEC_, MODE_>(int sum, int mul, const FwdParams &p, int grid, int threads,  \
                                                   size_t lds, hipStream_t s) {                                   \
        using T = T_;                                                                                             \
        constexpr int VEC = VEC_;                                                                                 \
        constexpr int MODE = MODE_;                                                                               \
        switch (mul == BIN_MUL_TYPED ? 100 : sum * 4 + mul) {                                                     \
            ULTRA_CASE(0, 0) ULTRA_CASE(0, 1) ULTRA_CASE(0, 2) ULTRA_CASE(0, 3)                                   \
            ULTRA_CASE(1, 0) ULTRA_CASE(1, 1) ULTRA_CASE(1, 2) ULTRA_CASE(1, 3)                                   \
            ULTRA_CASE(2, 0) ULTRA_CASE(2, 1) ULTRA_CASE(2, 2) ULTRA_CASE(2, 3)                                   \
            case 100: return launch_one<T, VEC, 0, BIN_MUL_TYPED, MODE>(p, grid, threads, lds, s);                 \
        }                                                                                                         \
        return hipErrorInvalidValue;                                                                              \
    }

}  // namespace ultra
