// Relational SpMM in the REFERENCE'S SUMMATION ORDER (ULTRA_PLAN_EXACT_ORDER plans), CDNA4 / wave64.
//
//   out[row, :] = NARY_{e in row, in sorted (row, col, edge id) order} w_e * BINARY(rel[type_e, :], x[col_e, :])
//
// rspmm.cpp:61-72 adds the messages of a row one after the other; fp32 addition does not associate, so scores (and
// the rankings derived from them) only reproduce if the GPU adds in that same order.  Two walks do it:
//
//   group items  (rows of at most chain_min edges, four per wave): a 16-lane group walks its row edge by edge with its
//                span's 64 accumulators in registers (16 B per lane) -- the order is the sorted edge order by construction.
//                Records (col, type) are loaded 16 steps at a time, one per lane, and broadcast inside the group with
//                ds_swizzle (no address VALU); source rows are gathered with uniform-base + 32-bit-offset 16-byte loads,
//                two 4-step chunks in flight; relation rows come from the LDS image of the span's relation slice.
//   chain rows   (longer rows; a power-law graph has a few with thousands of edges): walked by the whole workgroup.
//                Waves 1..15 are producers: each 16-lane group computes ONE message per chunk of 60 edges (record and
//                source row prefetched two chunks ahead) and parks it in an LDS ring slot; wave 0 is the consumer: lane l
//                owns element l of the span and adds the 60 parked messages in slot order, ds_read_b32 + one dependent
//                add per edge.  One workgroup barrier per chunk.  The serial part of a 9,000-edge row is thus
//                9,000 dependent adds (~6 cycles each) instead of 9,000 walk steps of a single lane group (~200).
//
// Work is assigned statically (plan.cpp build_schedule: longest-processing-time-first over chain rows and group units
// for the launch's workgroups-per-span), chain rows first.  Every output row is written exactly once by the lanes
// that summed it: no partial slots, no fix-up launch, no scratch memory -- a plan is immutable after upload and can be
// shared by any number of streams.
#pragma once

#include "rspmm_kernels.hpp"

#pragma clang fp contract(off)

namespace ultra {

constexpr int ORDER_THREADS = 1024;
static_assert(CHAIN_SLOTS == 4 * (ORDER_THREADS / 64 - 1), "one ring slot per producer group");

struct OrderParams {
    const int32_t *rec;       // (col, type) per sorted edge
    const int32_t *perm;      // sorted position -> original edge id
    const void *w;            // edge weights in ORIGINAL edge order, or NULL (all ones)
    const int4 *items;        // {row, begin, len, -}; chain rows first, group items from n_chain on
    const int32_t *unit_ptr, *units, *chunk_ptr;
    const int4 *chunks;       // {row, begin, count, flags}
    int32_t n_chain, n_item;
    MatArg rel, x, bnd;
    const long long *bnd_rows;   // point boundary: bnd holds ONE row per outer slice, added at row bnd_rows[outer] only
    void *out;
    long long out_stride_outer, out_stride_row;
    int32_t n_outer, row_len, spans_per_outer, n_span;
    int32_t num_rel, has_bnd, has_chain;
    int32_t smod, nparts;
    uint32_t x_row_bytes, rel_row_bytes;
};

// value of lane K of each 16-lane row, in every lane of that row (ds_swizzle bit mode: lane' = (lane & 0x10) | K within
// each half wave) -- crossbar only, no LDS memory, no address register
template <int K>
__device__ __forceinline__ int bcast16(int v) {
    return __builtin_amdgcn_ds_swizzle(v, 0x10 | (K << 5));
}
template <int K>
__device__ __forceinline__ float bcast16(float v) {
    return __int_as_float(bcast16<K>(__float_as_int(v)));
}
template <int K>
__device__ __forceinline__ double bcast16(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = bcast16<K>((int)(b & 0xffffffffll)), hi = bcast16<K>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <int J>
using StepTag = std::integral_constant<int, J>;

// ---- group items: four rows per wave, each walked sequentially by one 16-lane group ----
template <typename T, int SUM, int MUL, bool REL_LDS, bool WEIGHTED>
__device__ __forceinline__ Pack<T, 4> walk_row_in_order(const OrderParams &p, const int begin, const int cnt, const int nsteps,
                                                        const int nfull, const int l16, const char *xbase, const char *relbase,
                                                        const uint32_t lane_bytes, const T *lds_rel_lane) {
    constexpr int SPAN = 64;
    using P = Pack<T, 4>;
    using V = typename VecOf<T, 4>::type;
    V acc = V(nary_zero<T, SUM>());
    const T *wt = reinterpret_cast<const T *>(p.w);

    struct Records {   // this lane holds the record of step (batch base + l16) of its group's row
        int c, t;
        T w;
    };
    const auto load_records = [&](Records &r, const int k0) {
        r.c = 0;
        r.t = 0;
        r.w = T(1);
        const int k = k0 + l16;
        if (k0 < nsteps && k < cnt) {
            const int2 ct = *reinterpret_cast<const int2 *>(p.rec + 2 * (size_t)(begin + k));
            r.c = ct.x;
            r.t = ct.y;
            if (WEIGHTED) r.w = wt[p.perm[begin + k]];
        }
    };
    struct Fetched {   // one chunk = 4 consecutive steps
        int t[4];
        T w[4];
        P xv[4];
    };
    const auto fetch = [&](auto jtag, Fetched &f, const Records &r) {
        constexpr int J = decltype(jtag)::value;
        int c[4];
        c[0] = bcast16<J + 0>(r.c), c[1] = bcast16<J + 1>(r.c), c[2] = bcast16<J + 2>(r.c), c[3] = bcast16<J + 3>(r.c);
        if (MUL != BIN_RHS) {
            f.t[0] = bcast16<J + 0>(r.t), f.t[1] = bcast16<J + 1>(r.t), f.t[2] = bcast16<J + 2>(r.t), f.t[3] = bcast16<J + 3>(r.t);
        }
        if (WEIGHTED) {
            f.w[0] = bcast16<J + 0>(r.w), f.w[1] = bcast16<J + 1>(r.w), f.w[2] = bcast16<J + 2>(r.w), f.w[3] = bcast16<J + 3>(r.w);
        }
        if (MUL != BIN_LHS) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                f.xv[q] = *reinterpret_cast<const P *>(xbase + (__umul24((uint32_t)c[q], p.x_row_bytes) + lane_bytes));
        }
    };
    const auto compute = [&](auto pred_tag, const Fetched &f, const int kbase) {
        constexpr bool PRED = decltype(pred_tag)::value;
        P rv[4];
        if (MUL != BIN_RHS) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (REL_LDS)
                    rv[q] = *reinterpret_cast<const P *>(lds_rel_lane + f.t[q] * SPAN);
                else
                    rv[q] = *reinterpret_cast<const P *>(relbase + (__umul24((uint32_t)f.t[q], p.rel_row_bytes) + lane_bytes));
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const V rr = (MUL != BIN_RHS) ? to_vec<T, 4>(rv[q]) : V(T(0));
            const V xx = (MUL != BIN_LHS) ? to_vec<T, 4>(f.xv[q]) : V(T(0));
            V y = binary_vec<V, MUL>(rr, xx);
            if (WEIGHTED) y = V(f.w[q]) * y;            // w * x, rspmm.cpp:68
            const V cand = nary_vec<V, SUM>(acc, y);
            if (PRED)
                acc = (kbase + q < cnt) ? cand : acc;
            else
                acc = cand;
        }
    };
    const auto reduce = [&](const Fetched &f, const int kbase) {
        if (kbase + 4 <= nfull)
            compute(std::false_type{}, f, kbase);
        else
            compute(std::true_type{}, f, kbase);
    };

    if (nsteps > 0) {
        Records cur, nxt;
        load_records(cur, 0);
        load_records(nxt, 16);
        Fetched fa, fb;
        fetch(StepTag<0>{}, fa, cur);
        for (int b0 = 0; b0 < nsteps; b0 += 16) {
            // the loads of chunk i + 1 are issued before chunk i is reduced; the next batch's records were requested a
            // whole batch (16 steps) ago
            if (b0 + 4 < nsteps) fetch(StepTag<4>{}, fb, cur);
            reduce(fa, b0);
            if (b0 + 8 < nsteps) fetch(StepTag<8>{}, fa, cur);
            if (b0 + 4 < nsteps) reduce(fb, b0 + 4);
            if (b0 + 12 < nsteps) fetch(StepTag<12>{}, fb, cur);
            if (b0 + 8 < nsteps) reduce(fa, b0 + 8);
            cur = nxt;
            load_records(nxt, b0 + 32);
            if (b0 + 16 < nsteps) fetch(StepTag<0>{}, fa, cur);
            if (b0 + 12 < nsteps) reduce(fb, b0 + 12);
        }
    }
    return to_pack<T, 4>(acc);
}

// ---- consumer side of a chain chunk: acc (+)= ring[0], ring[1], ... in slot order ----
// The adds form one dependent chain per lane; the LDS reads are independent of it and are kept a block ahead.
template <typename T, int SUM>
__device__ __forceinline__ T consume_chunk(T acc, const T *ring_lane, const int count) {
    constexpr int SPAN = 64;
    constexpr int BLK = 15;
    if (count == CHAIN_SLOTS) {
        T a[BLK], b[BLK];
#pragma unroll
        for (int k = 0; k < BLK; ++k) a[k] = ring_lane[(0 * BLK + k) * SPAN];
#pragma unroll
        for (int k = 0; k < BLK; ++k) b[k] = ring_lane[(1 * BLK + k) * SPAN];
#pragma unroll
        for (int k = 0; k < BLK; ++k) acc = nary<T, SUM>(acc, a[k]);
#pragma unroll
        for (int k = 0; k < BLK; ++k) a[k] = ring_lane[(2 * BLK + k) * SPAN];
#pragma unroll
        for (int k = 0; k < BLK; ++k) acc = nary<T, SUM>(acc, b[k]);
#pragma unroll
        for (int k = 0; k < BLK; ++k) b[k] = ring_lane[(3 * BLK + k) * SPAN];
#pragma unroll
        for (int k = 0; k < BLK; ++k) acc = nary<T, SUM>(acc, a[k]);
#pragma unroll
        for (int k = 0; k < BLK; ++k) acc = nary<T, SUM>(acc, b[k]);
        return acc;
    }
    int k = 0;
    for (; k + 4 <= count; k += 4) {
        const T v0 = ring_lane[(k + 0) * SPAN], v1 = ring_lane[(k + 1) * SPAN], v2 = ring_lane[(k + 2) * SPAN],
                v3 = ring_lane[(k + 3) * SPAN];
        acc = nary<T, SUM>(acc, v0);
        acc = nary<T, SUM>(acc, v1);
        acc = nary<T, SUM>(acc, v2);
        acc = nary<T, SUM>(acc, v3);
    }
    for (; k < count; ++k) acc = nary<T, SUM>(acc, ring_lane[k * SPAN]);
    return acc;
}

template <typename T, int SUM, int MUL, bool REL_LDS>
__global__ void __launch_bounds__(ORDER_THREADS) rspmm_order_kernel(const OrderParams p) {
    constexpr int SPAN = 64;
    using P = Pack<T, 4>;
    using V = typename VecOf<T, 4>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *lds_rel = reinterpret_cast<T *>(smem);
    T *ring = lds_rel + ((REL_LDS && MUL != BIN_RHS) ? (size_t)p.num_rel * SPAN : 0);   // [2][CHAIN_SLOTS][SPAN]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = rfl(tid >> 6);
    constexpr int nwave = ORDER_THREADS / 64;
    const int grp = lane >> 4;
    const int l16 = lane & 15;
    const int part = blockIdx.x / p.smod;
    if (part >= p.nparts) return;
    const T *wt = reinterpret_cast<const T *>(p.w);

    for (int span = blockIdx.x % p.smod; span < p.n_span; span += p.smod) {
        const int outer = span / p.spans_per_outer;
        const int inner = span - outer * p.spans_per_outer;
        const int d0 = inner * SPAN + l16 * 4;
        const bool dvalid = d0 < p.row_len;
        const int d0c = dvalid ? d0 : 0;
        const char *xbase = reinterpret_cast<const char *>(reinterpret_cast<const T *>(p.x.ptr) + outer * p.x.stride_outer);
        const char *relbase =
            reinterpret_cast<const char *>(reinterpret_cast<const T *>(p.rel.ptr) + outer * p.rel.stride_outer);
        const uint32_t lane_bytes = (uint32_t)d0c * (uint32_t)sizeof(T);
        const long long bnd_row = p.bnd_rows ? p.bnd_rows[outer] : -1;   // (stride_row of a point boundary is 0)
        const T *lds_rel_lane = lds_rel + l16 * 4;

        if (REL_LDS && MUL != BIN_RHS) {
            __syncthreads();  // readers of the previous span are done with the LDS image
            stage_slice<T, 4>(lds_rel, reinterpret_cast<const T *>(p.rel.ptr) + outer * p.rel.stride_outer, p.rel.stride_row,
                              p.num_rel, inner, p.row_len, tid, ORDER_THREADS);
            __syncthreads();
        }

        // ================= chain rows of this workgroup =================
        const int c0 = p.has_chain ? p.chunk_ptr[part] : 0, c1 = p.has_chain ? p.chunk_ptr[part + 1] : 0;
        if (c1 > c0) {
            const bool consumer = wave == 0;   // wave-uniform
            const int slot = (wave - 1) * 4 + grp;
            // producer pipeline registers: x0 = source row of chunk `it` (in flight since iteration it - 1),
            // r1 = record of chunk it + 1 (requested in iteration it - 1)
            struct Rec {
                int c, t;
                T w;
                bool valid;
            };
            const auto load_rec = [&](const int ci) {
                Rec r;
                r.c = 0, r.t = 0, r.w = T(1), r.valid = false;
                if (ci < c1) {
                    const int4 ch = p.chunks[ci];
                    if (slot < ch.z) {
                        const int e = ch.y + slot;
                        const int2 ct = *reinterpret_cast<const int2 *>(p.rec + 2 * (size_t)e);
                        r.c = ct.x, r.t = ct.y, r.valid = true;
                        if (wt) r.w = wt[p.perm[e]];
                    }
                }
                return r;
            };
            const auto gather = [&](const Rec &r) {
                P v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v.v[e] = T(0);
                if (MUL != BIN_LHS && r.valid)
                    v = *reinterpret_cast<const P *>(xbase + (__umul24((uint32_t)r.c, p.x_row_bytes) + lane_bytes));
                return v;
            };
            Rec r0, r1;
            P x0;
            T cacc = nary_zero<T, SUM>();
            if (!consumer) {
                r0 = load_rec(c0);
                r1 = load_rec(c0 + 1);
                x0 = gather(r0);
            }
            for (int it = c0; it <= c1; ++it) {
                if (!consumer) {
                    // issue next chunk's gather and the record after that, then finish this chunk
                    const P x1 = gather(r1);
                    const Rec r2 = load_rec(it + 2);
                    if (it < c1 && r0.valid) {
                        P rv;
                        if (MUL != BIN_RHS) {
                            if (REL_LDS)
                                rv = *reinterpret_cast<const P *>(lds_rel_lane + r0.t * SPAN);
                            else
                                rv = *reinterpret_cast<const P *>(relbase + (__umul24((uint32_t)r0.t, p.rel_row_bytes) + lane_bytes));
                        }
                        const V rr = (MUL != BIN_RHS) ? to_vec<T, 4>(rv) : V(T(0));
                        const V xx = (MUL != BIN_LHS) ? to_vec<T, 4>(x0) : V(T(0));
                        V y = binary_vec<V, MUL>(rr, xx);
                        if (wt) y = V(r0.w) * y;
                        *reinterpret_cast<P *>(ring + ((size_t)(it & 1) * CHAIN_SLOTS + slot) * SPAN + l16 * 4) = to_pack<T, 4>(y);
                    }
                    r0 = r1;
                    r1 = r2;
                    x0 = x1;
                } else if (it > c0) {
                    const int4 ch = p.chunks[it - 1];
                    const int row = rfl(ch.x), count = rfl(ch.z), flags = rfl(ch.w);
                    if (flags & CHUNK_FIRST) cacc = nary_zero<T, SUM>();
                    cacc = consume_chunk<T, SUM>(cacc, ring + (size_t)((it - 1) & 1) * CHAIN_SLOTS * SPAN + lane, count);
                    if (flags & CHUNK_LAST) {
                        const int d = inner * SPAN + lane;
                        if (d < p.row_len) {
                            T v = cacc;
                            if (p.has_bnd && (bnd_row < 0 || bnd_row == row))
                                v = nary<T, SUM>(v, reinterpret_cast<const T *>(p.bnd.ptr)[outer * p.bnd.stride_outer +
                                                                                          (long long)row * p.bnd.stride_row + d]);
                            reinterpret_cast<T *>(p.out)[outer * p.out_stride_outer + (long long)row * p.out_stride_row + d] = v;
                        }
                    }
                }
                __syncthreads();
            }
        }

        // ================= group units of this workgroup =================
        const int u1 = p.unit_ptr[part + 1];
        for (int ui = p.unit_ptr[part] + wave; ui < u1; ui += nwave) {
            const int u = p.units[ui];
            const int q = p.n_chain + 4 * u + grp;
            int row = -1, begin = 0, cnt = 0;
            if (q < p.n_item) {
                const int4 it = p.items[q];
                row = it.x;
                begin = it.y;
                cnt = it.z;
            }
            const int m01 = max(__shfl(cnt, 0), __shfl(cnt, 16));
            const int m23 = max(__shfl(cnt, 32), __shfl(cnt, 48));
            const int nsteps = rfl(max(m01, m23));
            const int n01 = min(__shfl(cnt, 0), __shfl(cnt, 16));
            const int n23 = min(__shfl(cnt, 32), __shfl(cnt, 48));
            const int nfull = rfl(min(n01, n23));

            P acc;
            if (p.w)
                acc = walk_row_in_order<T, SUM, MUL, REL_LDS, true>(p, begin, cnt, nsteps, nfull, l16, xbase, relbase, lane_bytes,
                                                                    lds_rel_lane);
            else
                acc = walk_row_in_order<T, SUM, MUL, REL_LDS, false>(p, begin, cnt, nsteps, nfull, l16, xbase, relbase, lane_bytes,
                                                                     lds_rel_lane);
            if (row >= 0 && dvalid) {
                if (p.has_bnd && (bnd_row < 0 || bnd_row == row)) {
                    const P b = *reinterpret_cast<const P *>(reinterpret_cast<const T *>(p.bnd.ptr) + outer * p.bnd.stride_outer +
                                                             (long long)row * p.bnd.stride_row + d0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc.v[e] = nary<T, SUM>(acc.v[e], b.v[e]);
                }
                T *dst = reinterpret_cast<T *>(p.out) + outer * p.out_stride_outer + (long long)row * p.out_stride_row + d0;
                *reinterpret_cast<P *>(dst) = acc;
            }
        }
    }
}

// ---- per-variant launchers (explicitly instantiated in rspmm_order_*.hip) ----
template <typename T, int SUM, int MUL, bool REL_LDS>
inline hipError_t launch_order_one(const OrderParams &p, int grid, size_t lds, hipStream_t s) {
    auto kern = rspmm_order_kernel<T, SUM, MUL, REL_LDS>;
    static size_t lds_opted_in = 0;   // (see launch_one in rspmm_kernels.hpp)
    if (lds > 48 * 1024 && lds > lds_opted_in) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_opted_in = lds;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(ORDER_THREADS), lds, s, p);
    return hipGetLastError();
}

template <typename T, bool REL_LDS>
hipError_t launch_order_variant(int sum, int mul, const OrderParams &p, int grid, size_t lds, hipStream_t s);

#define ULTRA_ORDER_CASE(S, M) \
    case (S) * 4 + (M):        \
        return launch_order_one<T, S, M, REL_LDS>(p, grid, lds, s);

#define ULTRA_DEFINE_ORDER_VARIANT(T_, REL_LDS_)                                                                            \
    template <>                                                                                                             \
    hipError_t launch_order_variant<T_, REL_LDS_>(int sum, int mul, const OrderParams &p, int grid, size_t lds, hipStream_t s) { \
        using T = T_;                                                                                                       \
        constexpr bool REL_LDS = REL_LDS_;                                                                                  \
        switch (sum * 4 + mul) {                                                                                            \
            ULTRA_ORDER_CASE(0, 0) ULTRA_ORDER_CASE(0, 1) ULTRA_ORDER_CASE(0, 2) ULTRA_ORDER_CASE(0, 3)                     \
            ULTRA_ORDER_CASE(1, 0) ULTRA_ORDER_CASE(1, 1) ULTRA_ORDER_CASE(1, 2) ULTRA_ORDER_CASE(1, 3)                     \
            ULTRA_ORDER_CASE(2, 0) ULTRA_ORDER_CASE(2, 1) ULTRA_ORDER_CASE(2, 2) ULTRA_ORDER_CASE(2, 3)                     \
        }                                                                                                                   \
        return hipErrorInvalidValue;                                                                                        \
    }

}  // namespace ultra
