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
    int32_t keep_mode;        // weights are a 0/1 keep mask (see weigh(), rspmm_kernels.hpp)
    int32_t smod, nparts;
    uint32_t x_row_bytes, rel_row_bytes;
    long long *trace;         // measurement hook (NULL in production): per workgroup {start, chains done, end} shader clocks
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

// Read-only schedule data addressed by a wave-uniform index: loaded through the constant address space so that it
// comes in over the scalar cache (s_load) -- off the vector-memory counter, whose in-order bookkeeping would otherwise
// make a descriptor prefetch drain the gathers queued behind it.
template <typename U>
__device__ __forceinline__ U load_uniform(const U *ptr) {
    return *(const __attribute__((address_space(4))) U *)(ptr);
}
typedef int v4i __attribute__((ext_vector_type(4)));

// A load that sits behind a branch which re-joins the main path defeats the compiler's vmcnt bookkeeping (at the join
// it must assume the shorter queue and drains everything).  The walks below therefore never skip a load on a path that
// re-joins: every "is there more?" test either issues the next chunk's loads and carries on, or finishes the walk on
// its own exit path.  The record streams are padded (ORDER_PAD entries past the last edge are readable).

// ---- group items: four rows per wave, each walked sequentially by one 16-lane group ----
// Records come 8 steps at a time (lane l16 holds the record of step base + (l16 & 7)), requested a whole round before
// use and broadcast inside the group with ds_swizzle; source rows are gathered one 4-step chunk ahead of the chunk
// being reduced, so 8 rows per lane are in flight.  Steps a group does not have (its row is shorter than the unit's
// longest) gather row 0 -- an L1 hit -- and are discarded by the step predicate.
template <typename T, int SUM, int MUL, bool REL_LDS, bool WEIGHTED>
__device__ __forceinline__ Pack<T, 4> walk_row_in_order(const OrderParams &p, const int begin, const int cnt, const int nsteps,
                                                        const int nfull, const int l16, const char *xbase, const char *relbase,
                                                        const uint32_t lane_bytes, const T *lds_rel_lane) {
    constexpr int SPAN = 64;
    using P = Pack<T, 4>;
    using V = typename VecOf<T, 4>::type;
    V acc = V(nary_zero<T, SUM>());
    const T *wt = reinterpret_cast<const T *>(p.w);
    const int l8 = l16 & 7;
    const int2 *recs = reinterpret_cast<const int2 *>(p.rec) + begin + l8;
    const int32_t *perm = p.perm + begin + l8;

    struct Records {   // this lane holds the record of step (round base + l8) of its group's row
        int c, t;
        T w;
    };
    // raw = as loaded (a step past the row's end holds a neighbouring row's record); promote() masks those to node 0
    const auto load_records = [&](const int k0, int &pm) {
        Records r;
        const int2 ct = recs[k0];
        r.c = ct.x, r.t = ct.y, r.w = T(1);
        if (WEIGHTED) {
            r.w = wt[pm];           // pm: original edge id, requested one round earlier (w[perm[.]] is a dependent load)
            pm = perm[k0 + 8];
        }
        return r;
    };
    const auto promote = [&](const Records &raw, const int k0) {
        Records r = raw;
        r.c = (k0 + l8 < cnt) ? raw.c : 0;
        return r;
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
            if (WEIGHTED) y = weigh<T, SUM>(y, f.w[q], p.keep_mode);   // w * x, rspmm.cpp:68
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
        int pm = 0;
        if (WEIGHTED) pm = perm[0];
        Records nxt_raw = load_records(0, pm);
        Records cur = promote(nxt_raw, 0);
        nxt_raw = load_records(8, pm);
        Fetched fa, fb;
        fetch(StepTag<0>{}, fa, cur);
        for (int kb = 0;; kb += 8) {
            if (kb + 4 < nsteps) {
                fetch(StepTag<4>{}, fb, cur);
                reduce(fa, kb);
            } else {
                compute(std::true_type{}, fa, kb);
                break;
            }
            if (kb + 8 < nsteps) {
                cur = promote(nxt_raw, kb + 8);
                nxt_raw = load_records(kb + 16, pm);       // requested a whole round before its first use
                fetch(StepTag<0>{}, fa, cur);
                reduce(fb, kb + 4);
            } else {
                compute(std::true_type{}, fb, kb + 4);
                break;
            }
        }
    }
    return to_pack<T, 4>(acc);
}

// ---- consumer side of a chain chunk ----
// The adds form one dependent chain per lane -- the serial part of the whole kernel; the LDS reads do not depend on it.
// A full chunk is taken in two halves of CHAIN_HALF messages: the reads of a half are in flight while the previous
// half is added, and the second half of chunk i is added behind barrier i + 1 (its values are in registers by then),
// so the chain only ever waits for the barrier.
constexpr int CHAIN_HALF = CHAIN_SLOTS / 2;

template <typename T, int N>
__device__ __forceinline__ void ring_read(T (&v)[N], const T *ring_lane) {
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = ring_lane[k * 64];
}
template <typename T, int SUM, int N>
__device__ __forceinline__ T chain_add(T acc, const T (&v)[N]) {
#pragma unroll
    for (int k = 0; k < N; ++k) acc = nary<T, SUM>(acc, v[k]);
    return acc;
}
// acc (+)= v[0..N) while the next half is requested: the reads issue in the shadow of the dependent adds
template <typename T, int SUM, int N>
__device__ __forceinline__ T chain_add_and_read(T acc, const T (&v)[N], T (&next)[N], const T *ring_lane) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
        next[k] = ring_lane[k * 64];
        acc = nary<T, SUM>(acc, v[k]);
    }
    return acc;
}
template <typename T, int SUM>
__device__ __forceinline__ T consume_partial(T acc, const T *ring_lane, const int count) {
    int k = 0;
    for (; k + 4 <= count; k += 4) {
        const T v0 = ring_lane[(k + 0) * 64], v1 = ring_lane[(k + 1) * 64], v2 = ring_lane[(k + 2) * 64],
                v3 = ring_lane[(k + 3) * 64];
        acc = nary<T, SUM>(acc, v0);
        acc = nary<T, SUM>(acc, v1);
        acc = nary<T, SUM>(acc, v2);
        acc = nary<T, SUM>(acc, v3);
    }
    for (; k < count; ++k) acc = nary<T, SUM>(acc, ring_lane[k * 64]);
    return acc;
}

template <typename T, int SUM, int MUL, bool REL_LDS, bool WEIGHTED>
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

    if (p.trace && tid == 0) p.trace[3 * blockIdx.x + 0] = clock64();
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
        // Chunk i of the workgroup's chunk list is produced into ring half (i & 1) before barrier i and consumed after
        // it; the consumer reaches barrier i + 1 only after it has drained chunk i, so a ring half is never overwritten
        // early.  Producers and the consumer run separate loops with the same number of barriers.
        const int c0 = p.has_chain ? p.chunk_ptr[part] : 0, c1 = p.has_chain ? p.chunk_ptr[part + 1] : 0;
        if (c1 > c0) {
            if (wave == 0) {
                T cacc = nary_zero<T, SUM>();
                const auto finish_row = [&](const int row) {
                    const int d = inner * SPAN + lane;
                    if (d < p.row_len) {
                        T v = cacc;
                        if (p.has_bnd && (bnd_row < 0 || bnd_row == row))
                            v = nary<T, SUM>(v, reinterpret_cast<const T *>(p.bnd.ptr)[outer * p.bnd.stride_outer +
                                                                                      (long long)row * p.bnd.stride_row + d]);
                        reinterpret_cast<T *>(p.out)[outer * p.out_stride_outer + (long long)row * p.out_stride_row + d] = v;
                    }
                };
                // chunk descriptors are requested four chunks before their barrier (a load issued right before the
                // barrier and used right behind it would put an L2 round trip into every link of the chain)
                v4i dq[4];
                const v4i *chunks = reinterpret_cast<const v4i *>(p.chunks);
#pragma unroll
                for (int j = 0; j < 4; ++j) dq[j] = load_uniform(chunks + min(c0 + j, c1 - 1));
                T va[CHAIN_HALF], vb[CHAIN_HALF];
                bool pend = false;            // vb holds the second half of the previous chunk, not yet added
                int pend_row = -1;            // ... and that chunk closes this row (-1: it does not)
                const auto consume = [&](auto jtag, const int it) {
                    constexpr int J = decltype(jtag)::value;
                    const v4i ch = dq[J];
                    const int row = ch[0], count = ch[2], flags = ch[3];
                    const T *ring_lane = ring + (size_t)((it - c0) & 1) * CHAIN_SLOTS * SPAN + lane;
                    __syncthreads();
                    // (scalar loads count on lgkmcnt, which the barrier drains: requested right behind it, consumed
                    // four barriers later)
                    dq[J] = load_uniform(chunks + min(it + 4, c1 - 1));
                    if (count == CHAIN_SLOTS) {
                        if (pend) {
                            cacc = chain_add_and_read<T, SUM>(cacc, vb, va, ring_lane);
                            if (pend_row >= 0) finish_row(pend_row);
                        } else {
                            ring_read(va, ring_lane);
                        }
                        if (flags & CHUNK_FIRST) cacc = nary_zero<T, SUM>();
                        cacc = chain_add_and_read<T, SUM>(cacc, va, vb, ring_lane + CHAIN_HALF * SPAN);
                        pend = true;
                        pend_row = (flags & CHUNK_LAST) ? row : -1;
                    } else {
                        if (pend) {
                            cacc = chain_add<T, SUM>(cacc, vb);
                            if (pend_row >= 0) finish_row(pend_row);
                            pend = false;
                        }
                        if (flags & CHUNK_FIRST) cacc = nary_zero<T, SUM>();
                        cacc = consume_partial<T, SUM>(cacc, ring_lane, count);
                        if (flags & CHUNK_LAST) finish_row(row);
                    }
                };
                for (int it = c0;; it += 4) {
                    consume(StepTag<0>{}, it);
                    if (it + 1 >= c1) break;
                    consume(StepTag<1>{}, it + 1);
                    if (it + 2 >= c1) break;
                    consume(StepTag<2>{}, it + 2);
                    if (it + 3 >= c1) break;
                    consume(StepTag<3>{}, it + 3);
                    if (it + 4 >= c1) break;
                }
                if (pend) {
                    cacc = chain_add<T, SUM>(cacc, vb);
                    if (pend_row >= 0) finish_row(pend_row);
                }
            } else {
                // Producer group `slot` computes message `slot` of every chunk.  Software pipeline per lane: source rows
                // of the next 4 chunks in flight (xq), records of the 4 chunks after those requested (rq) -- all loads
                // unconditional (a slot past its chunk's count reads a neighbouring record; nobody consumes its message).
                const int slot = (wave - 1) * 4 + grp;
                const int2 *recs = reinterpret_cast<const int2 *>(p.rec) + slot;
                const int32_t *perm = p.perm + slot;
                const auto chunk_begin = [&](const int ci) {
                    return load_uniform(reinterpret_cast<const int32_t *>(p.chunks + min(ci, c1 - 1)) + 1);
                };
                struct Rec {
                    int c, t;
                    T w;
                };
                const auto load_rec = [&](const int b) {   // b: first edge of the chunk
                    const int2 ct = recs[b];
                    Rec r;
                    r.c = ct.x, r.t = ct.y, r.w = T(1);
                    if (WEIGHTED) r.w = wt[perm[b]];
                    return r;
                };
                const auto gather = [&](const Rec &r) {
                    P v;
                    if (MUL != BIN_LHS)
                        v = *reinterpret_cast<const P *>(xbase + (__umul24((uint32_t)r.c, p.x_row_bytes) + lane_bytes));
                    return v;
                };
                // depth of the pipeline in chunks (8 measured slower than 4: the compiler's in-order vmcnt bookkeeping
                // collapses at the loop back-edge and drains the deeper queue once per round)
                constexpr int D = 4;
                Rec rx[D], rq[D];   // rx[j]: record whose source row is in xq[j]; rq[j]: record of the chunk D further on
                P xq[D];
                int sb[D];          // first edge of the chunk 2 D further on (scalar loads, requested D chunks before use)
#pragma unroll
                for (int j = 0; j < D; ++j) rx[j] = load_rec(chunk_begin(c0 + j));
#pragma unroll
                for (int j = 0; j < D; ++j) rq[j] = load_rec(chunk_begin(c0 + D + j));
#pragma unroll
                for (int j = 0; j < D; ++j) sb[j] = chunk_begin(c0 + 2 * D + j);
#pragma unroll
                for (int j = 0; j < D; ++j) xq[j] = gather(rx[j]);
                const auto produce = [&](auto jtag, const int it) {
                    constexpr int J = decltype(jtag)::value;
                    const int b_next = sb[J];
                    sb[J] = chunk_begin(it + 3 * D);   // scalar load: issued right behind the previous barrier (see the consumer)
                    P rv;
                    if (MUL != BIN_RHS) {
                        if (REL_LDS)
                            rv = *reinterpret_cast<const P *>(lds_rel_lane + rx[J].t * SPAN);
                        else
                            rv = *reinterpret_cast<const P *>(relbase + (__umul24((uint32_t)rx[J].t, p.rel_row_bytes) + lane_bytes));
                    }
                    const V rr = (MUL != BIN_RHS) ? to_vec<T, 4>(rv) : V(T(0));
                    const V xx = (MUL != BIN_LHS) ? to_vec<T, 4>(xq[J]) : V(T(0));
                    V y = binary_vec<V, MUL>(rr, xx);
                    if (WEIGHTED) y = weigh<T, SUM>(y, rx[J].w, p.keep_mode);
                    *reinterpret_cast<P *>(ring + ((size_t)((it - c0) & 1) * CHAIN_SLOTS + slot) * SPAN + l16 * 4) = to_pack<T, 4>(y);
                    // refill this pipeline stage: source row of chunk it + D, record of chunk it + 2 D
                    rx[J] = rq[J];
                    xq[J] = gather(rx[J]);
                    rq[J] = load_rec(b_next);
                    __syncthreads();
                };
                for (int it = c0;; it += D) {
                    produce(StepTag<0>{}, it);
                    if (it + 1 >= c1) break;
                    produce(StepTag<1>{}, it + 1);
                    if (it + 2 >= c1) break;
                    produce(StepTag<2>{}, it + 2);
                    if (it + 3 >= c1) break;
                    produce(StepTag<3>{}, it + 3);
                    if (it + 4 >= c1) break;
                }
            }
        }

        if (p.trace && tid == 0) p.trace[3 * blockIdx.x + 1] = clock64();
        // ================= group units of this workgroup =================
        // (the next unit's item is requested before the current one is walked: two dependent loads off the critical path;
        // the unit list is read with a clamped index so that the request is unconditional)
        const int u1 = p.unit_ptr[part + 1];
        const auto load_item = [&](const int ui) {
            const int u = load_uniform(p.units + min(ui, u1 - 1));
            return p.items[min(p.n_chain + 4 * u + grp, p.n_item - 1)];
        };
        const auto item_valid = [&](const int ui) {   // (re-reads the unit id: L1 / L2 resident by then)
            return p.n_chain + 4 * load_uniform(p.units + min(ui, u1 - 1)) + grp < p.n_item;
        };
        int ui = p.unit_ptr[part] + wave;
        int4 item_next = make_int4(-1, 0, 0, 0);
        bool valid_next = false;
        if (ui < u1) {
            item_next = load_item(ui);
            valid_next = item_valid(ui);
        }
        for (; ui < u1; ui += nwave) {
            const int4 item = item_next;
            const bool valid = valid_next;
            item_next = load_item(ui + nwave);
            valid_next = item_valid(ui + nwave);
            const int row = valid ? item.x : -1, begin = valid ? item.y : 0, cnt = valid ? item.z : 0;
            const int m01 = max(__shfl(cnt, 0), __shfl(cnt, 16));
            const int m23 = max(__shfl(cnt, 32), __shfl(cnt, 48));
            const int nsteps = rfl(max(m01, m23));
            const int n01 = min(__shfl(cnt, 0), __shfl(cnt, 16));
            const int n23 = min(__shfl(cnt, 32), __shfl(cnt, 48));
            const int nfull = rfl(min(n01, n23));

            P acc = walk_row_in_order<T, SUM, MUL, REL_LDS, WEIGHTED>(p, begin, cnt, nsteps, nfull, l16, xbase, relbase, lane_bytes,
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
    if (p.trace) {
        __syncthreads();
        if (tid == 0) p.trace[3 * blockIdx.x + 2] = clock64();
    }
}

// ---- per-variant launchers (explicitly instantiated in rspmm_order_*.hip) ----
template <typename T, int SUM, int MUL, bool REL_LDS, bool WEIGHTED>
inline hipError_t launch_order_one(const OrderParams &p, int grid, size_t lds, hipStream_t s) {
    auto kern = rspmm_order_kernel<T, SUM, MUL, REL_LDS, WEIGHTED>;
    static size_t lds_opted_in = 0;   // (see launch_one in rspmm_kernels.hpp)
    if (lds > 48 * 1024 && lds > lds_opted_in) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_opted_in = lds;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(ORDER_THREADS), lds, s, p);
    return hipGetLastError();
}

template <typename T, bool REL_LDS, bool WEIGHTED>
hipError_t launch_order_variant(int sum, int mul, const OrderParams &p, int grid, size_t lds, hipStream_t s);

#define ULTRA_ORDER_CASE(S, M) \
    case (S) * 4 + (M):        \
        return launch_order_one<T, S, M, REL_LDS, WEIGHTED>(p, grid, lds, s);

#define ULTRA_DEFINE_ORDER_VARIANT(T_, REL_LDS_, WEIGHTED_)                                                                 \
    template <>                                                                                                             \
    hipError_t launch_order_variant<T_, REL_LDS_, WEIGHTED_>(int sum, int mul, const OrderParams &p, int grid, size_t lds,   \
                                                             hipStream_t s) {                                               \
        using T = T_;                                                                                                       \
        constexpr bool REL_LDS = REL_LDS_, WEIGHTED = WEIGHTED_;                                                            \
        switch (sum * 4 + mul) {                                                                                            \
            ULTRA_ORDER_CASE(0, 0) ULTRA_ORDER_CASE(0, 1) ULTRA_ORDER_CASE(0, 2) ULTRA_ORDER_CASE(0, 3)                     \
            ULTRA_ORDER_CASE(1, 0) ULTRA_ORDER_CASE(1, 1) ULTRA_ORDER_CASE(1, 2) ULTRA_ORDER_CASE(1, 3)                     \
            ULTRA_ORDER_CASE(2, 0) ULTRA_ORDER_CASE(2, 1) ULTRA_ORDER_CASE(2, 2) ULTRA_ORDER_CASE(2, 3)                     \
        }                                                                                                                   \
        return hipErrorInvalidValue;                                                                                        \
    }

}  // namespace ultra
