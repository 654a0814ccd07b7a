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
//
// (Since round 3 the group items of the fp32 / unit-weight kernels are walked as STREAMS -- the rows a 16-lane group was dealt,
// back to back, by the generated assembly loops of rspmm_order_asm.hpp, records broadcast with DPP moves -- and the UPDATE
// instances apply the layer update to the rows a workgroup aggregated: 1 in the kernel's tail, 2 / 3 beside the walk, by waves
// 12..15 (3: rows handed over through an LDS ring, x rows included).  In the schedules of forms
// 2 / 3 chain rows of up to 2.1 x the mean stream length are stream rows too.)
#pragma once

#include "rspmm_kernels.hpp"
#ifdef ULTRA_ORDER_ASM_HEADER   // measurement builds: a header generated with other switches (tools/build_variant.py)
#include ULTRA_ORDER_ASM_HEADER
#else
#include "rspmm_order_asm.hpp"
#endif
#include "update_tile.hpp"

static_assert(ULTRA_STREAM_PRESHIFT_GEN == ULTRA_STREAM_PRESHIFT,
              "the generated walk and plan.cpp must agree on the record format of the twelve-walker schedules (plan.hpp)");

#pragma clang fp contract(off)

namespace ultra {

constexpr int ORDER_THREADS = 64 * ORDER_WAVES;
// bounded spins of the hand-off protocols (OrderParams::err): polls before a wait gives up, and who gave up
constexpr uint32_t SPIN_CAP = 1u << 20;      // (a poll is an LDS round trip, 100+ cycles, mostly with an s_sleep: >= 0.05 s)
enum { ORDER_SPIN_LOOK = 1,        // an update wave waiting for a block of rows from the walkers
       ORDER_SPIN_MEET = 2,        // an update wave waiting for the other update waves at a meeting point
       ORDER_SPIN_CHAIN_WALKER = 3, // a walker waiting for the chain consumer to leave the ring
       ORDER_SPIN_CHAIN_UPDATER = 4, // an update wave waiting for the same
       ORDER_SPIN_PARK = 5 };      // a walker waiting for a free row of the hand-off ring (stream_park in the generator)
// UPDATE == 3 (the layer update runs BESIDE the walk): waves [0, ORDER_WALKERS) walk the streams, the last ORDER_UPDATERS waves --
// one per SIMD -- multiply the rows the walkers hand over through an LDS ring (update_tile.hpp, UPD2_*; the generator's HANDOFF2_*
// constants are the same numbers).  (Round 4 also kept form 2 -- rows handed over BY REFERENCE, through memory and an LDS queue
// of row offsets: bit-exact, never faster than the tail form, superseded by form 3; removed in round 5)
constexpr int ORDER_UPDATERS = 4, ORDER_WALKERS = ORDER_THREADS / 64 - ORDER_UPDATERS;
#ifndef ULTRA_CHAIN_PRIO
#define ULTRA_CHAIN_PRIO 1
#endif
static_assert(CHAIN_SLOTS == 4 * (ORDER_THREADS / 64 - 1), "one ring slot per producer group");

struct OrderParams {
    const int32_t *rec;       // (col, type) per sorted edge
    const int32_t *perm;      // sorted position -> original edge id
    const void *w;            // edge weights in ORIGINAL edge order, or NULL (all ones)
    const int4 *items;        // {row, begin, len, -}; chain rows first, group items from n_chain on
    const int32_t *unit_ptr, *units, *chunk_ptr;
    const int4 *chunks;       // {row, begin, count, flags}
    const int32_t *srec;      // group streams (plan.hpp Schedule): records and {first record, steps} per (workgroup, 16-lane group)
    const int2 *sdesc;
    int32_t use_streams;
    int32_t n_chain, n_item;
    MatArg rel, x, bnd;
    const long long *bnd_rows;   // point boundary: bnd holds ONE row per outer slice, added at row bnd_rows[outer] only
    // min / max with a point boundary: every OTHER row meets this value at its flush (layers.py:206-207 takes
    // max(update, boundary) against a tensor that is zero off the query rows); off: rows other than the boundary row are left alone
    int32_t bnd_fill_on;
    float bnd_fill;
    void *out;
    long long out_stride_outer, out_stride_row;
    int32_t n_outer, row_len, spans_per_outer, n_span;
    int32_t num_rel, has_bnd, has_chain;
    int32_t keep_mode;        // weights are a 0/1 keep mask (see weigh(), rspmm_kernels.hpp)
    int32_t smod, nparts;
    uint32_t x_row_bytes, rel_row_bytes;
    long long *trace;         // measurement hook (NULL in production): per workgroup {start, chains done, end} shader clocks
    // Every wait of the hand-off protocols (forms 2 / 3: LDS words polled by one wave until another wave writes them) is BOUNDED:
    // a wave that has polled SPIN_CAP times (+ an allowance that grows with the work the other side may legitimately still have)
    // stores (code << 24 | workgroup) here -- a word of pinned host memory, rspmm_api.hip device_error_word() -- lets the other
    // side through and carries on / ends, so a protocol error (or a broken schedule) is a failed call with a message
    // (ultra_device_error, checked at the next entry), not a hung GPU.  Codes: ORDER_SPIN_*.
    uint32_t *err;
    int32_t max_stream_steps; // longest group stream of the schedule in steps (allowance of the update waves' wait for rows)
    // UPDATE instances: the layer update (update_tile.hpp) of the rows this workgroup aggregated, applied after its walk
    struct Update {
        const float *weight, *bias, *ln_w, *ln_b;   // Linear(128 -> 64) [+ LayerNorm(64)]
        float *out;                                  // (n_outer, num_node, 64) with the strides below, in floats
        long long out_stride_outer, out_stride_row;
        const int32_t *prow, *prow_ptr;              // plan.hpp Schedule
        float eps;
        int32_t flags;                               // CONV_LN | CONV_RELU | CONV_RESIDUAL
        int32_t mode;                                // 1: in the kernel's tail, 2 / 3: beside the walk (rspmm_order_kernel, UPDATE)
        uint32_t ctl_off;                            // modes 2 and 3: byte offset of the hand-off / control block in LDS
    } upd;
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
    // record / permutation streams: uniform base + 32-bit per-lane offset (no 64-bit address pair to keep alive)
    const char *rec_base = reinterpret_cast<const char *>(p.rec);
    const char *perm_base = reinterpret_cast<const char *>(p.perm);
    const uint32_t first = (uint32_t)(begin + l8);
    // The loop below runs nsteps rounds for ALL four groups of the wave (nsteps = the longest of their rows).  A group whose
    // own row is short keeps requesting records past its end -- masked by promote() / the PRED computes, but REQUESTED: beside
    // a 4,000-step sibling a row at the end of the edge list asked for records (and, weighted, for w[perm[.]] of whatever
    // lay there) kilobytes past the ORDER_PAD entries that are readable.  The requests stop advancing a little past the
    // row's own end.
    const uint32_t last = (uint32_t)(begin + cnt) + 16u;

    struct Records {   // this lane holds the record of step (round base + l8) of its group's row
        int c, t;
        T w;
    };
    // raw = as loaded (a step past the row's end holds a neighbouring row's record); promote() masks those to node 0
    const auto load_records = [&](const int k0, int &pm) {
        Records r;
        const int2 ct = *reinterpret_cast<const int2 *>(rec_base + min(first + (uint32_t)k0, last) * 8u);
        r.c = ct.x, r.t = ct.y, r.w = T(1);
        if (WEIGHTED) {
            r.w = wt[pm];           // pm: original edge id, requested one round earlier (w[perm[.]] is a dependent load)
            pm = *reinterpret_cast<const int32_t *>(perm_base + min(first + (uint32_t)k0 + 8u, last) * 4u);
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
        if (WEIGHTED) pm = *reinterpret_cast<const int32_t *>(perm_base + first * 4u);
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

// The configurations whose two pipelined loops run as hand-scheduled assembly (rspmm_order_asm.hpp): fp32, unit
// weights, relation slice in LDS, mul / add messages -- the inference path.  Everything else takes the C++ loops below.
#ifndef ULTRA_ORDER_ASM
#define ULTRA_ORDER_ASM 1
#endif
#ifndef ULTRA_DBG_CHAIN
#define ULTRA_DBG_CHAIN 0
#endif
#ifndef ULTRA_UPD_PRIO
#define ULTRA_UPD_PRIO 3     /* s_setprio of the form-3 update waves */
#endif
#ifndef ULTRA_UPD_SKIP
#define ULTRA_UPD_SKIP 0     /* measurement builds (wrong results): the form-3 update waves skip 1 = operand reads and matrix chains, 2 = the matrix chains (a VALU add per operand instead), 3 = the operand reads */
#endif
#ifndef ULTRA_SPIN_GUARD
#define ULTRA_SPIN_GUARD 0   /* debugging: the form-3 update waves' spins give up after 2^20 turns and report (trace[24 grid + ..]) */
#endif
#ifndef ULTRA_ASM_PRODUCE
#define ULTRA_ASM_PRODUCE 1
#endif
template <typename T, int MUL, bool REL_LDS, bool WEIGHTED>
struct OrderAsm {
    static constexpr bool value =
        ULTRA_ORDER_ASM && std::is_same<T, float>::value && REL_LDS && !WEIGHTED && (MUL == BIN_MUL || MUL == BIN_ADD);
};
__device__ __forceinline__ uint32_t lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char *)p;
}

// ---- chain rows: ring layout and the consumer's reads ----
// A chunk's CHAIN_SLOTS messages sit in the ring as CHAIN_QUADS quads; quad q holds, for each of the span's 64 elements,
// the FOUR messages 4 q .. 4 q + 3 next to each other (16 B for fp32): [half][quad][lane][4].  The consumer lane reads
// four links of its chain with one ds_read_b128 -- one LDS instruction per four dependent adds instead of one per
// add (a wave issues roughly one instruction per four cycles, so with a read per add the reads, not the adds, paced
// the chain).  The producers hold a message as four consecutive elements per lane (16-lane group g of wave 1 + q holds
// message 4 q + g); a 4 x 4 transpose across the wave's four lane rows (two v_permlane32_swap + two v_permlane16_swap
// per 32-bit word) turns that into "lane (g, l16) holds element 4 l16 + g of messages 4 q .. 4 q + 3", which is
// stored with one conflict-free 16-byte write at [quad][lane].  Consumer lane c therefore owns element
// chain_element(c) = 4 (c % 16) + c / 16 of the span.
constexpr int CHAIN_QUADS = CHAIN_SLOTS / 4;
constexpr int CHAIN_QA = (CHAIN_QUADS + 1) / 2;    // quads read right behind the barrier ...
constexpr int CHAIN_QB = CHAIN_QUADS - CHAIN_QA;   // ... and quads read in the shadow of the first half's adds
static_assert(CHAIN_SLOTS % 4 == 0, "whole quads");

__device__ __forceinline__ int chain_element(const int lane) { return 4 * (lane & 15) + (lane >> 4); }

// rows of 16 lanes: (a0 a1 a2 a3), (b0 b1 b2 b3) -> (a0 a1 b0 b1), (a2 a3 b2 b3)
__device__ __forceinline__ void swap_rows32(uint32_t &a, uint32_t &b) {
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0], b = r[1];
}
// (a0 a1 a2 a3), (b0 b1 b2 b3) -> (a0 b0 a2 b2), (a1 b1 a3 b3)
__device__ __forceinline__ void swap_rows16(uint32_t &a, uint32_t &b) {
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0], b = r[1];
}
// y[e] of lane row g  <-  y[g] of lane row e   (g, e = 0..3; same lane % 16)
template <typename T>
__device__ __forceinline__ void transpose_lane_rows(typename VecOf<T, 4>::type &y) {
    constexpr int W = sizeof(T) / 4;   // 32-bit words per element
    uint32_t w[4][W];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const T v = y[e];
        __builtin_memcpy(w[e], &v, sizeof(T));
    }
#pragma unroll
    for (int k = 0; k < W; ++k) {
        swap_rows32(w[0][k], w[2][k]);
        swap_rows32(w[1][k], w[3][k]);
        swap_rows16(w[0][k], w[1][k]);
        swap_rows16(w[2][k], w[3][k]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        T v;
        __builtin_memcpy(&v, w[e], sizeof(T));
        y[e] = v;
    }
}

template <typename V, int N>
__device__ __forceinline__ void ring_read(V (&v)[N], const V *ring_lane) {
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = ring_lane[q * 64];
}
// the dependent chain: 4 N links
template <typename T, int SUM, typename V, int N>
__device__ __forceinline__ T chain_add(T acc, const V (&v)[N]) {
#pragma unroll
    for (int q = 0; q < N; ++q) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = nary<T, SUM>(acc, v[q][e]);
    }
    return acc;
}
// the first `count` links of quads [Q0, Q0 + N) of a chunk (count is wave-uniform)
template <typename T, int SUM, typename V, int N>
__device__ __forceinline__ T chain_add_partial(T acc, const V (&v)[N], const int q0, const int count) {
#pragma unroll
    for (int q = 0; q < N; ++q) {
        const int left = count - 4 * (q0 + q);
        if (left >= 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = nary<T, SUM>(acc, v[q][e]);
        } else if (left > 0) {
            acc = nary<T, SUM>(acc, v[q][0]);
            if (left > 1) acc = nary<T, SUM>(acc, v[q][1]);
            if (left > 2) acc = nary<T, SUM>(acc, v[q][2]);
        }
    }
    return acc;
}

// STREAMS: the group rows are walked as streams by the assembly loop (its own instantiation: the C++ unit walk
// and the assembly walk in one kernel cost each other registers around the asm statements).
// UPDATE: 0 none; 1 the layer update of the workgroup's rows in the kernel's TAIL (after every walk has ended); 3 the update
// BESIDE the walk with the aggregate passing THROUGH LDS -- the last ORDER_UPDATERS waves do not walk: the walkers park every
// finished row (and its x row) in 16-row tiles, the update waves keep their slice of the weight matrix in registers and split
// every tile by features (update_tile.hpp, UPD2_*); the aggregate never goes to memory, x and the output move as whole 256-byte
// rows.  (2 was round 4's by-reference form: removed.)
// (a workgroup of fewer than sixteen waves still keeps to 128 registers per lane: the wave slots it leaves are meant for
// workgroups of other kernels, which need their share of the register file)
#if ULTRA_ORDER_WAVES < 16
#define ULTRA_ORDER_VGPR_CAP __attribute__((amdgpu_waves_per_eu(4, 4)))
#else
#define ULTRA_ORDER_VGPR_CAP
#endif
#ifndef ULTRA_UPD_SIMDS
#define ULTRA_UPD_SIMDS 4
#endif
__device__ __forceinline__ int order_role(const int pw) {   // physical wave -> logical wave (a bijection that keeps wave 0)
#if ULTRA_UPD_SIMDS == 2 && ULTRA_ORDER_WAVES == 16
    return pw < 10 ? pw : (pw == 12 ? 10 : pw == 13 ? 11 : pw == 10 ? 12 : pw == 11 ? 13 : pw);
#elif ULTRA_UPD_SIMDS == 1 && ULTRA_ORDER_WAVES == 16
    return (pw & 3) == 3 ? 12 + (pw >> 2) : 3 * (pw >> 2) + (pw & 3);
#else
    return pw;
#endif
}
template <typename T, int SUM, int MUL, bool REL_LDS, bool WEIGHTED, bool STREAMS, int UPDATE = 0>
__global__ void __launch_bounds__(ORDER_THREADS) ULTRA_ORDER_VGPR_CAP rspmm_order_kernel(const OrderParams p) {
    static_assert(!STREAMS || OrderAsm<T, MUL, REL_LDS, WEIGHTED>::value, "group streams exist for the assembly configurations only");
    static_assert(!UPDATE || (STREAMS && sizeof(T) == 4), "the layer update follows the fp32 stream walk");
    static_assert(UPDATE != 3 || ORDER_UPDATERS == 4, "form 3 splits the 64 output features over four update waves");
    constexpr int SPAN = 64;
    using P = Pack<T, 4>;
    using V = typename VecOf<T, 4>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *lds_rel = reinterpret_cast<T *>(smem);
    T *ring = lds_rel + ((REL_LDS && MUL != BIN_RHS) ? (size_t)(p.num_rel + 1) * SPAN : 0);   // [2][CHAIN_QUADS][64][4]; (+ 1: marker row)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // Roles by wave.  The hardware deals a workgroup's waves round-robin onto the CU's four SIMDs (wave w on SIMD w % 4), and the
    // code below speaks of LOGICAL waves: 0 the chain consumer, 0 .. ORDER_WALKERS - 1 the walkers, the last ORDER_UPDATERS the
    // update waves of form 3.  ULTRA_UPD_SIMDS (measurement builds): 4 = one update wave per SIMD (logical = physical), 2 = the
    // four update waves on SIMDs 2 and 3 (physical 10, 11, 14, 15), 1 = all on SIMD 3 (physical 3, 7, 11, 15) -- a SIMD that
    // executes matrix instructions issues little else, so where the update waves sit decides which walkers they hold up.
    const int wave = order_role(rfl(tid >> 6));
    constexpr int nwave = ORDER_THREADS / 64;
    const int part = blockIdx.x / p.smod;
    if (part >= p.nparts) return;
    const T *wt = reinterpret_cast<const T *>(p.w);
    // UPDATE == 3: the control block of the hand-off (update_tile.hpp UPD2_CTL_*)
    volatile uint32_t *ctl = reinterpret_cast<volatile uint32_t *>(smem + (UPDATE == 3 ? p.upd.ctl_off : 0));
    const T fill = (SUM != 0 && p.bnd_fill_on) ? (T)p.bnd_fill : nary_zero<T, SUM>();   // (what a non-boundary row meets under min / max)

    // a bounded spin gave up (OrderParams::err): one store to the host's error word, system scope
    const auto report_spin = [&](const int code) {
        if (p.err) __hip_atomic_store(p.err, ((uint32_t)code << 24) | (blockIdx.x & 0xffffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    (void)report_spin;
    if (p.trace && tid == 0) p.trace[3 * blockIdx.x + 0] = clock64();
    for (int span = blockIdx.x % p.smod; span < p.n_span; span += p.smod) {
        const int outer = span / p.spans_per_outer;
        const int inner = span - outer * p.spans_per_outer;
        const char *xbase = reinterpret_cast<const char *>(reinterpret_cast<const T *>(p.x.ptr) + outer * p.x.stride_outer);
        const char *relbase =
            reinterpret_cast<const char *>(reinterpret_cast<const T *>(p.rel.ptr) + outer * p.rel.stride_outer);
        const long long bnd_row = p.bnd_rows ? p.bnd_rows[outer] : -1;   // (stride_row of a point boundary is 0)
        // Per-lane geometry of the span, derived from the lane id.  Each phase derives its own copy from an opaque lane id:
        // values computed once at the top would stay live through the chain phase (whose consumer holds 60 registers of
        // ring data) and come back from scratch in every unit of the walk.
        struct LaneGeom {
            int grp, l16, d0;
            bool dvalid;
            uint32_t lane_bytes;
            const T *lds_rel_lane;
        };
        const auto lane_geom = [&]() {
            int l = lane;
            asm volatile("" : "+v"(l));
            LaneGeom g;
            g.grp = l >> 4, g.l16 = l & 15;
            g.d0 = inner * SPAN + g.l16 * 4;
            g.dvalid = g.d0 < p.row_len;
            g.lane_bytes = (uint32_t)(g.dvalid ? g.d0 : 0) * (uint32_t)sizeof(T);
            g.lds_rel_lane = lds_rel + g.l16 * 4;
            return g;
        };

        if (REL_LDS && MUL != BIN_RHS) {
            __syncthreads();  // readers of the previous span are done with the LDS image
            // (the staging addresses are recomputed per span: hoisted out of the span loop they would stay live through
            // the walks below and spill)
            if constexpr (UPDATE == 3) {
                if (tid < UPD2_CTL_CTILE_BYTES / 4) ctl[tid] = 0;
            }
            int tid_stage = tid;
            asm volatile("" : "+v"(tid_stage));
            stage_slice<T, 4>(lds_rel, reinterpret_cast<const T *>(p.rel.ptr) + outer * p.rel.stride_outer, p.rel.stride_row,
                              p.num_rel, inner, p.row_len, tid_stage, ORDER_THREADS);
            __syncthreads();
            if (p.trace && tid == 0) p.trace[7 * gridDim.x + blockIdx.x] = clock64();   // (measurement hook: the relation slice is staged)
        }

        // ================= chain rows of this workgroup =================
        // Chunk i of the workgroup's chunk list is produced into ring half (i & 1) before barrier i and consumed after
        // it; the consumer reaches barrier i + 1 only after its reads of chunk i have returned (the barrier's
        // lgkmcnt(0)), so a ring half is never overwritten early.  Producers and the consumer run separate loops with
        // the same number of barriers: one per chunk.
        const int c0 = p.has_chain ? p.chunk_ptr[part] : 0, c1 = p.has_chain ? p.chunk_ptr[part + 1] : 0;
        constexpr int RING_HALF = CHAIN_QUADS * 64;   // in quads-of-lane units (V)
        if (c1 > c0) {
            const LaneGeom cg = lane_geom();
            const int grp = cg.grp, l16 = cg.l16;
            const uint32_t lane_bytes = cg.lane_bytes;
            const T *lds_rel_lane = cg.lds_rel_lane;
            (void)l16;
            if (wave == 0) {
#if ULTRA_CHAIN_PRIO
                __builtin_amdgcn_s_setprio(3);   // the chain is the launch's critical path: win VALU / LDS issue arbitration
#endif
                const int d = inner * SPAN + chain_element(lane);
                int n_listed = 0;            // UPDATE == 3: chain rows listed for the update waves (they fetch them from memory)
                const auto finish_row = [&](const int row, T v) {
                    if (d < p.row_len) {
                        if (p.has_bnd) {
                            if (bnd_row < 0 || bnd_row == row)
                                v = nary<T, SUM>(v, reinterpret_cast<const T *>(p.bnd.ptr)[outer * p.bnd.stride_outer +
                                                                                          (long long)row * p.bnd.stride_row + d]);
                            else if (SUM != 0 && p.bnd_fill_on)
                                v = nary<T, SUM>(v, fill);
                        }
                        reinterpret_cast<T *>(p.out)[outer * p.out_stride_outer + (long long)row * p.out_stride_row + d] = v;
                    }
                    if constexpr (UPDATE == 3) {
                        if (lane == 0) ctl[UPD2_CTL_CROW + min(n_listed, UPD2_MAX_CHAIN_ROWS - 1)] = (uint32_t)((long long)row * (long long)p.x_row_bytes);
                        ++n_listed;      // (the host does not pick this form for a schedule with more chain rows per workgroup)
                    }
                };
                const v4i *chunks = reinterpret_cast<const v4i *>(p.chunks);
                const V *ring_lane = reinterpret_cast<const V *>(ring) + lane;
                // The consumer walks ROWS: the first chunk's descriptor carries the row's length (flags >> 2), every
                // later chunk of the row follows from it -- no descriptor load (and no scalar-memory wait) inside a
                // row.  The next row's descriptor is requested before the row's first barrier, whose lgkmcnt(0)
                // collects it.
                v4i desc = load_uniform(chunks + c0);
                int par = 0;
#if ULTRA_DBG_CHAIN == 3   /* measurement build: cycles the consumer spends in its barriers -> trace[3 b + 2] of the LAST span */
                long long bar_cycles = 0;
#define ULTRA_CHAIN_BARRIER()                     \
    do {                                          \
        const long long t_ = clock64();           \
        __syncthreads();                          \
        bar_cycles += clock64() - t_;             \
    } while (0)
#else
#define ULTRA_CHAIN_BARRIER() __syncthreads()
#endif
                const auto chunk_wait = [&]() { ULTRA_CHAIN_BARRIER(); };
                const auto chunk_read = [&]() {};
                for (int it = c0; it < c1;) {
                    const int row = desc[0], len = desc[3] >> 2;
                    const int nfull = len / CHAIN_SLOTS, rem = len - nfull * CHAIN_SLOTS;
                    const int nch = nfull + (rem > 0 ? 1 : 0);
                    desc = load_uniform(chunks + min(it + nch, c1 - 1));
                    T cacc = nary_zero<T, SUM>();
                    V va[CHAIN_QA], vb[CHAIN_QB];
                    // (defined on every path: left undefined, the register allocator carries the two arrays as live
                    // values around the whole span loop and spills them at the assembly statements)
#pragma unroll
                    for (int q = 0; q < CHAIN_QA; ++q) va[q] = V(T(0));
#pragma unroll
                    for (int q = 0; q < CHAIN_QB; ++q) vb[q] = V(T(0));
#if ULTRA_DBG_CHAIN == 1   /* measurement build: the consumer only keeps the barrier count (results are wrong) */
                    for (int k = 0; k < nch; ++k) ULTRA_CHAIN_BARRIER();
                    if (false)
#endif
                    if (nfull > 0) {
                        // first full chunk: nothing pending yet
                        chunk_wait();
                        ring_read(va, ring_lane + par * RING_HALF);
                        ring_read(vb, ring_lane + par * RING_HALF + CHAIN_QA * 64);
                        chunk_read();
                        __builtin_amdgcn_sched_barrier(0);
                        cacc = chain_add<T, SUM>(cacc, va);
                        par ^= 1;
                        // steady state, one chunk per barrier: the first half is requested right behind the barrier and
                        // lands while the previous chunk's second half (in registers since the barrier) is added; the
                        // second half is requested before the first half's adds and collected by the next barrier.
                        for (int k = 1; k < nfull; ++k) {
                            chunk_wait();
                            ring_read(va, ring_lane + par * RING_HALF);
                            __builtin_amdgcn_sched_barrier(0);
                            cacc = chain_add<T, SUM>(cacc, vb);
                            __builtin_amdgcn_sched_barrier(0);
                            ring_read(vb, ring_lane + par * RING_HALF + CHAIN_QA * 64);
                            chunk_read();
                            __builtin_amdgcn_sched_barrier(0);
                            cacc = chain_add<T, SUM>(cacc, va);
                            par ^= 1;
                        }
                        if (rem == 0) cacc = chain_add<T, SUM>(cacc, vb);
                    }
#if ULTRA_DBG_CHAIN == 1
                    if (false)
#endif
                    if (rem > 0) {
                        // last, partial chunk (slots past `rem` hold messages nobody asked for): same shape as a
                        // steady-state step
                        chunk_wait();
                        ring_read(va, ring_lane + par * RING_HALF);
                        __builtin_amdgcn_sched_barrier(0);
                        if (nfull > 0) cacc = chain_add<T, SUM>(cacc, vb);
                        __builtin_amdgcn_sched_barrier(0);
                        ring_read(vb, ring_lane + par * RING_HALF + CHAIN_QA * 64);
                        chunk_read();
                        __builtin_amdgcn_sched_barrier(0);
                        cacc = chain_add_partial<T, SUM>(cacc, va, 0, rem);
                        cacc = chain_add_partial<T, SUM>(cacc, vb, CHAIN_QA, rem);
                        par ^= 1;
                    }
                    finish_row(row, cacc);
                    it += nch;
                }
                if constexpr (UPDATE == 3) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the chain rows are in memory: the update waves read them there
                    if (lane == 0) {
                        ctl[UPD2_CTL_NCHAIN] = (uint32_t)n_listed;
                        asm volatile("" ::: "memory");
                        ctl[3] = 1;       // ... and the ring is free: tiles may take its place
                    }
                }
#if ULTRA_DBG_CHAIN == 3
                if (p.trace && lane == 0) p.trace[3 * gridDim.x + blockIdx.x] = bar_cycles;
#endif
#if ULTRA_CHAIN_PRIO
                __builtin_amdgcn_s_setprio(0);
#endif
            } else {
                // Producer group `slot` computes message `slot` of every chunk.  Software pipeline per lane: source rows
                // of the next 4 chunks in flight (xq), records of the 4 chunks after those requested (rq) -- all loads
                // unconditional (a slot past its chunk's count reads a neighbouring record; nobody consumes its message).
                const int slot = (wave - 1) * 4 + grp;
                const auto chunk_begin = [&](const int ci) {
                    return load_uniform(reinterpret_cast<const int32_t *>(p.chunks + min(ci, c1 - 1)) + 1);
                };
#if ULTRA_DBG_CHAIN == 2   /* measurement build: the producers only keep the barrier count */
                if constexpr (true) {
                    for (int k = c0; k < c1; ++k) __syncthreads();
                } else
#endif
                if constexpr (STREAMS && ULTRA_ASM_PRODUCE) {   // (the unit-walk kernels keep the C++ producers: see STREAMS)
                    order_produce_asm<MUL>(c1 - c0, p.chunks + c0, (uint32_t)slot * 8u, lane_bytes, lds_addr(lds_rel_lane),
                                           lds_addr(ring) + (uint32_t)((wave - 1) * 64 + lane) * 16u, xbase,
                                           reinterpret_cast<const char *>(p.rec), p.x_row_bytes);
                } else {
                    const int2 *recs = reinterpret_cast<const int2 *>(p.rec) + slot;
                    const int32_t *perm = p.perm + slot;
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
                        // park it: 4 x 4 transpose across the wave's lane rows, then [quad = wave - 1][lane] (see CHAIN_QUADS)
                        transpose_lane_rows<T>(y);
                        reinterpret_cast<V *>(ring)[(((it - c0) & 1) * CHAIN_QUADS + (wave - 1)) * 64 + lane] = y;
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
        }

        if (p.trace && tid == 0) p.trace[3 * blockIdx.x + 1] = clock64();
        // ================= group units of this workgroup =================
        // (the next unit's item is requested before the current one is walked: two dependent loads off the critical path;
        // the unit list is read with a clamped index so that the request is unconditional)
        const LaneGeom ug = lane_geom();
        const int grp = ug.grp, l16 = ug.l16, d0 = ug.d0;
        const bool dvalid = ug.dvalid;
        const uint32_t lane_bytes = ug.lane_bytes;
        const T *lds_rel_lane = ug.lds_rel_lane;
        if constexpr (STREAMS) {
            if (UPDATE != 3 || wave < ORDER_WALKERS) {
                // ---- group streams: one continuous walk per 16-lane group (rspmm_order_asm.hpp) ----
                if constexpr (UPDATE == 3) {   // (the tiles the walk parks its rows in take the ring's place)
                    if (c1 > c0) {
                        // (bounded: a chunk of the chain takes ~ 700 cycles, a poll >= 200 -- 16 polls a chunk are a wide allowance)
                        const uint32_t cap = SPIN_CAP + 16u * (uint32_t)(c1 - c0);
                        uint32_t polls = 0;
                        while (ctl[3] == 0) {
                            if (++polls >= cap) {
                                if (lane == 0) report_spin(ORDER_SPIN_CHAIN_WALKER);
                                break;
                            }
                            __builtin_amdgcn_s_sleep(2);
                        }
                    }
                }
                const int2 sd = p.sdesc[(part * nwave + wave) * 4 + grp];
                const int len = sd.y;
                const int m01 = max(__shfl(len, 0), __shfl(len, 16)), m23 = max(__shfl(len, 32), __shfl(len, 48));
                const int n01 = min(__shfl(len, 0), __shfl(len, 16)), n23 = min(__shfl(len, 32), __shfl(len, 48));
                const int ns = rfl(max(m01, m23)), nf = rfl(min(n01, n23));
                float bv[4] = {0.f, 0.f, 0.f, 0.f};
                uint32_t bndoff = 0xffffffffu;
                if (p.has_bnd) {   // (a point boundary: one row per outer slice, stride_row 0)
                    const P b = *reinterpret_cast<const P *>(reinterpret_cast<const T *>(p.bnd.ptr) + outer * p.bnd.stride_outer + d0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) bv[e] = b.v[e];
                    bndoff = (uint32_t)bnd_row * p.x_row_bytes + lane_bytes;
                }
                // (min / max: a flushed row other than the boundary row meets `bz`: the fill, or the value that changes nothing)
                const float bz = (SUM != 0 && p.bnd_fill_on) ? p.bnd_fill : (SUM == 1 ? __builtin_inff() : -__builtin_inff());
                // (ULTRA_STREAM_DIET, rspmm_order_asm.hpp: the walk tests "is this lane's step inside its stream" as step < lim with the
                // step in an SGPR, lim = len - lane % 8; a marker record's type is num_rel -- times 256 in the pre-shifted records of
                // the twelve-walker schedules, which the hand-off forms read)
                constexpr int POST = UPDATE == 3 ? 2 : 0;      // (the generator's POST == 2: rows parked in LDS, stream_park)
#if ULTRA_STREAM_DIET
                const int l8_arg = len - (l16 & 7);
                const uint32_t rmk = (uint32_t)p.num_rel << ((POST != 0 && ULTRA_STREAM_PRESHIFT_GEN) ? 8 : 0);
#else
                const int l8_arg = l16 & 7;
                const uint32_t rmk = 0;
#endif
                if (ns > 0)
                    order_stream_asm<SUM, MUL, POST>(len, (uint32_t)(sd.x + (l16 & 7)) * 8u, l8_arg, lane_bytes, lds_addr(lds_rel_lane),
                                                     lds_addr(lds_rel_lane) + (uint32_t)p.num_rel * 256u, bndoff, bv, bz, ns, nf, xbase,
                                                     reinterpret_cast<const char *>(p.srec),
                                                     reinterpret_cast<const char *>(reinterpret_cast<const T *>(p.out) + outer * p.out_stride_outer),
                                                     p.x_row_bytes, lds_addr(const_cast<uint32_t *>(ctl)), lds_addr(ring), rmk);
                if constexpr (UPDATE == 3) {
                    // this wave has posted all its rows (the walk ends on vmcnt(0) + its last posts; LDS operations of a
                    // wave execute in order)
                    asm volatile("" ::: "memory");
                    // (the generated walk cannot reach OrderParams: a park wait that gave up left its code in the control block)
                    if (lane == 0 && ctl[UPD2_CTL_ERR] != 0) report_spin(ORDER_SPIN_PARK);
                    if (lane == 0) __hip_atomic_fetch_add(const_cast<uint32_t *>(ctl) + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            if constexpr (UPDATE == 3) {
                if (wave >= ORDER_WALKERS) {
                    // ---- update waves, form 3 (update_tile.hpp UPD2_*): wave u owns output features [16 u, 16 u + 16) ----
                    using f32x4m = float __attribute__((ext_vector_type(4)));
                    int lane_u = lane;
                    asm volatile("" : "+v"(lane_u));
                    const int u = wave - ORDER_WALKERS;
                    const int i16 = lane_u & 15, kk = lane_u >> 4;
                    // the walkers' ring: 64 aggregate rows [64][68], behind them the same rows' x [64][68] -- a block's pre-norm rows
                    // later take the place of its x rows
                    // (typed LDS pointers: through generic ones every access here becomes a FLAT instruction -- the texture-address path
                    // the walkers saturate -- instead of a DS one)
                    using lds_f = __attribute__((address_space(3))) float;
                    using lds_f4 = __attribute__((address_space(3))) f32x4m;     // (the native vector: HIP's float4 class has no address-space overloads)
                    using lds_vu = volatile __attribute__((address_space(3))) uint32_t;
                    using lds_u = __attribute__((address_space(3))) uint32_t;
                    lds_vu *const lctl = (lds_vu *)ctl;
                    lds_f *const tiles = (lds_f *)reinterpret_cast<float *>(ring), *const xring = tiles + UPD2_NT * UPD2_TILE_FLOATS;
                    lds_f *const c_tile = (lds_f *)reinterpret_cast<float *>(const_cast<uint32_t *>(ctl)) + UPD2_CTL_CTILE_BYTES / 4;
                    // A operands: this wave's 16 rows of W (lane (i, kk) holds W[16 u + i][4 s + kk]); small vectors
                    float wfrag[32];
#pragma unroll
                    for (int s = 0; s < 32; ++s) wfrag[s] = p.upd.weight[(16 * u + i16) * 128 + 4 * s + kk];
                    const int f0 = 16 * u + 4 * kk;      // D: lane l, reg r -> feature 16 u + 4 (l >> 4) + r of tile row l & 15
                    float biasv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) biasv[r] = p.upd.bias ? p.upd.bias[f0 + r] : 0.f;
                    if (c1 > c0) {
                        const uint32_t cap = SPIN_CAP + 16u * (uint32_t)(c1 - c0);
                        uint32_t polls = 0;
                        while (lctl[3] == 0) {   // the chain consumer still reads the ring
                            if (++polls >= cap) {
                                if (lane == 0) report_spin(ORDER_SPIN_CHAIN_UPDATER);
                                break;
                            }
                            __builtin_amdgcn_s_sleep(4);
                        }
                    }
                    const char *aggbase = reinterpret_cast<const char *>(reinterpret_cast<const T *>(p.out) + outer * p.out_stride_outer);
                    char *ubase = reinterpret_cast<char *>(p.upd.out + outer * p.upd.out_stride_outer);
                    uint32_t epoch = 0;
                    // Bounded spins (OrderParams::err).  A wait that has polled `cap` times reports, lets the walkers through (they
                    // never wait for a free ring row again) and ends this wave's work: the launch terminates with an error word
                    // instead of hanging the GPU -- results of the workgroup are then incomplete, and the host says so.
                    // ULTRA_SPIN_GUARD builds also leave the control words in the trace buffer (tools/spin_guard_probe.py).
                    int t_now = 0;
                    bool tripped = false;
                    const uint32_t look_cap = SPIN_CAP + 16u * (uint32_t)p.max_stream_steps;   // (a walker may be inside one long row)
                    const auto spin_guard = [&](uint32_t &n, const uint32_t cap, const int code) {
                        if (++n < cap) return false;
                        if (lane == 0 && !tripped) {
                            report_spin(code);
#if ULTRA_SPIN_GUARD
                            if (p.trace) {
                                long long *dst = p.trace + 24 * gridDim.x + 16 * blockIdx.x + 4 * u;
                                dst[0] = ((long long)code << 48) | ((long long)t_now << 32) | epoch;
                                dst[1] = ((long long)lctl[0] << 32) | lctl[1];
                                dst[2] = ((long long)lctl[2] << 32) | lctl[UPD2_CTL_CONSUMED];
                                dst[3] = ((long long)lctl[UPD2_CTL_POSTED] << 32) | lctl[UPD2_CTL_POSTED + 1];
                            }
#endif
                        }
                        tripped = true;
                        lctl[UPD2_CTL_CONSUMED] = 0x3fffffffu;     // the walkers are let through
                        return true;
                    };
                    (void)t_now;
#define ULTRA_SPIN(n, cap, code) if (spin_guard(n, cap, code)) break
                    // The four update waves meet in two halves (s_barrier would count the walkers too): `arrive` costs nothing -- LDS
                    // operations of a wave execute in order, so everything this wave did to LDS before is done when its arrival shows --
                    // and between arriving and `meet` the wave does work that does not depend on the others.  One counter serves every
                    // meeting: nobody arrives at meeting m + 1 before everybody has arrived at m.
                    // (CONV_DBG_LOSE_ARRIVAL, tests only: update wave 3 never arrives anywhere -- the protocol error the bounded
                    // spins exist for; the launch must end with an error word, not hang)
                    const bool lose_arrival = (p.upd.flags & CONV_DBG_LOSE_ARRIVAL) && u == 3;
                    const auto arrive = [&]() {
                        asm volatile("" ::: "memory");
                        epoch += (uint32_t)ORDER_UPDATERS;
                        if (lane == 0 && !lose_arrival)
                            __hip_atomic_fetch_add((lds_u *)lctl + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    };
                    const auto meet = [&]() {
                        uint32_t spins = 0;
                        while (!tripped && lctl[2] < epoch) {
                            ULTRA_SPIN(spins, SPIN_CAP, ORDER_SPIN_MEET);
                        }
                        asm volatile("" ::: "memory");
                    };
                    // An update wave's time is LDS round trips (several hundred cycles each while twelve waves walk: the LDS serves
                    // requests in arrival order) and it takes no memory round trip at all: the walkers park x[row] beside the row's
                    // aggregate (a marker step of a stream gathers at its own row's offset), so a complete block is ready to multiply.
                    // The unit of work is a BLOCK of 32 rows -- two of the walkers' generations, two independent matrix chains.  Per
                    // block: every wave sees for itself that the block is complete; reads its B operands; the waves meet once the x
                    // rows are read (under the matrix chains) -- then the pre-norm rows take the x rows' place -- and once those are
                    // written; every wave finishes 8 rows and, its LDS reads done, counts itself out of the block: at four the walkers
                    // may reuse its rows.  Blocks t = 0, 1, ...: first the workgroup's chain rows, 8 to a block (listed by the chain
                    // consumer, aggregates in memory: staged with their x into the control block's tile), then the walkers' generations
                    // in pairs.  Lane (kk, i16) of update wave u finishes 16 bytes of rows rr and 16 + rr of a block.
                    const int rr = 4 * u + kk;
                    const int n_chain_rows = (c1 > c0) ? min((int)lctl[UPD2_CTL_NCHAIN], UPD2_MAX_CHAIN_ROWS) : 0;
                    const int n_ctile = (n_chain_rows + 7) >> 3;
                    struct Block {              // (small on purpose: `next` lives across the finishing of `cur`; tile addresses follow from b0)
                        int n0, n1;             // rows of its two halves (n0 < 0: not complete yet; n0 == 0: there is none)
                        uint32_t off0, off1;    // byte offsets of this lane's two rows inside a batch slice
                        int b0;                 // first of its two ring tiles; -1: a chain block (8 rows, from memory, in the control block's tile)
                    };
                    const auto chain_block = [&](const int t) {
                        Block bl;
                        bl.n0 = min(8, n_chain_rows - 8 * t), bl.n1 = 0;
                        bl.off0 = lctl[UPD2_CTL_CROW + 8 * t + (rr & 7)], bl.off1 = 0u;
                        bl.b0 = -1;
                        return bl;
                    };
                    // generations 2 B and 2 B + 1 of the walkers' rows (`wait`: blocks until they are complete)
                    const auto look = [&](const int B, const bool wait) {
                        const int b0 = (2 * B) & (UPD2_NT - 1);
                        const uint32_t need = 16u * (uint32_t)(B / (UPD2_NT / 2) + 1);
                        Block bl;
                        bl.b0 = b0;
                        uint32_t spins = 0;
                        for (;;) {
                            // (`walked` is read FIRST and the row lists last: with every walker done the counts are final, and with
                            // a generation's count full its list is)
                            const uint32_t walked = lctl[1], have0 = lctl[UPD2_CTL_POSTED + b0], have1 = lctl[UPD2_CTL_POSTED + b0 + 1], tail = lctl[0];
                            bl.off0 = lctl[UPD2_CTL_ROWID + 16 * b0 + rr], bl.off1 = lctl[UPD2_CTL_ROWID + 16 * b0 + 16 + rr];
                            if (have0 >= need && have1 >= need) {
                                bl.n0 = bl.n1 = 16;
                                break;
                            }
                            if (walked == (uint32_t)ORDER_WALKERS) {
                                const int rem = (int)tail - 32 * B;
                                bl.n0 = max(0, min(16, rem)), bl.n1 = max(0, min(16, rem - 16));
                                break;
                            }
                            bl.n0 = bl.n1 = -1;
                            if (!wait) break;
                            ULTRA_SPIN(spins, look_cap, ORDER_SPIN_LOOK);
                            __builtin_amdgcn_s_sleep(1);
                        }
                        bl.n0 = rfl(bl.n0), bl.n1 = rfl(bl.n1);
                        return bl;
                    };
                    __builtin_amdgcn_s_setprio(ULTRA_UPD_PRIO);   // (the youngest waves of their SIMDs: without it every walker instruction goes first)
                    // measurement hook (trace[3 grid + 4 workgroup + k]): cycles update wave 0 spends k = 0 waiting for a block (and
                    // staging a chain block), 1 multiplying, 2 until the pre-norm rows are all written, 3 finishing
                    long long ph0 = 0, ph1 = 0, ph2 = 0, ph3 = 0, t_ph = p.trace ? clock64() : 0;
                    const auto lap = [&](long long &ph) {
                        if (p.trace) {
                            const long long now = clock64();
                            ph += now - t_ph;
                            t_ph = now;
                        }
                    };
                    bool ahead = false;         // `next` was seen complete while the previous block's pre-norm rows were being written
                    int next_n0 = 0, next_n1 = 0, next_b0 = 0;
                    for (int t = 0;; ++t) {
                        t_now = t;
                        if (tripped) break;     // (a wait of this wave gave up: see spin_guard)
                        Block cur;
                        if (ahead) {
                            // (only what is uniform was kept; the rows' offsets are needed at the very end: read again, never waited for)
                            cur.n0 = next_n0, cur.n1 = next_n1, cur.b0 = next_b0;
                            cur.off0 = lctl[UPD2_CTL_ROWID + 16 * cur.b0 + rr], cur.off1 = lctl[UPD2_CTL_ROWID + 16 * cur.b0 + 16 + rr];
                        } else {
                            cur = t < n_ctile ? chain_block(t) : look(t - n_ctile, true);
                            if (cur.n0 <= 0) break;
                        }
                        const bool have0 = rr < cur.n0, have1 = rr < cur.n1;
                        const bool from_memory = cur.b0 < 0;
                        const int rowmask = from_memory ? 7 : 15;     // (a chain block: tile rows 8..15 repeat 0..7)
                        const int tile0 = rfl(max(cur.b0, 0)) * UPD2_TILE_FLOATS;
                        lds_f *const px0 = from_memory ? c_tile : xring + tile0, *const px1 = from_memory ? c_tile : px0 + UPD2_TILE_FLOATS;
                        lds_f *const pa0 = from_memory ? c_tile + 8 * UPD2_ROW_FLOATS : tiles + tile0;
                        lds_f *const pa1 = from_memory ? pa0 : pa0 + UPD2_TILE_FLOATS;
                        if (from_memory) {
                            // update waves 0, 1: the 8 x rows; 2, 3: the 8 aggregate rows (16 lanes a row)
                            const int j = 4 * (u & 1) + kk;
                            const uint32_t off = j < cur.n0 ? lctl[UPD2_CTL_CROW + 8 * t + j] : 0u;
                            *reinterpret_cast<lds_f4 *>((u < 2 ? px0 : pa0) + j * UPD2_ROW_FLOATS + 4 * i16) =
                                *reinterpret_cast<const f32x4m *>((u < 2 ? xbase : aggbase) + off + 16u * (uint32_t)i16);
                            arrive();
                            meet();
                        }
                        lap(ph0);
                        const int ri = i16 & rowmask, rf = rr & rowmask;
                        f32x4m d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
                        f32x4m x0, x1;      // this lane's 16 bytes of the x rows it finishes (the residual)
#if ULTRA_UPD_SKIP != 1
#if ULTRA_UPD_SKIP == 2
#define UPD_MFMA(a, b, c) f32x4m{(c)[0] + (b), (c)[1], (c)[2], (c)[3]}
#define UPD_OP(expr) (expr)
#elif ULTRA_UPD_SKIP == 3
#define UPD_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
#define UPD_OP(expr) (wfrag[s])
#else
#define UPD_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
#define UPD_OP(expr) (expr)
#endif
#if ULTRA_UPD_SKIP == 4   /* measurement build (wrong results): the same matrix-pipe time as HALF as many, twice as long instructions */
#define UPD_CHAIN16(W0, Q, D)                                                                                          \
    _Pragma("unroll") for (int s = 0; s < 8; ++s) big = __builtin_amdgcn_mfma_f32_32x32x2f32(wfrag[(W0) + 2 * s], Q[2 * s], big, 0, 0, 0); \
    D[0] += big[0];
                        f32x16 big;
#pragma unroll
                        for (int r = 0; r < 16; ++r) big[r] = 0.f;
#else
#define UPD_CHAIN16(W0, Q, D) _Pragma("unroll") for (int s = 0; s < 16; ++s) D = UPD_MFMA(wfrag[(W0) + s], Q[s], D);
#endif
                        {
                            // four quarters of operands (x and aggregate of half 0, of half 1), each requested while the quarter
                            // before it is multiplied: one LDS round trip in the open instead of two, 32 operand registers.  (One
                            // chain at a time keeps the matrix pipe as busy as two interleaved: an instruction occupies it for its
                            // 8 passes.)
                            float q0[16], q1[16];
#pragma unroll
                            for (int s = 0; s < 16; ++s) q0[s] = UPD_OP(px0[ri * UPD2_ROW_FLOATS + 4 * s + kk]);
#pragma unroll
                            for (int s = 0; s < 16; ++s) q1[s] = UPD_OP(pa0[ri * UPD2_ROW_FLOATS + 4 * s + kk]);
                            x0 = *reinterpret_cast<const lds_f4 *>(px0 + rf * UPD2_ROW_FLOATS + 4 * i16);
                            asm volatile("" ::: "memory");
                            UPD_CHAIN16(0, q0, d0)
#pragma unroll
                            for (int s = 0; s < 16; ++s) q0[s] = UPD_OP(px1[ri * UPD2_ROW_FLOATS + 4 * s + kk]);
                            x1 = *reinterpret_cast<const lds_f4 *>(px1 + rf * UPD2_ROW_FLOATS + 4 * i16);
                            arrive();       // (this wave's reads of the x rows are served before its arrival shows)
                            UPD_CHAIN16(16, q1, d0)
#pragma unroll
                            for (int s = 0; s < 16; ++s) q1[s] = UPD_OP(pa1[ri * UPD2_ROW_FLOATS + 4 * s + kk]);
                            asm volatile("" ::: "memory");
                            UPD_CHAIN16(0, q0, d1)
                            UPD_CHAIN16(16, q1, d1)
                        }
#else
                        x0 = x1 = f32x4m{0.f, 0.f, 0.f, 0.f};
                        arrive();
#endif
                        lap(ph1);
                        meet();     // everybody has read the block's x rows: the pre-norm rows take their place (bias after the chain, like addmm)
                        if (i16 <= rowmask)
                            *reinterpret_cast<lds_f4 *>(px0 + i16 * UPD2_ROW_FLOATS + f0) =
                                f32x4m{d0[0] + biasv[0], d0[1] + biasv[1], d0[2] + biasv[2], d0[3] + biasv[3]};
                        if (cur.n1 > 0)
                            *reinterpret_cast<lds_f4 *>(px1 + i16 * UPD2_ROW_FLOATS + f0) =
                                f32x4m{d1[0] + biasv[0], d1[1] + biasv[1], d1[2] + biasv[2], d1[3] + biasv[3]};
                        arrive();
                        // (LayerNorm's weights of this lane's features 4 i16 + e: fetched per block, here, where the operand registers
                        // have just come free -- kept across the loop they would spill)
                        float lnw[4] = {1.f, 1.f, 1.f, 1.f}, lnb[4] = {0.f, 0.f, 0.f, 0.f};
                        if (p.upd.flags & CONV_LN) {
                            using glb_f4 = const __attribute__((address_space(1))) f32x4m;
                            const float *gw = p.upd.ln_w, *gb = p.upd.ln_b;
                            asm volatile("" : "+s"(gw), "+s"(gb));
                            const f32x4m vw = *((glb_f4 *)gw + i16), vb = *((glb_f4 *)gb + i16);
                            lnw[0] = vw.x, lnw[1] = vw.y, lnw[2] = vw.z, lnw[3] = vw.w;
                            lnb[0] = vb.x, lnb[1] = vb.y, lnb[2] = vb.z, lnb[3] = vb.w;
                        }
                        // while the others arrive: is the next block complete already?
                        ahead = false;
                        if (t + 1 >= n_ctile) {
                            const Block next = look(t + 1 - n_ctile, false);
                            ahead = next.n0 > 0;
                            next_n0 = next.n0, next_n1 = next.n1, next_b0 = rfl(next.b0);
                        }
                        meet();
                        lap(ph2);
                        float ya[4], yb[4];
                        const lds_f *row_a = px0 + rf * UPD2_ROW_FLOATS, *row_b = px1 + rf * UPD2_ROW_FLOATS;
                        {
                            const f32x4m va = *reinterpret_cast<const lds_f4 *>(row_a + 4 * i16), vb = *reinterpret_cast<const lds_f4 *>(row_b + 4 * i16);
                            ya[0] = va.x, ya[1] = va.y, ya[2] = va.z, ya[3] = va.w;
                            yb[0] = vb.x, yb[1] = vb.y, yb[2] = vb.z, yb[3] = vb.w;
                        }
                        if (p.upd.flags & CONV_LN) ln_row_group2(ya, yb, row_a, row_b, i16, p.upd.eps, lnw, lnb);
                        // this wave is done with the block's rows in LDS (its reads above are served before the count shows): at four
                        // the walkers may reuse them
                        asm volatile("" ::: "memory");
                        if (!from_memory && lane == 0)
                            __hip_atomic_fetch_add((lds_u *)lctl + UPD2_CTL_CONSUMED, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (p.upd.flags & CONV_RELU) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) ya[e] = fmaxf(ya[e], 0.f), yb[e] = fmaxf(yb[e], 0.f);
                        }
                        if (p.upd.flags & CONV_RESIDUAL) {
                            ya[0] += x0.x, ya[1] += x0.y, ya[2] += x0.z, ya[3] += x0.w;
                            yb[0] += x1.x, yb[1] += x1.y, yb[2] += x1.z, yb[3] += x1.w;
                        }
                        if (have0) *reinterpret_cast<float4 *>(ubase + cur.off0 + 16u * (uint32_t)i16) = make_float4(ya[0], ya[1], ya[2], ya[3]);
                        if (have1) *reinterpret_cast<float4 *>(ubase + cur.off1 + 16u * (uint32_t)i16) = make_float4(yb[0], yb[1], yb[2], yb[3]);
                        lap(ph3);
                    }
                    __builtin_amdgcn_s_setprio(0);
                    if (p.trace && u == 0 && lane == 0) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) p.trace[3 * gridDim.x + 4 * blockIdx.x + k] = k == 0 ? ph0 : k == 1 ? ph1 : k == 2 ? ph2 : ph3;
                    }
                }
            }
            // (measurement hook: when each wave's walk ended, trace[8 * grid + 16 * workgroup + wave]; the buffer holds 32 * grid words)
            if (p.trace && lane == 0) p.trace[8 * gridDim.x + 16 * blockIdx.x + wave] = clock64();
            if constexpr (UPDATE == 1) {
                // ---- layer update of this workgroup's own rows (whole spans only: row_len == 64) ----
                // Every flush above has completed (each walk ends on vmcnt(0); the chain consumer's stores are collected
                // by the barrier's wait), and a workgroup's waves share their CU's vector L1: after the barrier the
                // aggregate rows are readable.  The relation slice is dead: its LDS takes the weight image.
                // (Measured and dropped -- the tail is a burst of 60 MB of x and aggregate rows across the chip, then 12 us of
                // matrix work, and every workgroup reaches it at the same time: the weight image staged into the ring's
                // place right after the chain phase + the x rows requested before the barrier, 100.7 us per layer instead
                // of 95.1; two shifts -- waves 0..7 request and multiply while waves 8..15 stage the image and request
                // behind them -- 97.8 us: half the burst takes as long as the whole one; each chunk's operand swaps right in
                // front of its matrix instructions, so that the chain starts on the first chunk's arrival -- 89.1 vs 89.3 us.)
                __syncthreads();
                // (measurement hook: tail timestamps at trace[3 * grid + 4 * workgroup + {0: walks done, 1: weights staged,
                // 2: wave 0's operands landed, 3: wave 0's tile done}])
                long long *utrace = p.trace ? p.trace + 3 * gridDim.x + 4 * blockIdx.x : nullptr;
                if (utrace && tid == 0) utrace[0] = clock64();
                float *lds_w = reinterpret_cast<float *>(smem);
                int tid_u = tid;
                asm volatile("" : "+v"(tid_u));
                update_stage_weights(lds_w, p.upd.weight, p.upd.bias, p.upd.ln_w, p.upd.ln_b, p.upd.flags, tid_u, ORDER_THREADS);
                __syncthreads();
                if (utrace && tid == 0) utrace[1] = clock64();
                const int lane_u = tid_u & 63, ju = lane_u & 31, hu = lane_u >> 5;
                const int r1 = p.upd.prow_ptr[part + 1];
                const char *aggbase = reinterpret_cast<const char *>(reinterpret_cast<const T *>(p.out) + outer * p.out_stride_outer);
                float *ubase = p.upd.out + outer * p.upd.out_stride_outer;
                for (int base = p.upd.prow_ptr[part] + 32 * wave; base < r1; base += 32 * nwave) {
                    const int row = p.upd.prow[base + ju];
                    const bool valid = row >= 0;
                    const uint32_t roff = (uint32_t)(valid ? row : 0) * p.x_row_bytes + (uint32_t)hu * 16u;
                    const float4 *xr = reinterpret_cast<const float4 *>(xbase + roff);
                    const float4 *ar = reinterpret_cast<const float4 *>(aggbase + roff);
                    float4 b[16];
#pragma unroll
                    for (int i = 0; i < 8; ++i) b[i] = xr[2 * i];
#pragma unroll
                    for (int i = 0; i < 8; ++i) b[8 + i] = ar[2 * i];
                    if (utrace && tid == 0) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        utrace[2] = clock64();
                    }
                    update_tile(b, lds_w, lane_u, p.upd.flags, p.upd.eps, ubase + (long long)(valid ? row : 0) * p.upd.out_stride_row, valid);
                    if (utrace && tid == 0) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        utrace[3] = clock64();
                    }
                }
            }
        } else {
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
                } else if (SUM != 0 && p.has_bnd && p.bnd_fill_on) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc.v[e] = nary<T, SUM>(acc.v[e], fill);
                }
                T *dst = reinterpret_cast<T *>(p.out) + outer * p.out_stride_outer + (long long)row * p.out_stride_row + d0;
                *reinterpret_cast<P *>(dst) = acc;
            }
        }
        }   // (unit walk)
    }
    if (p.trace) {
        __syncthreads();
        if (tid == 0) p.trace[3 * blockIdx.x + 2] = clock64();
        if constexpr (UPDATE == 3) {   // (of the LAST span: how often this workgroup's walkers found the hand-off ring full)
            if (tid == 0) p.trace[24 * gridDim.x + blockIdx.x] = ctl[UPD2_CTL_RETRIES];
        }
    }
}

// ---- per-variant launchers (explicitly instantiated in rspmm_order_*.hip) ----
template <typename T, int SUM, int MUL, bool REL_LDS, bool WEIGHTED, bool STREAMS, int UPDATE = 0>
inline hipError_t launch_order_inst(const OrderParams &p, int grid, size_t lds, hipStream_t s) {
    auto kern = rspmm_order_kernel<T, SUM, MUL, REL_LDS, WEIGHTED, STREAMS, UPDATE>;
    static size_t lds_opted_in = 0;   // (see launch_one in rspmm_kernels.hpp)
    if (lds > 48 * 1024 && lds > lds_opted_in) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_opted_in = lds;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(ORDER_THREADS), lds, s, p);
    return hipGetLastError();
}
template <typename T, int SUM, int MUL, bool REL_LDS, bool WEIGHTED>
inline hipError_t launch_order_one(const OrderParams &p, int grid, size_t lds, hipStream_t s) {
    if constexpr (OrderAsm<T, MUL, REL_LDS, WEIGHTED>::value) {
        if constexpr (sizeof(T) == 4) {
            if (p.use_streams && p.upd.weight)
                return p.upd.mode == 3 ? launch_order_inst<T, SUM, MUL, REL_LDS, WEIGHTED, true, 3>(p, grid, lds, s)
                                       : launch_order_inst<T, SUM, MUL, REL_LDS, WEIGHTED, true, 1>(p, grid, lds, s);
        }
        if (p.use_streams) return launch_order_inst<T, SUM, MUL, REL_LDS, WEIGHTED, true>(p, grid, lds, s);
    }
    if (p.upd.weight) return hipErrorInvalidValue;   // (the caller checks use_streams first)
    return launch_order_inst<T, SUM, MUL, REL_LDS, WEIGHTED, false>(p, grid, lds, s);
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
