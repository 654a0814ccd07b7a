// One whole GeneralizedRelationalConv layer on a dense-format plan, IN THE REFERENCE'S SUMMATION ORDER:
//
//     agg[i] = (((0 + m(i, j0, t0)) + m(i, j0, t1)) + ... )   over the edges of row i in sorted (col, edge id) order,
//              m(i, j, t) = rel[b, t] * x[b, j]                                              rspmm.cpp:61-72
//     out    = [x +] relu( LayerNorm( W . cat[x, agg + boundary] + bias ) )                   layers.py:199-200, 233-240
//
// ULTRA's relation graph (a few hundred nodes, 4 edge types, nearly every (row, type, col) cell occupied) gives every row
// ~1,900 edges: one long sequential sum per output element.  On the matrix pipe that sum costs nothing extra:
// v_mfma_f32_16x16x4_f32 is, bit for bit, a k-ordered chain of fp32 fmaf (cdna_hip_programming.md, section 3), so with
//     A[i][k] = 1 if the edge (row i <- col j, type k) exists else 0,     B[k][n] = fl(rel[k][n] * x[j][n])
// one instruction per source column j performs  acc = fma(A[i][3], B[3], fma(A[i][2], B[2], fma(A[i][1], B[1],
// fma(A[i][0], B[0], acc))))  -- fma(1, b, c) = fl(c + b) and fma(0, b, c) = c exactly (finite b) -- i.e. it adds the
// separately rounded messages of column j's up to four parallel edges in type order, and the columns follow in
// ascending order: exactly the reference's loop, provided parallel edges are sorted by type (checked when the plan is
// built: ULTRA's relation graphs list hh, tt, ht, th in that order, tasks.py:186-189) and no edge is repeated.
//
// A workgroup (4 waves) owns 16 output rows of one sample; wave w owns the 16 features [16 w, 16 w + 16).  Its chain of
// num_node dependent matrix instructions (40 cycles each) is the critical path: ~8 us at 474 nodes; the operands (one
// byte of adjacency, one float of x per instruction and lane) are prefetched 16 columns ahead.  Phases 2 / 3: boundary,
// update product on the matrix pipe with k ascending (= the reference's nn.Linear chain), LayerNorm in the reference's
// operation order (torch_math.hpp), ReLU, residual.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <string>

#ifndef ULTRA_DOL_ASM
#define ULTRA_DOL_ASM 0
#endif
#if ULTRA_DOL_ASM   // measurement builds only: the header is GENERATED for them (python tools/gen_dense_order_asm.py writes
                    // ultra_amd/lib/variants/dense_order_asm.hpp; tools/build_variant.py NAME -DULTRA_DOL_ASM=1
                    // -DULTRA_DOL_ASM_HEADER=\"<path>\") -- it is not part of the source tree or of the default build
#include ULTRA_DOL_ASM_HEADER
#endif
#include "plan.hpp"
#include "torch_math.hpp"

// MEASUREMENT BUILD (-DULTRA_DOL_ASM=1): the chain of phase 1 as generated assembly (tools/gen_dense_order_asm.py) -- four
// stages of operands in flight, every wait counted, optionally a touch of the whole x slice up front.  hipcc's loop (below: three
// stages) drains its load queue once per iteration (s_waitcnt vmcnt(0) a third into the body), which looked like the reason
// why the chain runs at 67 cycles per column against the instruction's 32-36.  It is not: bit-exact, and 21.7 us per layer
// against the C++ loop's 21.5 (with the touch: 22.6) on one box (round 4, tools/dense_order_probe.py) -- the chain does not wait
// for memory; the byte -> float conversion, the product and the load issue of a column do not hide under its matrix instruction.
#pragma clang fp contract(off)

namespace ultra {

using f32x4 = float __attribute__((ext_vector_type(4)));

enum { DOL_LN = 1, DOL_RELU = 2, DOL_RESIDUAL = 4 };

struct DenseOrderParams {
    const uint4 *a_ex;        // [n_rt16][n_jc][64 lanes] : 16 bytes = A[row0 + lane % 16][type lane / 16][16 jc + 0..15]
    const float *rel, *x, *bnd;
    const long long *bnd_rows;
    const float *weight, *bias, *ln_w, *ln_b;
    float *out;
    long long rel_so, rel_sr, x_so, x_sr, bnd_so, bnd_sr, out_so, out_sr;
    int n_out, n_in, n_rel, n_jc, n_rt16, has_bnd, flags;
    float eps;
};

constexpr int DOL_ROW_STRIDE = 68;   // floats per LDS tile row (see dense_layer.hip)

__device__ __forceinline__ float byte_of(uint32_t w, int q) { return (float)((w >> (8 * q)) & 0xffu); }

// ULTRA_DOL_VGPR_CAP (measurement builds): keep the kernel to 128 registers per lane, so that one of its workgroups (one wave
// per SIMD) fits beside a 12-wave reference-order workgroup of ANOTHER launch on the same CU (plan.hpp ULTRA_ORDER_WAVES)
#ifdef ULTRA_DOL_VGPR_CAP
#define ULTRA_DOL_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#else
#define ULTRA_DOL_ATTR
#endif
// TILES: 16-row tiles of one workgroup (1 or 2).  Two tiles share every B operand of the chain -- the messages rel * x[column] are
// the same for all rows; only the adjacency bytes and the accumulator differ -- so a wave runs two independent chains over one
// stream of operands: half the workgroups (120 instead of 240 at 474 nodes x 8 samples), half the x traffic, and the second
// chain's matrix instruction hides the byte -> float conversions and the product that one chain per SIMD leaves in the open.
// Chosen where the launch does not own the chip (two batches in flight: the relation-graph layers of one batch run on the 64 CUs
// the other batch's entity layers leave -- 240 workgroups need two rounds there, 120 one); same bits either way.
// LEAN: the same body in at most 128 registers a lane, so that FOUR of its workgroups share a CU (four waves per SIMD, their
// chains interleaved on the matrix pipe: 4 x 32 cycles per column where one wave alone takes 67).  With batches in flight the
// relation-graph layers of one batch run on the 64 CUs the other batches' entity layers leave: 240 workgroups at three a CU
// (162 registers) need two rounds there -- 58 us a layer as timed against 22.5 alone -- at four a CU one.  What is
// given up: the operands of phases 2 / 3 (update weights, own x rows, boundary, small vectors) are requested AFTER the chain
// instead of before it (their latency shows once per workgroup), and the offsets of the last, clamped stage are computed where
// they are used instead of being held in registers.  Same instruction sequence on the data: the same bits.
// SUBS: quartets of waves per workgroup (1 or 4), each quartet a row tile of its own.  The hardware spreads the workgroups of a
// launch over whatever CUs have room; a CU that holds ONE 256-thread workgroup of this kernel no longer has the LDS (and, at four a
// SIMD, the registers) for a workgroup of the entity layer, which wants the whole CU -- so 240 small workgroups can block up to 240
// CUs for the length of a chain that keeps one wave a SIMD busy.  Sixteen waves in ONE workgroup pack four chains onto every SIMD
// of one CU by construction: 60 workgroups a layer instead of 240, the matrix pipe of the CUs they do take busy 128 of 128 cycles a
// column.  (With batches in flight the step is bound by CU-time: profiles/r5_experiments.txt.)
template <int TILES, bool LEAN, int SUBS = 1>
__device__ __forceinline__ void dense_order_layer_impl(const DenseOrderParams &p) {
    __shared__ __attribute__((aligned(16))) float x_lds_all[SUBS * TILES * 16 * DOL_ROW_STRIDE];     // the tiles' own rows of x
    __shared__ __attribute__((aligned(16))) float agg_lds_all[SUBS * TILES * 16 * DOL_ROW_STRIDE];
    __shared__ float ln_mom_all[SUBS][TILES * 16][8][2];
    __shared__ float ln_stat_all[SUBS][TILES * 16][2];
    const int sub = SUBS > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;
    float *x_lds = x_lds_all + sub * (TILES * 16 * DOL_ROW_STRIDE), *agg_lds = agg_lds_all + sub * (TILES * 16 * DOL_ROW_STRIDE);
    float(&ln_mom)[TILES * 16][8][2] = ln_mom_all[sub];
    float(&ln_stat)[TILES * 16][2] = ln_stat_all[sub];
    const int tid = threadIdx.x & 255, lane = tid & 63;      // (tid: within the quartet)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kk = lane >> 4;
    const int n_wt = (p.n_rt16 + TILES * SUBS - 1) / (TILES * SUBS);       // workgroups per sample
    const int rt_raw = (blockIdx.x % n_wt) * SUBS + sub, outer = blockIdx.x / n_wt;
    // (a quartet past the graph walks the last tile again -- every wave takes part in the workgroup's barriers -- and stores nothing)
    const bool quartet_live = rt_raw * TILES < p.n_rt16;
    const int rt = quartet_live ? rt_raw : (p.n_rt16 - 1) / TILES;
    const float *xo = p.x + (long long)outer * p.x_so;
    const int row0 = rt * 16 * TILES;
    const int c0 = 16 * wave;
    // (a second tile past the graph: reads the last tile's adjacency, its rows are never stored)
    const bool tile1 = TILES > 1 && TILES * rt + 1 < p.n_rt16;

    // ---- phase 1 operands: all requested before anything waits ----
    // One wave per SIMD: whatever the wave does between two matrix instructions of the chain stalls the chain.  The fetch
    // of a stage is therefore pure load issue: a uniform (scalar) base per stage + per-lane byte offsets computed once
    // (global_load v, v_off, s[base]) -- no multiplies, no clamps in the loop.  Only the last stage can touch columns past
    // n_in (their adjacency bytes are 0); it has its own, clamped offsets.
    const uint4 *ap[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t) ap[t] = p.a_ex + (size_t)min(TILES * rt + t, p.n_rt16 - 1) * p.n_jc * 64;
    const char *xbase = reinterpret_cast<const char *>(xo);
    const uint32_t x_row_bytes = (uint32_t)p.x_sr * 4u;
    const uint32_t lane_bytes = (uint32_t)(c0 + i16) * 4u;       // B operand: lane (kk, n) reads x[j][c0 + n]
#if !ULTRA_DOL_ASM
    uint32_t voff[16], voff_last[LEAN ? 1 : 16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        voff[q] = (uint32_t)q * x_row_bytes + lane_bytes;
        if (!LEAN) voff_last[q] = (uint32_t)min(16 * (p.n_jc - 1) + q, p.n_in - 1) * x_row_bytes + lane_bytes;
    }
    // (LEAN: the last stage's offsets relative to ITS base, clamped to the graph's last row, from one register)
    const uint32_t off_max = (uint32_t)(p.n_in - 1 - 16 * (p.n_jc - 1)) * x_row_bytes + lane_bytes;
    struct Stage {
        uint4 a[TILES];
        float x[16];
    };
    const auto fetch = [&](const int jc, Stage &st) {
        const bool last = jc >= p.n_jc - 1;                                   // uniform
        const int jcc = last ? p.n_jc - 1 : jc;
#pragma unroll
        for (int t = 0; t < TILES; ++t) st.a[t] = ap[t][(size_t)jcc * 64 + lane];
        if (LEAN) {
            const char *sbase = xbase + (uint32_t)jcc * 16u * x_row_bytes;
            const uint32_t cap = last ? off_max : 0xffffffffu;
#pragma unroll
            for (int q = 0; q < 16; ++q) st.x[q] = *reinterpret_cast<const float *>(sbase + min(voff[q], cap));
        } else {
            const char *sbase = xbase + (last ? 0u : (uint32_t)jcc * 16u * x_row_bytes);
#pragma unroll
            for (int q = 0; q < 16; ++q) st.x[q] = *reinterpret_cast<const float *>(sbase + (last ? voff_last[q] : voff[q]));
        }
    };
    // three stages (48 source columns, ~1,900 cycles of chain) in flight: one wave per SIMD has nothing else to hide an
    // L2 round trip behind
    Stage st0, st1, st2;
    fetch(0, st0);
    fetch(1, st1);
    fetch(2, st2);
#endif
    const float relv = kk < p.n_rel ? p.rel[(long long)outer * p.rel_so + (long long)kk * p.rel_sr + c0 + i16] : 0.f;

    // the tiles' own x rows (update input and residual), the update weights of this wave's feature tile and the small
    // vectors of phases 2 / 3: requested now, consumed after the chain (LEAN: requested after the chain)
    float4 xtile[TILES];
    float wfrag[32];   // A operand of phase 3: lane (i, kk) holds W[16 ft + i][4 s + kk], ft = wave
    const int f0 = 16 * wave + 4 * kk;   // first of this lane's 4 features in phase 3
    float biasv[4], lnw[4], lnb[4];
    float bndv[TILES][4];   // boundary addends of the accumulator elements (row 4 kk + r of a tile, column c0 + i16)
    const auto fetch_late_operands = [&]() {
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            const int row = min(row0 + 16 * t + (tid >> 4), p.n_out - 1);
            xtile[t] = *reinterpret_cast<const float4 *>(xo + (long long)row * p.x_sr + 4 * (tid & 15));
        }
#pragma unroll
        for (int s = 0; s < 32; ++s) wfrag[s] = p.weight[(16 * wave + i16) * 128 + 4 * s + kk];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            biasv[r] = p.bias ? p.bias[f0 + r] : 0.f;
            lnw[r] = (p.flags & DOL_LN) ? p.ln_w[f0 + r] : 1.f;
            lnb[r] = (p.flags & DOL_LN) ? p.ln_b[f0 + r] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) bndv[t][r] = 0.f;
        if (p.has_bnd) {
            const long long bnd_row = p.bnd_rows ? p.bnd_rows[outer] : -1;
#pragma unroll
            for (int t = 0; t < TILES; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = row0 + 16 * t + 4 * kk + r;
                    if (row < p.n_out && (bnd_row < 0 || bnd_row == row))
                        bndv[t][r] = p.bnd[(long long)outer * p.bnd_so + (long long)row * p.bnd_sr + c0 + i16];   // (bnd_sr == 0 for a point)
                }
        }
    };
    if (!LEAN) fetch_late_operands();

    // ---- phase 1: the chain(s) ----
    f32x4 acc[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
#if ULTRA_DOL_ASM
    static_assert(TILES == 1, "the assembly chain of the measurement build walks one tile");
    dense_order_chain_asm(acc[0], p.n_jc, p.n_in, (uint32_t)lane * 16u, ap[0], xbase, lane_bytes, x_row_bytes, relv, (uint32_t)tid * 128u);
#else
    const auto chain = [&](const Stage &cur, const int jc) {
        // (stages past the graph chain zeros: fma(0, b, acc) = acc exactly -- rounds have no conditional exit, which
        // would be a join where the compiler stops counting outstanding loads and drains the queue)
        const uint32_t m = jc < p.n_jc ? 0xffffffffu : 0u;
        uint32_t aw[TILES][4];
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            aw[t][0] = cur.a[t].x & m, aw[t][1] = cur.a[t].y & m, aw[t][2] = cur.a[t].z & m, aw[t][3] = cur.a[t].w & m;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float b = relv * cur.x[q];          // the message, rounded on its own like rspmm.cpp:67
#pragma unroll
            for (int t = 0; t < TILES; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(byte_of(aw[t][q >> 2], q & 3), b, acc[t], 0, 0, 0);
        }
    };
    for (int jc = 0; jc < p.n_jc; jc += 3) {
        chain(st0, jc);
        fetch(jc + 3, st0);
        chain(st1, jc + 1);
        fetch(jc + 4, st1);
        chain(st2, jc + 2);
        fetch(jc + 5, st2);
    }
#endif

    if (LEAN) fetch_late_operands();

    // ---- phase 2: + boundary (layers.py:199-200), aggregate tiles and x tiles to LDS ----
    // D layout: lane l, reg r -> tile row 4 (l >> 4) + r, column l & 15
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = acc[t][r];
            if (p.has_bnd) v += bndv[t][r];   // (exactly 0 where there is no boundary value: same bits as not adding)
            agg_lds[(16 * t + 4 * kk + r) * DOL_ROW_STRIDE + c0 + i16] = v;
        }
        *reinterpret_cast<float4 *>(x_lds + (16 * t + (tid >> 4)) * DOL_ROW_STRIDE + 4 * (tid & 15)) = xtile[t];
    }
    __syncthreads();

    // ---- phase 3: update (feature tile ft = wave): B operand lane (kk, j) holds data[row j][4 s + kk], data = cat[x, agg] ----
    f32x4 d[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) d[t][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s)
            d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfrag[s], x_lds[(16 * t + i16) * DOL_ROW_STRIDE + 4 * s + kk], d[t], 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 16; ++s)
            d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfrag[16 + s], agg_lds[(16 * t + i16) * DOL_ROW_STRIDE + 4 * s + kk], d[t], 0, 0, 0);
    }
    // D: lane l, reg r -> feature 16 ft + 4 (l >> 4) + r of tile row l & 15
    float y[TILES][4];
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) y[t][r] = d[t][r] + biasv[r];
    if (p.flags & DOL_LN) {
        // LayerNorm in the reference's operation order (torch_math.hpp): the pre-norm tiles go through LDS (agg_lds is
        // free again), 8 threads per row run the Welford accumulators i = 0..7 over features 8 j + i, one merges them.
        __syncthreads();   // every wave is done reading agg_lds as its B operand
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) agg_lds[(16 * t + i16) * DOL_ROW_STRIDE + f0 + r] = y[t][r];
        __syncthreads();
        if (tid < 128 * TILES) {
            const int row = tid >> 3, i = tid & 7;
            float xv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[j] = agg_lds[row * DOL_ROW_STRIDE + 8 * j + i];
            const Moments w = welford8(xv);
            ln_mom[row][i][0] = w.m1;
            ln_mom[row][i][1] = w.m2;
        }
        __syncthreads();
        if (tid < 16 * TILES) {
            Moments all[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) all[i] = Moments{ln_mom[tid][i][0], ln_mom[tid][i][1]};
            float mean, rstd;
            merge8(all, p.eps, mean, rstd);
            ln_stat[tid][0] = mean;
            ln_stat[tid][1] = rstd;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            const float mean = ln_stat[16 * t + i16][0], rstd = ln_stat[16 * t + i16][1];
#pragma unroll
            for (int r = 0; r < 4; ++r) y[t][r] = ln_apply(y[t][r], mean, rstd, lnw[r], lnb[r]);
        }
    }
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
        if (p.flags & DOL_RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) y[t][r] = fmaxf(y[t][r], 0.f);
        }
        if (p.flags & DOL_RESIDUAL) {
#pragma unroll
            for (int r = 0; r < 4; ++r) y[t][r] += x_lds[(16 * t + i16) * DOL_ROW_STRIDE + f0 + r];
        }
        const int row = row0 + 16 * t + i16;
        if (row < p.n_out && (t == 0 || tile1) && quartet_live)
            *reinterpret_cast<float4 *>(p.out + (long long)outer * p.out_so + (long long)row * p.out_sr + f0) =
                make_float4(y[t][0], y[t][1], y[t][2], y[t][3]);
    }
}

template <int TILES>
__global__ void __launch_bounds__(256) ULTRA_DOL_ATTR dense_order_layer_kernel(const DenseOrderParams p) {
    dense_order_layer_impl<TILES, false>(p);
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) dense_order_layer_lean_kernel(const DenseOrderParams p) {
    dense_order_layer_impl<1, true>(p);
}

__global__ void __launch_bounds__(1024) dense_order_layer_packed_kernel(const DenseOrderParams p) {
    dense_order_layer_impl<1, true, 4>(p);
}

static bool dol_ok16(const ultra_mat *m) {
    return (reinterpret_cast<uintptr_t>(m->ptr) & 15u) == 0 && m->stride_row % 4 == 0 && m->stride_outer % 4 == 0;
}

// Called by ultra_nbf_dense_layer (rspmm_api.hip) with the plan uploaded, when the caller asks for the reference order.
int launch_dense_order_layer(ultra_plan *p, const ultra_mat *rel, const ultra_mat *x, const ultra_mat *bnd, const int64_t *bnd_rows,
                             const void *weight, const void *bias, const void *ln_w, const void *ln_b, float eps, int flags,
                             const ultra_mat *out, hipStream_t stream, bool shared_chip) {
    if (!(p->flags & ULTRA_PLAN_DENSE) || p->a_ex.empty() || !p->d.a_ex) {
        set_error("ultra_nbf_dense_layer (reference order) needs a ULTRA_PLAN_DENSE plan of a graph with at most 4 relation types "
                  "whose parallel edges are sorted by type and never repeated");
        return ULTRA_ERR_UNSUPPORTED;
    }
    if (out->row_len != 64 || p->num_out != p->num_in) {
        set_error("ultra_nbf_dense_layer: hidden dim 64 on a square graph only");
        return ULTRA_ERR_UNSUPPORTED;
    }
    if (!dol_ok16(rel) || !dol_ok16(x) || !dol_ok16(out) || (bnd && !dol_ok16(bnd)) || (reinterpret_cast<uintptr_t>(weight) & 15u)) {
        set_error("ultra_nbf_dense_layer: operands must be 16-byte aligned with strides that are multiples of 4");
        return ULTRA_ERR_UNSUPPORTED;
    }
    if ((uint64_t)p->num_in * (uint64_t)x->stride_row * 4u >= (1ull << 32)) {
        set_error("ultra_nbf_dense_layer: an input slice (rows * stride_row) exceeds 4 GiB");
        return ULTRA_ERR_UNSUPPORTED;
    }
    DenseOrderParams dp;
    dp.a_ex = reinterpret_cast<const uint4 *>(p->d.a_ex);
    dp.rel = static_cast<const float *>(rel->ptr);
    dp.x = static_cast<const float *>(x->ptr);
    dp.bnd = bnd ? static_cast<const float *>(bnd->ptr) : nullptr;
    dp.bnd_rows = bnd ? reinterpret_cast<const long long *>(bnd_rows) : nullptr;
    dp.weight = static_cast<const float *>(weight);
    dp.bias = static_cast<const float *>(bias);
    dp.ln_w = static_cast<const float *>(ln_w);
    dp.ln_b = static_cast<const float *>(ln_b);
    dp.out = static_cast<float *>(out->ptr);
    dp.rel_so = rel->stride_outer, dp.rel_sr = rel->stride_row;
    dp.x_so = x->stride_outer, dp.x_sr = x->stride_row;
    dp.bnd_so = bnd ? bnd->stride_outer : 0, dp.bnd_sr = (bnd && !bnd_rows) ? bnd->stride_row : 0;
    dp.out_so = out->stride_outer, dp.out_sr = out->stride_row;
    dp.n_out = (int)p->num_out, dp.n_in = (int)p->num_in, dp.n_rel = (int)p->num_rel;
    dp.n_jc = (int)((p->num_in + 15) / 16), dp.n_rt16 = (int)((p->num_out + 15) / 16);
    dp.has_bnd = bnd ? 1 : 0;
    dp.flags = flags;
    dp.eps = eps;
    const char *tiles_str = std::getenv("ULTRA_DOL_TILES");      // (read per launch: the test toggles it)
    const int tiles_env = tiles_str ? std::atoi(tiles_str) : 0;
    // Measured (round 5, tools/dense_order_probe.py and tools/step_probe.py on one box): the two-tile form runs 32.2 us per layer
    // against 23.7 stand-alone -- its two chains share one matrix pipe, 64 cycles a column -- and the step with two batches in
    // flight 0.603 ms against 0.581: the relation-graph layers on the 64 free CUs are not what bounds that step.  So one tile
    // per workgroup stays the choice everywhere; ULTRA_DOL_TILES=2 selects the other form (same bits: tests/test_order_gpu.py).
    // ULTRA_DOL_LEAN=0 / 1 overrides the choice of the four-workgroups-a-CU form (default: where the launch shares the chip).
    const char *lean_str = std::getenv("ULTRA_DOL_LEAN");
    // 0: the 160-register form; 1: four 256-thread workgroups a CU; 2 (default where the launch shares the chip): ONE 1024-thread
    // workgroup of four quartets -- the packing by construction
    const int lean = (ULTRA_DOL_ASM || tiles_env == 2) ? 0 : (lean_str ? std::atoi(lean_str) : (shared_chip ? 2 : 0));
    const int tiles = (ULTRA_DOL_ASM || dp.n_rt16 < 2) ? 1 : (tiles_env == 2 ? 2 : 1);
    const long long blocks = (long long)((dp.n_rt16 + tiles - 1) / tiles) * out->n_outer;
    if (tiles == 2) {
#if !ULTRA_DOL_ASM
        hipLaunchKernelGGL(dense_order_layer_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, stream, dp);
#endif
    } else if (lean == 2 && dp.n_rt16 >= 4) {
#if !ULTRA_DOL_ASM
        const long long packed = (long long)((dp.n_rt16 + 3) / 4) * out->n_outer;
        hipLaunchKernelGGL(dense_order_layer_packed_kernel, dim3((unsigned)packed), dim3(1024), 0, stream, dp);
#endif
    } else if (lean) {
#if !ULTRA_DOL_ASM
        hipLaunchKernelGGL(dense_order_layer_lean_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dp);
#endif
    } else {
        hipLaunchKernelGGL(dense_order_layer_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream, dp);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error(std::string("dense_order_layer_kernel launch: ") + hipGetErrorString(e));
        return ULTRA_ERR_HIP;
    }
    return ULTRA_OK;
}

}  // namespace ultra
