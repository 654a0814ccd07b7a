// The NBFNet layer update on one 32-row tile (layers.py:233-240 + models.py:158-160):
//
//   out = [x +] relu( LayerNorm( W . [x ; agg] + b ) )        W: (64, 128) row-major, x / agg / out rows of 64 floats
//
// as the TRANSPOSED product D[feature][row] on v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chains over k = 0, 1, 2, ...: the
// reference's nn.Linear order, torch_math.hpp), so that a data row's 64 features land in the lane pair (lane, lane ^ 32)
// and LayerNorm needs one cross-lane exchange.  Shared by conv_update_kernel (dense_kernels.hip: the stand-alone launch)
// and by the tail of the reference-order rspmm kernel (rspmm_order_kernels.hpp: the update applied by the workgroup that
// aggregated the rows).
#pragma once

#include <hip/hip_runtime.h>

#include "torch_math.hpp"

namespace ultra {

using f32x16 = float __attribute__((ext_vector_type(16)));

enum {
    CONV_LN = 1,
    CONV_RELU = 2,
    CONV_RESIDUAL = 4,
    CONV_DBG_NO_MATRIX = 256,  /* measurement: skip the matrix chain */
    CONV_DBG_NO_UPDATE = 512,  /* measurement (update beside the walk): the update waves only drain their queue */
    CONV_DBG_LOSE_ARRIVAL = 1024 /* tests (form 3): one update wave never arrives at a meeting -- the bounded spins must end the launch */
};

// feature owned by accumulator register r of feature tile m in lane half h (32x32 C/D layout)
__device__ __forceinline__ int feat_of(int m, int r, int h) { return 32 * m + (r & 3) + 8 * (r >> 2) + 4 * h; }

// v_permlane32_swap: lanes 32..63 of `lo_pair` trade places with lanes 0..31 of `hi_pair`.  A lane half h holds the
// 16-byte chunk k = 8 i + 4 h .. + 3 of its row; after swapping (.x, .y) and (.z, .w) the four registers hold, in lane
// half h, element 2 s + h of the k pairs s = 4 i, 4 i + 2 and 4 i + 1, 4 i + 3: each v_mfma_f32_32x32x2_f32 then consumes
// two CONSECUTIVE k, and the accumulator chain runs over k = 0, 1, 2, ... exactly like the reference's nn.Linear
// (torch_math.hpp).
__device__ __forceinline__ void swap32(float &lo_pair, float &hi_pair) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo_pair), __float_as_uint(hi_pair), false, false);
    lo_pair = __uint_as_float(r[0]);
    hi_pair = __uint_as_float(r[1]);
}

// LayerNorm statistics of a row held by the lane pair (lane, lane ^ 32) in the accumulator layout of the transposed
// product: lane half h owns features 32 m + (r & 3) + 8 (r >> 2) + 4 h, i.e. ALL eight members 8 j + i of the Welford
// accumulators i = (r & 3) + 4 h (torch_math.hpp) -- four accumulators per lane, the other four come over one swap.
__device__ __forceinline__ void row_moments_pair(const float (&v)[2][16], const int h, const float eps, float &mean, float &rstd) {
    Moments own[4], other[4];
#pragma unroll
    for (int il = 0; il < 4; ++il) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = v[j >> 2][4 * (j & 3) + il];
        own[il] = welford8(x);
        other[il].m1 = __shfl_xor(own[il].m1, 32);
        other[il].m2 = __shfl_xor(own[il].m2, 32);
    }
    Moments all[8];
#pragma unroll
    for (int il = 0; il < 4; ++il) {
        all[il].m1 = h ? other[il].m1 : own[il].m1;
        all[il].m2 = h ? other[il].m2 : own[il].m2;
        all[4 + il].m1 = h ? own[il].m1 : other[il].m1;
        all[4 + il].m2 = h ? own[il].m2 : other[il].m2;
    }
    merge8(all, eps, mean, rstd);
}

// ---- the update on 16-row tiles held in LDS, the feature axis split over four waves (rspmm_order_kernel, UPDATE == 3) ----
// The hand-off of the order kernel's form 3 (UPD2_*: the generator's HANDOFF2_*): walkers park finished aggregate rows in LDS tiles (16 rows x UPD2_ROW_FLOATS floats: 64 +
// 4 pad -- lane (row n, k-quarter q) reading element 4 s + q of row n hits 64 distinct banks); each of the four update waves
// keeps ITS 16 features of W in 32 registers (A operand of v_mfma_f32_16x16x4_f32: lane (i, q) holds W[16 u + i][4 s + q]) and
// multiplies every tile -- the products are k-ascending fmaf chains, the reference's nn.Linear order (the same arrangement as
// phase 3 of dense_order_layer.hip, pinned there bit for bit); the pre-norm tile meets in LDS, then every wave finishes four
// rows: LayerNorm in torch's operation order by a 16-lane group per row (as layer0_kernels.hpp), ReLU, residual, one
// coalesced 256-byte store per row.
constexpr int UPD2_NT = 4;                       // aggregate tiles the walkers fill round-robin (the generator's HANDOFF2_NT)
constexpr int UPD2_ROW_FLOATS = 68;              // 272 bytes (HANDOFF2_ROW_BYTES)
constexpr int UPD2_TILE_FLOATS = 16 * UPD2_ROW_FLOATS;
constexpr int UPD2_MAX_CHAIN_ROWS = 64;          // chain rows of a workgroup the control block can list
// overlay (in the ring's place once the chain is done): NT aggregate tiles (a ring of 64 rows, [64][UPD2_ROW_FLOATS]) | behind
// them the SAME 64 rows' x, same pitch, elements in NATURAL order (the walkers' stream_park writes both with one ds_write_b128 per
// lane each; an update lane reads element 4 s + kk of row n: 2-way bank conflicts, the minimum for 64 lanes).  A block's pre-norm
// rows are written IN PLACE of its x rows once all four update waves have read them (rspmm_order_kernels.hpp, form-3 block) --
// there is no separate pre-norm area: (NT + 4) tiles = NT aggregate tiles + NT x tiles
constexpr int UPD2_OVERLAY_BYTES = (UPD2_NT + 4) * UPD2_TILE_FLOATS * 4;
// control block (behind ring / overlay; bytes): 0 tail, 4 walkers done, 8 update-wave barrier (arrivals), 12 chain done, 16
// generations consumed, 20 chain rows listed, 32 posted[NT], 64 rowid[16 NT], 320 chain row offsets[64], 576 chain tile
constexpr int UPD2_CTL_CONSUMED = 4, UPD2_CTL_NCHAIN = 5, UPD2_CTL_RETRIES = 6 /* park waits that found the ring full (diagnostic) */,
              UPD2_CTL_ERR = 7 /* a park wait of the generated walk gave up (HANDOFF2_ERR_OFF) */,
              UPD2_CTL_POSTED = 8, UPD2_CTL_ROWID = 16,
              UPD2_CTL_CROW = 80;   // (word offsets)
constexpr int UPD2_CTL_CTILE_BYTES = 576, UPD2_CTL_BYTES = UPD2_CTL_CTILE_BYTES + UPD2_TILE_FLOATS * 4;

// LayerNorm (torch's operation order, torch_math.hpp) of a 64-feature row held by a 16-lane group, feature 4 l16 + e in y[e].
// `row`: the row's 64 floats in LDS (already there); `mom`: 16 floats of LDS private to the group.  LDS operations of one wave
// execute in order: no barrier between the group's writes and reads.
__device__ __forceinline__ void ln_row_group(float (&y)[4], const float *row, float *mom, const int l16, const float eps,
                                             const float (&gamma)[4], const float (&beta)[4]) {
    float xv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) xv[j] = row[8 * j + (l16 & 7)];
    const Moments w = welford8(xv);
    if (l16 < 8) {
        mom[2 * l16] = w.m1;
        mom[2 * l16 + 1] = w.m2;
    }
    Moments all[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) all[i] = Moments{mom[2 * i], mom[2 * i + 1]};
    float mean, rstd;
    merge8(all, eps, mean, rstd);
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = ln_apply(y[e], mean, rstd, gamma[e], beta[e]);
}

// lane I of this lane's row of 16 lanes (DPP row_share: no LDS round trip)
template <int I>
__device__ __forceinline__ float row_share(const float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + I, 0xf, 0xf, false));
}

__device__ __forceinline__ void gather_moments8(const Moments w, Moments (&all)[8]) {
    all[0] = Moments{row_share<0>(w.m1), row_share<0>(w.m2)};
    all[1] = Moments{row_share<1>(w.m1), row_share<1>(w.m2)};
    all[2] = Moments{row_share<2>(w.m1), row_share<2>(w.m2)};
    all[3] = Moments{row_share<3>(w.m1), row_share<3>(w.m2)};
    all[4] = Moments{row_share<4>(w.m1), row_share<4>(w.m2)};
    all[5] = Moments{row_share<5>(w.m1), row_share<5>(w.m2)};
    all[6] = Moments{row_share<6>(w.m1), row_share<6>(w.m2)};
    all[7] = Moments{row_share<7>(w.m1), row_share<7>(w.m2)};
}

// two rows at once (rows a and b of one 16-lane group), the eight partial moments of a row exchanged between lanes in registers:
// one LDS round trip for the pair -- an update wave's time is LDS round trips
template <class RowPtr>      // (an LDS-typed pointer: DS instructions, not FLAT ones)
__device__ __forceinline__ void ln_row_group2(float (&ya)[4], float (&yb)[4], const RowPtr row_a, const RowPtr row_b, const int l16,
                                              const float eps, const float (&gamma)[4], const float (&beta)[4]) {
    float xa[8], xb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) xa[j] = row_a[8 * j + (l16 & 7)];
#pragma unroll
    for (int j = 0; j < 8; ++j) xb[j] = row_b[8 * j + (l16 & 7)];
    float mean, rstd;
    Moments all[8];
    gather_moments8(welford8(xa), all);
    merge8(all, eps, mean, rstd);
#pragma unroll
    for (int e = 0; e < 4; ++e) ya[e] = ln_apply(ya[e], mean, rstd, gamma[e], beta[e]);
    gather_moments8(welford8(xb), all);
    merge8(all, eps, mean, rstd);
#pragma unroll
    for (int e = 0; e < 4; ++e) yb[e] = ln_apply(yb[e], mean, rstd, gamma[e], beta[e]);
}

// ---- the tile as a unit (the rspmm tail; 8 weight registers instead of 16: it runs under a 128-register cap) ----
constexpr int UPDATE_LDS_FLOATS = 32 * 64 * 4 + 3 * 64;   // weight image + {bias, LayerNorm weight, LayerNorm bias}

// Weight image [hs (32)][lane (64)] float4: one 16-byte read per lane feeds the four matrix instructions of half-step hs
// (k = 4 hs .. 4 hs + 3): {tile 0, pair s = 2 hs; tile 1, same pair; tile 0, pair s + 1; tile 1, pair s + 1}, element
// 2 s + (lane >> 5) of the pair, feature 32 m + (lane & 31).
__device__ __forceinline__ void update_stage_weights(float *lds_w, const float *weight, const float *bias, const float *ln_w,
                                                     const float *ln_b, const int flags, const int tid, const int nthreads) {
    for (int idx4 = tid; idx4 < 32 * 64; idx4 += nthreads) {
        const int l = idx4 & 63, hs = idx4 >> 6;
        const float *wr = weight + (l & 31) * 128 + 4 * hs;
        const float4 m0 = *reinterpret_cast<const float4 *>(wr), m1 = *reinterpret_cast<const float4 *>(wr + 32 * 128);
        const bool odd = (l >> 5) != 0;
        reinterpret_cast<float4 *>(lds_w)[idx4] =
            make_float4(odd ? m0.y : m0.x, odd ? m1.y : m1.x, odd ? m0.w : m0.z, odd ? m1.w : m1.z);
    }
    float *lds_vec = lds_w + 32 * 64 * 4;
    if (tid < 64) {
        lds_vec[tid] = bias ? bias[tid] : 0.f;
        lds_vec[64 + tid] = (flags & CONV_LN) ? ln_w[tid] : 1.f;
        lds_vec[128 + tid] = (flags & CONV_LN) ? ln_b[tid] : 0.f;
    }
}

// b[0..7]: the eight 16-byte chunks 2 i + h of the lane's x row, b[8..15]: of its aggregate row (lane = (row j, half h)).
// Writes the 32 output features this lane ends up owning to out_row[32 m + 8 g + 4 h ..] when `valid`.
__device__ __forceinline__ void update_tile(float4 (&b)[16], const float *lds_w, const int lane, const int flags, const float eps,
                                            float *out_row, const bool valid) {
    const int h = lane >> 5;
    const float4 *w4 = reinterpret_cast<const float4 *>(lds_w);
    const float *lds_vec = lds_w + 32 * 64 * 4;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc0[r] = 0.f;
        acc1[r] = 0.f;
    }
    float4 wc = w4[lane];
    // all operand swaps BEFORE the chain (an instruction between two dependent matrix instructions delays the second one
    // far beyond its own issue time); in place -- the x chunks are swapped back afterwards for the residual
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        swap32(b[i].x, b[i].y);   // .x: k pair s = 4 i,     .y: s = 4 i + 2
        swap32(b[i].z, b[i].w);   // .z: k pair s = 4 i + 1, .w: s = 4 i + 3
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!(flags & CONV_DBG_NO_MATRIX))
#pragma unroll
    for (int hs = 0; hs < 32; ++hs) {
        // k ascending: half 0 of chunk i holds s = 4 i (.x), 4 i + 1 (.z); half 1 holds s = 4 i + 2 (.y), 4 i + 3 (.w)
        const float b0 = (hs & 1) ? b[hs >> 1].y : b[hs >> 1].x;
        const float b1 = (hs & 1) ? b[hs >> 1].w : b[hs >> 1].z;
        float4 wn = wc;
        if (hs + 1 < 32) wn = w4[(hs + 1) * 64 + lane];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.x, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.y, b0, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.z, b1, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.w, b1, acc1, 0, 0, 0);
        wc = wn;
        __builtin_amdgcn_sched_barrier(0);
    }
    if (flags & CONV_RESIDUAL) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            swap32(b[i].x, b[i].y);
            swap32(b[i].z, b[i].w);
        }
    }
    // ---- epilogue: bias (added after the chain, like addmm), LayerNorm in the reference's operation order
    // (torch_math.hpp), ReLU, residual ----
    float v[2][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        v[0][r] = acc0[r] + lds_vec[feat_of(0, r, h)];
        v[1][r] = acc1[r] + lds_vec[feat_of(1, r, h)];
    }
    if (flags & CONV_LN) {
        float mean, rstd;
        row_moments_pair(v, h, eps, mean, rstd);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = feat_of(m, r, h);
                v[m][r] = ln_apply(v[m][r], mean, rstd, lds_vec[64 + f], lds_vec[128 + f]);
            }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 y = make_float4(v[m][4 * g + 0], v[m][4 * g + 1], v[m][4 * g + 2], v[m][4 * g + 3]);
            if (flags & CONV_RELU) {
                y.x = fmaxf(y.x, 0.f);
                y.y = fmaxf(y.y, 0.f);
                y.z = fmaxf(y.z, 0.f);
                y.w = fmaxf(y.w, 0.f);
            }
            if (flags & CONV_RESIDUAL) {
                const float4 xi = b[4 * m + g];  // x[row][32 m + 8 g + 4 h ..]: the chunk this lane holds
                y.x += xi.x;
                y.y += xi.y;
                y.z += xi.z;
                y.w += xi.w;
            }
            if (valid) *reinterpret_cast<float4 *>(out_row + 32 * m + 8 * g + 4 * h) = y;
        }
}

}  // namespace ultra
