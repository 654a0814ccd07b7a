// Backward of the fused layer update (fine-tuning path, BASELINE.json config 5):
//
//     out = [x +] relu( LayerNorm( W . [x ; agg] + b ) )                       layers.py:233-240 (+ models.py:158-160)
//
// In the reference this is five autograd nodes (cat, addmm, native_layer_norm, relu, add) whose backward reads and
// writes (rows, 64) / (rows, 128) tensors a dozen times and ends in two skinny GEMMs with K = rows.  Here:
//
//   conv_update_bwd_rows_kernel     per 32-row tile of a wave: recompute z = W.[x;agg] + b on the matrix cores (nothing
//                                   but x and agg was saved by the forward), LayerNorm / ReLU backward in registers,
//                                   d[x;agg] = dz . W as a second MFMA product whose B operand IS the accumulator layout
//                                   of the first (the contraction index may be visited in any order), dz written once
//                                   for the weight kernel, d gamma / d beta summed per lane.
//   conv_update_bwd_weights_kernel  dW = dz^T . [x;agg] (64 x 128, K = rows) and db: persistent waves keep the whole
//                                   dW in 128 accumulator registers, operands arrive as 128-byte row segments.
//   conv_update_bwd_reduce_kernel   sums the per-workgroup partials in a fixed order (no atomics: run-to-run
//                                   deterministic gradients).
//
// Gradients are not part of the reference-order contract (the reference's own GPU backward is atomicAdd scatter): plain
// two-pass LayerNorm statistics, free summation order.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../include/ultra_nbfnet.h"
#include "../../include/ultra_rspmm.h"
#include "plan.hpp"
#include "device_scope.hpp"

namespace ultra {

using f32x16 = float __attribute__((ext_vector_type(16)));

enum { CB_LN = 1, CB_RELU = 2, CB_RESIDUAL = 4,
       // measurement switches of the fused kernel (results are wrong with any of them): tools/conv_bwd_probe.py PROBE_FLAGS
       CB_DBG_NO_W_MFMA = 256, CB_DBG_NO_DX_MFMA = 512, CB_DBG_NO_Z_MFMA = 1024,
       // workgroup 0 leaves its shader-clock ticks (s_memtime) and its 100-MHz ticks (s_memrealtime) in the first 16 bytes of the workspace
       CB_DBG_CLOCK = 2048 };
constexpr int CB_WR_STRIDE = 136;                 // row stride (floats) of the row-major weight copy: 4 rows apart = 32 banks apart
constexpr int CB_PART = 64 * 128 + 3 * 64;        // floats per workgroup partial: dW, db, d gamma, d beta

struct ConvBwdParams {
    const float *x, *agg, *gout;
    const float *weight, *bias, *ln_w, *ln_b;
    float *gx, *gagg;
    float *dz;         // (rows, 64) scratch
    float *part;       // (n_part, CB_PART) scratch
    float *gweight, *gbias, *gln_w, *gln_b;
    long long rows;
    int n_part;
    float eps;
    int flags;
};

__device__ __forceinline__ int cb_feat(int m, int r, int h) { return 32 * m + (r & 3) + 8 * (r >> 2) + 4 * h; }

__global__ void __launch_bounds__(512) conv_update_bwd_rows_kernel(const ConvBwdParams p) {
    // fragment copy for the recompute: [m][c][lane] float4 = W[32 m + (lane & 31)][8 c + 4 (lane >> 5) .. + 3]
    __shared__ __attribute__((aligned(16))) float lds_wf[2 * 16 * 64 * 4];
    // row-major copy for d[x;agg] = dz . W
    __shared__ float lds_wr[64 * CB_WR_STRIDE];
    __shared__ float lds_vec[3 * 64];
    const int tid = threadIdx.x;
    for (int idx4 = tid; idx4 < 2 * 16 * 64; idx4 += blockDim.x) {
        const int l = idx4 & 63, c = (idx4 >> 6) & 15, m = idx4 >> 10;
        reinterpret_cast<float4 *>(lds_wf)[idx4] =
            *reinterpret_cast<const float4 *>(p.weight + (32 * m + (l & 31)) * 128 + 8 * c + 4 * (l >> 5));
    }
    for (int idx = tid; idx < 64 * 128; idx += blockDim.x) lds_wr[(idx >> 7) * CB_WR_STRIDE + (idx & 127)] = p.weight[idx];
    if (tid < 64) {
        lds_vec[tid] = p.bias ? p.bias[tid] : 0.f;
        lds_vec[64 + tid] = (p.flags & CB_LN) ? p.ln_w[tid] : 1.f;
        lds_vec[128 + tid] = (p.flags & CB_LN) ? p.ln_b[tid] : 0.f;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const float4 *w4 = reinterpret_cast<const float4 *>(lds_wf);
    const long long ntile = (p.rows + 31) / 32;
    const int wpb = blockDim.x >> 6;
    const long long tstride = (long long)gridDim.x * wpb;
    float dgam[2][16], dbet[2][16];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) dgam[m][r] = 0.f, dbet[m][r] = 0.f;

    // (wave-major: workgroup b takes tiles b, b + grid, ... -- with fewer tiles than waves every workgroup still gets its share)
    for (long long tile = blockIdx.x + (long long)gridDim.x * wave; tile < ntile; tile += tstride) {
        const long long row = tile * 32 + j;
        const bool valid = row < p.rows;
        const long long rc = valid ? row : p.rows - 1;
        const float4 *xr = reinterpret_cast<const float4 *>(p.x + rc * 64);
        const float4 *ar = reinterpret_cast<const float4 *>(p.agg + rc * 64);
        const float4 *gr = reinterpret_cast<const float4 *>(p.gout + rc * 64);
        float4 bx[8], ba[8], g4[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) bx[i] = xr[2 * i + h];
#pragma unroll
        for (int i = 0; i < 8; ++i) ba[i] = ar[2 * i + h];
        // the incoming gradient in the accumulator layout: features 32 m + 8 g + 4 h .. + 3
#pragma unroll
        for (int i = 0; i < 8; ++i) g4[i] = gr[8 * (i >> 2) + 2 * (i & 3) + h];

        // ---- z = W . [x ; agg] (+ b) ----
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = 0.f, acc1[r] = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const float4 b = c < 8 ? bx[c] : ba[c - 8];
            const float4 a0 = w4[(0 * 16 + c) * 64 + lane];
            const float4 a1 = w4[(1 * 16 + c) * 64 + lane];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b.w, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);      // keeps the 32 fragment reads (128 VGPRs) from being hoisted to the top
        }
        float v[2][16], d[2][16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            v[0][r] = acc0[r] + lds_vec[cb_feat(0, r, h)];
            v[1][r] = acc1[r] + lds_vec[cb_feat(1, r, h)];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            d[i >> 2][4 * (i & 3) + 0] = g4[i].x;
            d[i >> 2][4 * (i & 3) + 1] = g4[i].y;
            d[i >> 2][4 * (i & 3) + 2] = g4[i].z;
            d[i >> 2][4 * (i & 3) + 3] = g4[i].w;
        }
        // ---- LayerNorm / ReLU backward: v becomes z^ (normalised), d becomes dz ----
        float rstd = 1.f;
        if (p.flags & CB_LN) {
            float s1 = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) s1 += v[m][r];
            s1 += __shfl_xor(s1, 32);
            const float mean = s1 * (1.f / 64.f);
            float s2 = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    v[m][r] -= mean;
                    s2 += v[m][r] * v[m][r];
                }
            s2 += __shfl_xor(s2, 32);
            rstd = 1.f / sqrtf(s2 * (1.f / 64.f) + p.eps);
        }
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = cb_feat(m, r, h);
                const float gam = lds_vec[64 + f], bet = lds_vec[128 + f];
                const float zh = (p.flags & CB_LN) ? v[m][r] * rstd : v[m][r];
                const float y = (p.flags & CB_LN) ? zh * gam + bet : zh;
                float dy = d[m][r];
                if ((p.flags & CB_RELU) && !(y > 0.f)) dy = 0.f;
                if (!valid) dy = 0.f;
                if (p.flags & CB_LN) {
                    dgam[m][r] += dy * zh;
                    dbet[m][r] += dy;
                    dy *= gam;              // d z^
                    m1 += dy;
                    m2 += dy * zh;
                }
                v[m][r] = zh;
                d[m][r] = dy;
            }
        if (p.flags & CB_LN) {
            m1 += __shfl_xor(m1, 32);
            m2 += __shfl_xor(m2, 32);
            m1 *= (1.f / 64.f);
            m2 *= (1.f / 64.f);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) d[m][r] = rstd * (d[m][r] - m1 - v[m][r] * m2);
        }
        // ---- dz to memory (operand of the weight kernel) ----
        if (valid) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4 *>(p.dz + row * 64 + 32 * m + 8 * g + 4 * h) =
                        make_float4(d[m][4 * g + 0], d[m][4 * g + 1], d[m][4 * g + 2], d[m][4 * g + 3]);
        }
        // ---- d[x ; agg][k][row] = sum_f W[f][k] dz[row][f]: contraction pairs {cb_feat(m, r, 0), cb_feat(m, r, 1)} ----
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x16 o0, o1;
#pragma unroll
            for (int r = 0; r < 16; ++r) o0[r] = 0.f, o1[r] = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float *wrow = lds_wr + cb_feat(m, r, h) * CB_WR_STRIDE + 64 * half + j;
                    o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[0], d[m][r], o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[32], d[m][r], o1, 0, 0, 0);
                    if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            float *dst = (half == 0 ? p.gx : p.gagg) + row * 64;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x16 &o = kt == 0 ? o0 : o1;
                    float4 y = make_float4(o[4 * g + 0], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]);
                    if (half == 0 && (p.flags & CB_RESIDUAL)) {
                        const float4 gi = g4[4 * kt + g];      // grad_out[row][32 kt + 8 g + 4 h ..]
                        y.x += gi.x;
                        y.y += gi.y;
                        y.z += gi.z;
                        y.w += gi.w;
                    }
                    if (valid) *reinterpret_cast<float4 *>(dst + 32 * kt + 8 * g + 4 * h) = y;
                }
        }
    }
    // ---- d gamma / d beta of this workgroup: sum over the 32 rows a lane half holds, then over the waves ----
    __syncthreads();                                   // weights are no longer needed: reuse their LDS
    float *red = lds_wf;                               // [wave][reg 0..31][lane] = 8 KB per wave (<= 4 waves at a time fit lds_wf)
    float *tot = lds_wr;                               // [wave][2][64]
    for (int which = 0; which < 2; ++which) {
        for (int w0 = 0; w0 < wpb; w0 += 4) {
            if (wave >= w0 && wave < w0 + 4) {
                float *mine = red + (wave - w0) * 32 * 64;
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mine[(16 * m + r) * 64 + lane] = which == 0 ? dgam[m][r] : dbet[m][r];
            }
            __syncthreads();
            if (wave >= w0 && wave < w0 + 4) {
                const float *mine = red + (wave - w0) * 32 * 64;
                const int reg = lane & 31, hh = lane >> 5;
                float s = 0.f;
                for (int q = 0; q < 32; ++q) s += mine[reg * 64 + hh * 32 + ((q + lane) & 31)];
                tot[(wave * 2 + which) * 64 + cb_feat(reg >> 4, reg & 15, hh)] = s;
            }
            __syncthreads();
        }
    }
    if (tid < 128) {
        const int which = tid >> 6, f = tid & 63;
        float s = 0.f;
        for (int w = 0; w < wpb; ++w) s += tot[(w * 2 + which) * 64 + f];
        p.part[(long long)blockIdx.x * CB_PART + 64 * 128 + 64 + tid] = s;     // [.. + 64]: d gamma, [.. + 128]: d beta
    }
}

// dW[f][k] = sum_row dz[row][f] X[row][k], X = [x ; agg]; db[f] = sum_row dz[row][f].
// A operand (M = f, K = row): lane (i, hh) supplies dz[row0 + 2 s + hh][32 m + i]; B operand (K = row, N = k):
// lane (n, hh) supplies X[row0 + 2 s + hh][32 kt + n] -- every load instruction fetches two 128-byte row segments.
__global__ void __launch_bounds__(512) conv_update_bwd_weights_kernel(const ConvBwdParams p) {
    __shared__ float lds_dw[64 * 128 + 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, hh = lane >> 5;
    const int wpb = blockDim.x >> 6;
    const long long ntile = (p.rows + 31) / 32;
    const long long tstride = (long long)gridDim.x * wpb;
    f32x16 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][kt][r] = 0.f;
    float dbias[2] = {0.f, 0.f};
    // operands of 4 contraction steps (8 rows) at a time; the next group is requested before the 32 MFMAs of the
    // current one, across tile boundaries too
    struct Group {
        float a[4][2], b[4][4];
    };
    const auto load_group = [&](Group &g, const long long tile, const int s4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long row = tile * 32 + 2 * (4 * s4 + u) + hh;
            const bool ok = row < p.rows;
            const long long rc = ok ? row : p.rows - 1;
            const float *dzr = p.dz + rc * 64 + n, *xr = p.x + rc * 64 + n, *ar = p.agg + rc * 64 + n;
            g.a[u][0] = ok ? dzr[0] : 0.f;
            g.a[u][1] = ok ? dzr[32] : 0.f;
            g.b[u][0] = xr[0];
            g.b[u][1] = xr[32];
            g.b[u][2] = ar[0];
            g.b[u][3] = ar[32];
        }
    };
    long long tile = blockIdx.x + (long long)gridDim.x * wave;
    Group cur;
    if (tile < ntile) load_group(cur, tile, 0);
    for (; tile < ntile; tile += tstride) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            Group nxt;
            {
                const bool last = s4 == 3;
                const long long tn = last ? tile + tstride : tile;
                load_group(nxt, tn < ntile ? tn : tile, last ? 0 : s4 + 1);      // (past the end: a harmless reload)
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                dbias[0] += cur.a[u][0];
                dbias[1] += cur.a[u][1];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
                        acc[m][kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[u][m], cur.b[u][kt], acc[m][kt], 0, 0, 0);
            }
            cur = nxt;
        }
    }
    dbias[0] += __shfl_xor(dbias[0], 32);
    dbias[1] += __shfl_xor(dbias[1], 32);
    // waves fold their accumulators into LDS one after the other (fixed order)
    for (int w = 0; w < wpb; ++w) {
        if (wave == w) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int idx = cb_feat(m, r, hh) * 128 + 32 * kt + n;
                        lds_dw[idx] = w == 0 ? acc[m][kt][r] : lds_dw[idx] + acc[m][kt][r];
                    }
            if (hh == 0) {
                lds_dw[64 * 128 + n] = w == 0 ? dbias[0] : lds_dw[64 * 128 + n] + dbias[0];
                lds_dw[64 * 128 + 32 + n] = w == 0 ? dbias[1] : lds_dw[64 * 128 + 32 + n] + dbias[1];
            }
        }
        __syncthreads();
    }
    float *dst = p.part + (long long)blockIdx.x * CB_PART;
    for (int idx = tid; idx < 64 * 128 + 64; idx += blockDim.x) dst[idx] = lds_dw[idx];
}

// ---- rows + weights in ONE launch (round 6): dz never goes through memory ----
// Waves 0 - 3 of a workgroup run the rows kernel's tile loop, waves 4 - 7 the weight kernel's; wave r and wave r + 4 share a SIMD and
// a tile sequence.  The rows wave parks each tile's dz (32 x 64 floats) in one of its pair's two LDS buffers instead of storing it;
// its partner reads the A operands of dW = dz^T [x ; agg] from there and the B operands ([x ; agg] as 128-byte row segments) from
// L2, where the rows wave's own loads have just put them.  HBM traffic per call: x, agg, grad_out in, gx, gagg out -- 1.26 GB at
// 985 k rows where the two launches moved 2.27 GB (dz out and in again, x and agg in again); the SIMD's matrix pipe sees the same
// 384 instructions a tile, now from two waves that fill each other's gaps.  Hand-off: one flag per buffer (0 = free, 1 = full),
// release / acquire at workgroup scope; strictly alternating, so neither side can run ahead by more than two tiles.
// LDS layout of a parked tile: row-major [32][64] with the two 32-column halves of ODD rows swapped (column ^ 32): the weight wave's
// A-operand reads -- lane (n, hh) takes dz[2 q + hh][n] and [32 + n] -- then touch 64 different banks.
constexpr int CBF_TILE = 32 * 64;                  // floats per parked dz tile

__global__ void __launch_bounds__(512) conv_update_bwd_fused_kernel(const ConvBwdParams p) {
    __shared__ __attribute__((aligned(16))) float lds_wf[2 * 16 * 64 * 4];
    __shared__ float lds_wr[64 * CB_WR_STRIDE];
    __shared__ float lds_vec[3 * 64];
    __shared__ __attribute__((aligned(16))) float lds_dz[4 * 2 * CBF_TILE];      // [pair][buffer][tile]; the weight waves' fold afterwards
    __shared__ int lds_flag[4 * 2];
    const int tid = threadIdx.x;
    const unsigned long long clk0 = __builtin_readcyclecounter(), real0 = __builtin_amdgcn_s_memrealtime();
    for (int idx4 = tid; idx4 < 2 * 16 * 64; idx4 += blockDim.x) {
        const int l = idx4 & 63, c = (idx4 >> 6) & 15, m = idx4 >> 10;
        reinterpret_cast<float4 *>(lds_wf)[idx4] =
            *reinterpret_cast<const float4 *>(p.weight + (32 * m + (l & 31)) * 128 + 8 * c + 4 * (l >> 5));
    }
    for (int idx = tid; idx < 64 * 128; idx += blockDim.x) lds_wr[(idx >> 7) * CB_WR_STRIDE + (idx & 127)] = p.weight[idx];
    if (tid < 64) {
        lds_vec[tid] = p.bias ? p.bias[tid] : 0.f;
        lds_vec[64 + tid] = (p.flags & CB_LN) ? p.ln_w[tid] : 1.f;
        lds_vec[128 + tid] = (p.flags & CB_LN) ? p.ln_b[tid] : 0.f;
    }
    if (tid < 8) lds_flag[tid] = 0;
    __syncthreads();
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool rows_role = wave < 4;
    const int pair = wave & 3;
    const long long ntile = (p.rows + 31) / 32;
    const long long tstride = (long long)gridDim.x * 4;
    float *const my_dz = lds_dz + pair * 2 * CBF_TILE;
    int *const my_flag = lds_flag + pair * 2;
    // (each role's persistent registers -- d gamma / d beta partials here, the 128 accumulators of dW there -- live in its own branch
    // only; both branches pass the same five workgroup barriers)
    float *const lds_dw = lds_dz;
    if (rows_role) {
        float dgam[2][16], dbet[2][16];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) dgam[m][r] = 0.f, dbet[m][r] = 0.f;
        const int j = lane & 31, h = lane >> 5;
        const float4 *w4 = reinterpret_cast<const float4 *>(lds_wf);
        int buf = 0;
        // One rows wave per SIMD has nobody to hide its loads behind (the two-launch form ran two): the NEXT tile's x and agg rows are
        // requested once this tile's LayerNorm backward is through (their registers are free after the recompute, and the
        // LayerNorm phase is the register peak) and arrive under the 128 matrix instructions of d[x ; agg]; this tile's grad_out
        // rows are requested at the top and arrive under the recompute's 128.
        float4 bx[8], ba[8], g4[8];
        const auto row_of = [&](const long long tile) {
            const long long row = tile * 32 + j;
            return row < p.rows ? row : p.rows - 1;
        };
        const auto load_xa = [&](const long long tile, float4 (&tx)[8], float4 (&ta)[8]) {
            const float4 *xr = reinterpret_cast<const float4 *>(p.x + row_of(tile) * 64);
            const float4 *ar = reinterpret_cast<const float4 *>(p.agg + row_of(tile) * 64);
#pragma unroll
            for (int i = 0; i < 8; ++i) tx[i] = xr[2 * i + h];
#pragma unroll
            for (int i = 0; i < 8; ++i) ta[i] = ar[2 * i + h];
        };
        const auto load_g = [&](const long long tile, float4 (&tg)[8]) {
            // the incoming gradient in the accumulator layout: features 32 m + 8 g + 4 h .. + 3
            const float4 *gr = reinterpret_cast<const float4 *>(p.gout + row_of(tile) * 64);
#pragma unroll
            for (int i = 0; i < 8; ++i) tg[i] = gr[8 * (i >> 2) + 2 * (i & 3) + h];
        };
        long long tile = blockIdx.x + (long long)gridDim.x * pair;
        if (tile < ntile) load_xa(tile, bx, ba);
        for (; tile < ntile; tile += tstride, buf ^= 1) {
            const long long row = tile * 32 + j;
            const bool valid = row < p.rows;
            const long long tnext = tile + tstride < ntile ? tile + tstride : tile;      // (past the end: a harmless reload)
            load_g(tile, g4);      // (needed after the recompute: arrives under its 128 matrix instructions)
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[r] = 0.f, acc1[r] = 0.f;
            if (p.flags & CB_DBG_NO_Z_MFMA) {      // (timing-only build switch: wrong results)
#pragma unroll
                for (int c = 0; c < 8; ++c) acc0[c] = bx[c].x + ba[c].y, acc1[c] = bx[c].z + ba[c].w;
            } else
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const float4 b = c < 8 ? bx[c] : ba[c - 8];
                const float4 a0 = w4[(0 * 16 + c) * 64 + lane];
                const float4 a1 = w4[(1 * 16 + c) * 64 + lane];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b.x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b.y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b.z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b.w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b.w, acc1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            float v[2][16], d[2][16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                v[0][r] = acc0[r] + lds_vec[cb_feat(0, r, h)];
                v[1][r] = acc1[r] + lds_vec[cb_feat(1, r, h)];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                d[i >> 2][4 * (i & 3) + 0] = g4[i].x;
                d[i >> 2][4 * (i & 3) + 1] = g4[i].y;
                d[i >> 2][4 * (i & 3) + 2] = g4[i].z;
                d[i >> 2][4 * (i & 3) + 3] = g4[i].w;
            }
            float rstd = 1.f;
            if (p.flags & CB_LN) {
                float s1 = 0.f;
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s1 += v[m][r];
                s1 += __shfl_xor(s1, 32);
                const float mean = s1 * (1.f / 64.f);
                float s2 = 0.f;
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        v[m][r] -= mean;
                        s2 += v[m][r] * v[m][r];
                    }
                s2 += __shfl_xor(s2, 32);
                rstd = 1.f / sqrtf(s2 * (1.f / 64.f) + p.eps);
            }
            float m1 = 0.f, m2 = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = cb_feat(m, r, h);
                    const float gam = lds_vec[64 + f], bet = lds_vec[128 + f];
                    const float zh = (p.flags & CB_LN) ? v[m][r] * rstd : v[m][r];
                    const float y = (p.flags & CB_LN) ? zh * gam + bet : zh;
                    float dy = d[m][r];
                    if ((p.flags & CB_RELU) && !(y > 0.f)) dy = 0.f;
                    if (!valid) dy = 0.f;
                    if (p.flags & CB_LN) {
                        dgam[m][r] += dy * zh;
                        dbet[m][r] += dy;
                        dy *= gam;
                        m1 += dy;
                        m2 += dy * zh;
                    }
                    v[m][r] = zh;
                    d[m][r] = dy;
                }
            if (p.flags & CB_LN) {
                m1 += __shfl_xor(m1, 32);
                m2 += __shfl_xor(m2, 32);
                m1 *= (1.f / 64.f);
                m2 *= (1.f / 64.f);
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) d[m][r] = rstd * (d[m][r] - m1 - v[m][r] * m2);
            }
            // (this tile's x and agg rows were consumed by the recompute: their registers take the next tile's, which arrive under the
            // 128 matrix instructions of d[x ; agg] below)
            __builtin_amdgcn_sched_barrier(0);
            load_xa(tnext, bx, ba);
            __builtin_amdgcn_sched_barrier(0);
            // ---- dz to the pair's buffer (an invalid row parks zeros: dy was zeroed above) ----
            while (__hip_atomic_load(my_flag + buf, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) __builtin_amdgcn_s_sleep(1);
            {
                float *dst = my_dz + buf * CBF_TILE + j * 64;
                const int swap = 32 * (j & 1);
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<float4 *>(dst + ((32 * m + 8 * g + 4 * h) ^ swap)) =
                            make_float4(d[m][4 * g + 0], d[m][4 * g + 1], d[m][4 * g + 2], d[m][4 * g + 3]);
            }
            if (lane == 0) __hip_atomic_store(my_flag + buf, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            // ---- d[x ; agg] = dz . W ----
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                // (the residual's share of gx -- grad_out itself -- is the chain's starting value: accumulator element 4 g + c of
                // tile kt is feature 32 kt + 8 g + 4 h + c of this lane's row, the layout grad_out was loaded in; its registers are
                // then free for the whole product)
                f32x16 o0, o1;
                const bool res = half == 0 && (p.flags & CB_RESIDUAL);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    o0[4 * g + 0] = res ? g4[g].x : 0.f, o0[4 * g + 1] = res ? g4[g].y : 0.f;
                    o0[4 * g + 2] = res ? g4[g].z : 0.f, o0[4 * g + 3] = res ? g4[g].w : 0.f;
                    o1[4 * g + 0] = res ? g4[4 + g].x : 0.f, o1[4 * g + 1] = res ? g4[4 + g].y : 0.f;
                    o1[4 * g + 2] = res ? g4[4 + g].z : 0.f, o1[4 * g + 3] = res ? g4[4 + g].w : 0.f;
                }
                if (!(p.flags & CB_DBG_NO_DX_MFMA))      // (timing-only build switch: wrong results)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float *wrow = lds_wr + cb_feat(m, r, h) * CB_WR_STRIDE + 64 * half + j;
                        o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[0], d[m][r], o0, 0, 0, 0);
                        o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[32], d[m][r], o1, 0, 0, 0);
                        if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                    }
                float *dst = (half == 0 ? p.gx : p.gagg) + row * 64;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x16 &o = kt == 0 ? o0 : o1;
                        if (valid)
                            *reinterpret_cast<float4 *>(dst + 32 * kt + 8 * g + 4 * h) =
                                make_float4(o[4 * g + 0], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]);
                    }
            }
        }
        __syncthreads();      // (1) every tile done: the weight copies and the dz buffers are free
        // ---- d gamma / d beta: each rows wave through its 8 KB of lds_wf, then over the waves ----
        {
            float *mine = lds_wf + pair * 32 * 64;
            for (int which = 0; which < 2; ++which) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mine[(16 * m + r) * 64 + lane] = which == 0 ? dgam[m][r] : dbet[m][r];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                const int reg = lane & 31, hq = lane >> 5;
                float sum = 0.f;
                for (int q = 0; q < 32; ++q) sum += mine[reg * 64 + hq * 32 + ((q + lane) & 31)];
                lds_wr[(pair * 2 + which) * 64 + cb_feat(reg >> 4, reg & 15, hq)] = sum;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
            }
        }
        for (int w = 0; w < 4; ++w) __syncthreads();      // (2 - 5) the weight waves' fold
    } else {
        f32x16 acc[2][4];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][kt][r] = 0.f;
        float dbias[2] = {0.f, 0.f};
        const int n = lane & 31, hh = lane >> 5;
        // B operands ([x ; agg] as 128-byte row segments) of four contraction steps (8 rows) at a time; the next group -- across
        // tile boundaries too -- is requested before the 32 matrix instructions of the current one
        struct Group {
            float b[4][4];
        };
        const auto load_group = [&](Group &g, const long long tile, const int s4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long row = tile * 32 + 2 * (4 * s4 + u) + hh;
                const long long rc = row < p.rows ? row : p.rows - 1;      // (past the end: dz there is zero)
                const float *xr = p.x + rc * 64 + n, *ar = p.agg + rc * 64 + n;
                g.b[u][0] = xr[0], g.b[u][1] = xr[32], g.b[u][2] = ar[0], g.b[u][3] = ar[32];
            }
        };
        // (two groups ahead: a group's 32 matrix instructions take ~ 2 k cycles, a load under the kernel's own traffic more)
        const auto group_at = [&](Group &g, const long long tile, const int s4) {      // group s4 >= 4: of the workgroup's next tile
            const long long tn = s4 < 4 ? tile : tile + tstride;
            load_group(g, tn < ntile ? tn : tile, s4 & 3);                             // (past the end: a harmless reload)
        };
        long long tile = blockIdx.x + (long long)gridDim.x * pair;
        Group cur, nx1;
        if (tile < ntile) {
            load_group(cur, tile, 0);
            load_group(nx1, tile, 1);
        }
        int buf = 0;
        for (; tile < ntile; tile += tstride, buf ^= 1) {
            const float *dzt = my_dz + buf * CBF_TILE;
            while (__hip_atomic_load(my_flag + buf, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != 1) __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                Group nx2;
                group_at(nx2, tile, s4 + 2);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int rt = 2 * (4 * s4 + u) + hh;          // row inside the tile; odd rows hold their halves swapped
                    const float a0 = dzt[rt * 64 + (n ^ (32 * hh))], a1 = dzt[rt * 64 + ((32 + n) ^ (32 * hh))];
                    dbias[0] += a0;
                    dbias[1] += a1;
                    if (p.flags & CB_DBG_NO_W_MFMA) {      // (timing-only build switch: wrong results)
                        acc[0][0][0] += a0 * cur.b[u][0] + a1 * cur.b[u][1] + cur.b[u][2] + cur.b[u][3];
                        continue;
                    }
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt) {
                        acc[0][kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, cur.b[u][kt], acc[0][kt], 0, 0, 0);
                        acc[1][kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, cur.b[u][kt], acc[1][kt], 0, 0, 0);
                    }
                }
                cur = nx1;
                nx1 = nx2;
            }
            // (LDS operations of a wave execute in order: the reads above are done when this store is)
            if (lane == 0) __hip_atomic_store(my_flag + buf, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        dbias[0] += __shfl_xor(dbias[0], 32);
        dbias[1] += __shfl_xor(dbias[1], 32);
        __syncthreads();      // (1)
        // the weight waves fold their accumulators into the (now free) dz buffers one after the other: a fixed order
        for (int w = 0; w < 4; ++w) {
            if (pair == w) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int idx = cb_feat(m, r, hh) * 128 + 32 * kt + n;
                            lds_dw[idx] = w == 0 ? acc[m][kt][r] : lds_dw[idx] + acc[m][kt][r];
                        }
                if (hh == 0) {
                    lds_dw[64 * 128 + n] = w == 0 ? dbias[0] : lds_dw[64 * 128 + n] + dbias[0];
                    lds_dw[64 * 128 + 32 + n] = w == 0 ? dbias[1] : lds_dw[64 * 128 + 32 + n] + dbias[1];
                }
            }
            __syncthreads();      // (2 - 5)
        }
    }
    float *dst = p.part + (long long)blockIdx.x * CB_PART;
    for (int idx = tid; idx < 64 * 128 + 64; idx += blockDim.x) dst[idx] = lds_dw[idx];
    if (tid < 128) {
        const int which = tid >> 6, f = tid & 63;
        float s = 0.f;
        for (int w = 0; w < 4; ++w) s += lds_wr[(w * 2 + which) * 64 + f];
        dst[64 * 128 + 64 + tid] = s;
    }
    if ((p.flags & CB_DBG_CLOCK) && blockIdx.x == 0 && tid == 0) {
        unsigned long long *out = reinterpret_cast<unsigned long long *>(p.dz);
        out[0] = __builtin_readcyclecounter() - clk0;
        out[1] = __builtin_amdgcn_s_memrealtime() - real0;
    }
}

// 64 consecutive entries per workgroup; the partials are split over sixteen thread groups (q, q + 16, q + 32, ... in
// ascending order each, four loads in flight), folded 0 + 1 + ... + 15 through LDS: a fixed order, reproducible run to run.
__global__ void __launch_bounds__(1024) conv_update_bwd_reduce_kernel(const ConvBwdParams p) {
    __shared__ float lds_q[16][64];
    const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + e;        // CB_PART is a multiple of 64
    float s = 0.f;
    int k = q;
    for (; k + 48 < p.n_part; k += 64) {
        const float a = p.part[(long long)k * CB_PART + idx], b = p.part[(long long)(k + 16) * CB_PART + idx];
        const float c = p.part[(long long)(k + 32) * CB_PART + idx], d = p.part[(long long)(k + 48) * CB_PART + idx];
        s = (((s + a) + b) + c) + d;
    }
    for (; k < p.n_part; k += 16) s += p.part[(long long)k * CB_PART + idx];
    lds_q[q][e] = s;
    __syncthreads();
    if (q != 0) return;
    s = lds_q[0][e];
#pragma unroll
    for (int g = 1; g < 16; ++g) s += lds_q[g][e];
    if (idx < 64 * 128)
        p.gweight[idx] = s;
    else if (idx < 64 * 128 + 64) {
        if (p.gbias) p.gbias[idx - 64 * 128] = s;
    } else if (idx < 64 * 128 + 128) {
        if (p.gln_w) p.gln_w[idx - 64 * 128 - 64] = s;
    } else if (p.gln_b)
        p.gln_b[idx - 64 * 128 - 128] = s;
}

// Launch shape of the two persistent kernels: `blocks` workgroups (= the number of partial-sum rows both write), wpb_rows /
// wpb_weights waves in each.  One workgroup per CU at most (the kernels hold 232 / 218 VGPRs: built for 128 they spill -- the
// FB15k237-shape step 6.1 -> 9.3 ms, profiles/r5_experiments.txt), and as many waves as it has 32-row tiles, up to 8.
// Few rows (the relation model's 3,792, the last layer's 2,056 listed rows: one tile a wave at most) are latency, not
// throughput: round 5 packed them into ceil(tiles / 8) workgroups of 8 waves -- 15 workgroups, 29 + 40 us a call whatever the
// row count (profiles/r6_01_timeline_eager.txt), of which the weight kernel's eight-deep serial fold through LDS and the
// rows kernel's staging by few workgroups are most.  Now: one tile per workgroup while the CUs last; the weight kernel runs ONE
// wave there (no fold), the rows kernel keeps four (three of them only help staging the weight matrix into LDS).
// ULTRA_CONV_BWD_SHAPE=blocks_cap,wpb_rows,wpb_weights overrides (0: the rule above; measurements).
struct BwdShape {
    int blocks, wpb_rows, wpb_weights;
};
static BwdShape bwd_shape(long long rows) {
    static int cu = 0, env[3] = {0, 0, 0};   // queried once (kept out of hipGraph capture)
    if (cu == 0) {
        int dev = 0, v = 0;
        cu = 256;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            cu = v;
        if (const char *e = getenv("ULTRA_CONV_BWD_SHAPE")) (void)sscanf(e, "%d,%d,%d", &env[0], &env[1], &env[2]);
    }
    const long long ntile = (rows + 31) / 32;
    const long long cap = env[0] > 0 ? env[0] : cu;
    long long blocks = ntile < cap ? ntile : cap;
    if (blocks < 1) blocks = 1;
    long long per = (ntile + blocks - 1) / blocks;      // tiles per workgroup
    if (per < 1) per = 1;
    BwdShape sh;
    sh.blocks = (int)blocks;
    // (measured, profiles/r6_experiments.txt: four waves of the weight kernel beside eight of the rows kernel -- 102.5 us per call at
    // 116 k rows against 119.4 with eight and eight, 575 against 584 at 985 k: half the serial fold, the loads still covered)
    sh.wpb_weights = env[2] > 0 ? env[2] : (int)(per > 4 ? 4 : per);
    sh.wpb_rows = env[1] > 0 ? env[1] : (int)(per > 8 ? 8 : (per < 4 ? 4 : per));
    return sh;
}

}  // namespace ultra

using namespace ultra;

extern "C" {

int64_t ultra_conv_update_backward_workspace(int64_t rows) {
    if (rows < 0) return 0;
    return rows * 64 * (int64_t)sizeof(float) + (int64_t)bwd_shape(rows).blocks * CB_PART * (int64_t)sizeof(float);
}

int32_t ultra_conv_update_backward(const void *x, const void *agg, const void *grad_out, const void *weight, const void *bias,
                                   const void *ln_weight, const void *ln_bias, void *grad_x, void *grad_agg, void *grad_weight,
                                   void *grad_bias, void *grad_ln_weight, void *grad_ln_bias, void *workspace,
                                   int64_t workspace_bytes, int64_t rows, int32_t input_dim, int32_t output_dim, float eps,
                                   int32_t flags, void *stream) {
    ULTRA_DEVICE_SCOPE(stream, x);
    if (input_dim != 64 || output_dim != 64) {
        set_error("ultra_conv_update_backward: only input_dim = output_dim = 64 is built (the ULTRA checkpoints' shape)");
        return ULTRA_ERR_UNSUPPORTED;
    }
    if (!x || !agg || !grad_out || !weight || !grad_x || !grad_agg || !grad_weight || !workspace || rows <= 0 ||
        ((flags & CB_LN) && (!ln_weight || !ln_bias))) {
        set_error("ultra_conv_update_backward: NULL operand (or rows <= 0)");
        return ULTRA_ERR_INVALID;
    }
    if (workspace_bytes < ultra_conv_update_backward_workspace(rows)) {
        set_error("ultra_conv_update_backward: workspace smaller than ultra_conv_update_backward_workspace(rows)");
        return ULTRA_ERR_INVALID;
    }
    ConvBwdParams p;
    p.x = (const float *)x;
    p.agg = (const float *)agg;
    p.gout = (const float *)grad_out;
    p.weight = (const float *)weight;
    p.bias = (const float *)bias;
    p.ln_w = (const float *)ln_weight;
    p.ln_b = (const float *)ln_bias;
    p.gx = (float *)grad_x;
    p.gagg = (float *)grad_agg;
    p.dz = (float *)workspace;
    p.part = p.dz + rows * 64;
    p.gweight = (float *)grad_weight;
    p.gbias = (float *)grad_bias;
    p.gln_w = (flags & CB_LN) ? (float *)grad_ln_weight : nullptr;
    p.gln_b = (flags & CB_LN) ? (float *)grad_ln_bias : nullptr;
    p.rows = rows;
    const BwdShape shape = bwd_shape(rows);
    p.n_part = shape.blocks;
    p.eps = eps;
    p.flags = flags;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    // ULTRA_CONV_BWD_FUSED=0: rows kernel and weights kernel as two launches (round 5's form, dz through memory)
    static const bool fused = [] {
        const char *e = getenv("ULTRA_CONV_BWD_FUSED");
        return !(e && e[0] == '0');
    }();
    if (fused) {
        hipLaunchKernelGGL(conv_update_bwd_fused_kernel, dim3(p.n_part), dim3(512), 0, s, p);
    } else {
        hipLaunchKernelGGL(conv_update_bwd_rows_kernel, dim3(p.n_part), dim3(64 * shape.wpb_rows), 0, s, p);
        hipLaunchKernelGGL(conv_update_bwd_weights_kernel, dim3(p.n_part), dim3(64 * shape.wpb_weights), 0, s, p);
    }
    hipLaunchKernelGGL(conv_update_bwd_reduce_kernel, dim3(CB_PART / 64), dim3(1024), 0, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error(std::string("conv_update backward launch: ") + hipGetErrorString(e));
        return ULTRA_ERR_HIP;
    }
    return ULTRA_OK;
}

}  // extern "C"
