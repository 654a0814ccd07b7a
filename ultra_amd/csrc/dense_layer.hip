// One whole GeneralizedRelationalConv layer on a dense-format plan in a single launch:
//
//     agg = sum_t rel[b, t] * (A_t . x[b]) + boundary                 rspmm add_mul + layers.py:199-200
//     out = [x +] relu( LayerNorm( W . cat[x, agg] + bias ) )         layers.py:233-240, models.py:158-160
//
// This is the steady-state layer of ULTRA's RelNBFNet (models.py:72-80): a few hundred relation nodes, 4 edge types,
// a (nearly) complete graph, hidden dim 64.  The separate kernels (rspmm_dense.hip + conv_update) spend as long on
// launch, prologue and the 1 MB aggregate round trip as on arithmetic at this size, so the layer is fused per
// 16-row tile: a workgroup (8 waves) owns 16 output nodes of one sample and all 64 features.
//   phase 1  v_mfma_f32_16x16x4_f32 over the byte-packed adjacency: wave (kq, ch) takes k-quarter kq and the column
//            tiles {2 ch, 2 ch + 1}, one accumulator per (column tile, type); B operand (x) straight from L2.
//   phase 2  per-type scaling by rel (a lane owns one column), k-quarters added in order through LDS, boundary added
//            -> the aggregate tile (16 x 64) sits in LDS, never in HBM.
//   phase 3  waves 0..3: transposed update product D[feature][row] (one feature tile each, 32 matrix instructions),
//            LayerNorm statistics exchanged through LDS, ReLU, residual, 16-byte stores.
// Deterministic, no atomics; products of phase 1 are exact (integer multiplicities), see rspmm_dense.hip.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "plan.hpp"

namespace ultra {

using f32x4 = float __attribute__((ext_vector_type(4)));

enum { DL_LN = 1, DL_RELU = 2, DL_RESIDUAL = 4 };

struct DenseLayerParams {
    const uint32_t *a16;      // [n_rt16][n_chunk][64 lanes][4 steps] words of 4 type bytes
    const float *rel, *x, *bnd;
    const long long *bnd_rows;   // point boundary (bnd = one row per outer slice) or NULL (bnd = full tensor or NULL)
    const float *weight, *bias, *ln_w, *ln_b;
    float *out;
    long long rel_so, rel_sr, x_so, x_sr, bnd_so, bnd_sr, out_so, out_sr;
    int n_out, n_in, n_rel, n_chunk, n_rt16, has_bnd, flags;
    float eps;
};

constexpr int DL_ROW_STRIDE = 68;   // floats per LDS tile row: 16-lane column reads of 4 consecutive floats hit 64 banks

__global__ void __launch_bounds__(512) dense_layer_kernel(const DenseLayerParams p) {
    __shared__ __attribute__((aligned(16))) float red[4 * 4 * 4 * 64];        // [kq][column tile][reg][lane]
    __shared__ __attribute__((aligned(16))) float x_lds[16 * DL_ROW_STRIDE];   // this tile's own rows of x
    __shared__ __attribute__((aligned(16))) float agg_lds[16 * DL_ROW_STRIDE];
    __shared__ float ln_part[2][4][16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = wave & 3, ch = wave >> 2;
    const int i16 = lane & 15, kk = lane >> 4;
    const int rt = blockIdx.x % p.n_rt16, outer = blockIdx.x / p.n_rt16;
    const float *xo = p.x + (long long)outer * p.x_so;
    const int row0 = rt * 16;

    // ---- first stage of phase 1 requested before anything else ----
    const int cpq = p.n_chunk / 4;                       // chunks (16 source rows each) per k-quarter
    const int c_begin = kq * cpq, c_end = c_begin + cpq;
    const char *ap = reinterpret_cast<const char *>(p.a16 + ((size_t)rt * p.n_chunk * 64 + lane) * 4);
    const char *xb = reinterpret_cast<const char *>(xo + 32 * ch + i16);   // B operand: lane (kk, j) holds x[4 s + kk][col + j]
    const uint32_t x_row_bytes = (uint32_t)p.x_sr * 4u;
    // one stage = 2 chunks = 32 source rows = 64 matrix instructions; the next stage's loads are all issued before
    // the current stage's matrix work and only waited for after it (two waves share a SIMD: >= 2 us of cover)
    struct Stage {
        uint4 a[2];
        float x[8][2];
    };
    const auto fetch = [&](int chunk, Stage &st) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            st.a[h] = *reinterpret_cast<const uint4 *>(ap + (uint32_t)(chunk + h) * (uint32_t)(64 * 16));
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int k = 16 * (chunk + h) + 4 * s + kk;   // rows past n_in: read a valid row, zeroed where it is consumed
                const char *xr = xb + (uint32_t)min(k, p.n_in - 1) * x_row_bytes;
                st.x[4 * h + s][0] = *reinterpret_cast<const float *>(xr);
                st.x[4 * h + s][1] = *reinterpret_cast<const float *>(xr + 64);
            }
        }
    };
    Stage cur, nxt;
    fetch(c_begin, cur);

    // ---- the tile's own x rows (update input and residual): requested now, parked in LDS in phase 2 -- nothing in
    // the prologue waits for a load, so every first-touch latency of this kernel overlaps ----
    float4 xtile = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 16 * 16) {
        const int row = min(row0 + (tid >> 4), p.n_out - 1);
        xtile = *reinterpret_cast<const float4 *>(xo + (long long)row * p.x_sr + 4 * (tid & 15));
    }
    // update weights of waves 0..3 (A operand of phase 3: lane (i, kk) holds W[16 ft + i][4 s + kk]), requested now
    float wfrag[32];
    if (ch == 0) {
#pragma unroll
        for (int s = 0; s < 32; ++s) wfrag[s] = p.weight[(16 * kq + i16) * 128 + 4 * s + kk];
    }

    // every small operand of phases 2 and 3 is requested here as well: its latency then hides under phase 1 instead of
    // forming a chain of dependent loads behind it
    float relv[2][4];
    {
        const float *relb = p.rel + (long long)outer * p.rel_so + 32 * ch + i16;   // lane owns column i16 of its two tiles
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < 4; ++t) relv[c][t] = t < p.n_rel ? relb[(long long)t * p.rel_sr + 16 * c] : 0.f;
    }
    const int f0 = 16 * kq + 4 * kk;   // first of this lane's 4 features in phase 3 (feature tile = kq)
    float biasv[4], lnw[4], lnb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        biasv[r] = p.bias ? p.bias[f0 + r] : 0.f;
        lnw[r] = (p.flags & DL_LN) ? p.ln_w[f0 + r] : 1.f;
        lnb[r] = (p.flags & DL_LN) ? p.ln_b[f0 + r] : 0.f;
    }
    float bndv[2] = {0.f, 0.f};   // boundary addends of the two tile elements this thread finalises in phase 2
    if (p.has_bnd) {
        const long long bnd_row = p.bnd_rows ? p.bnd_rows[outer] : -1;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + 512 * u;
            const int row = row0 + (e >> 6), col = e & 63;
            if (row < p.n_out && (bnd_row < 0 || bnd_row == row))
                bndv[u] = p.bnd[(long long)outer * p.bnd_so + (long long)row * p.bnd_sr + col];   // (bnd_sr == 0 for a point)
        }
    }

    // ---- phase 1: adjacency product ----
    f32x4 acc[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[c][t][r] = 0.f;
    for (int chunk = c_begin; chunk < c_end; chunk += 2) {
        fetch(min(chunk + 2, c_end - 2), nxt);   // (the last stage re-reads itself: harmless)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t aw[4] = {cur.a[h].x, cur.a[h].y, cur.a[h].z, cur.a[h].w};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bool in = 16 * (chunk + h) + 4 * s + kk < p.n_in;   // padding rows: multiplicity 0 times an exact 0
                const float x0 = in ? cur.x[4 * h + s][0] : 0.f, x1 = in ? cur.x[4 * h + s][1] : 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float a = (float)((aw[s] >> (8 * t)) & 0xffu);
                    acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, x0, acc[0][t], 0, 0, 0);
                    acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, x1, acc[1][t], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    }

    // ---- phase 2: rel scaling (lane owns column i16 of each tile), k-quarters through LDS, boundary ----
    {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            f32x4 tot;
#pragma unroll
            for (int r = 0; r < 4; ++r) tot[r] = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < p.n_rel) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) tot[r] += relv[c][t] * acc[c][t][r];
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((kq * 4 + 2 * ch + c) * 4 + r) * 64 + lane] = tot[r];
        }
        if (tid < 16 * 16) *reinterpret_cast<float4 *>(x_lds + (tid >> 4) * DL_ROW_STRIDE + 4 * (tid & 15)) = xtile;
    }
    __syncthreads();
    {
        // 1024 tile elements, two per thread.  D layout: lane l, reg r -> tile row 4 (l >> 4) + r, column l & 15
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + 512 * u;
            const int r = e >> 6, col = e & 63;
            const int ct = col >> 4, j = col & 15;
            const int l = (r >> 2) * 16 + j, reg = r & 3;
            float v = red[((0 * 4 + ct) * 4 + reg) * 64 + l];
#pragma unroll
            for (int q = 1; q < 4; ++q) v += red[((q * 4 + ct) * 4 + reg) * 64 + l];
            if (p.has_bnd) v += bndv[u];   // (exactly 0 where there is no boundary value: same bits as not adding)
            agg_lds[r * DL_ROW_STRIDE + col] = v;
        }
    }
    __syncthreads();

    // ---- phase 3: update (waves 0..3, feature tile ft = kq) ----
    const int ft = kq;
    f32x4 d;
#pragma unroll
    for (int r = 0; r < 4; ++r) d[r] = 0.f;
    if (ch == 0) {
        // B operand: lane (kk, j) holds data[row j][4 s + kk]; data = cat[x, agg]
#pragma unroll
        for (int s = 0; s < 16; ++s)
            d = __builtin_amdgcn_mfma_f32_16x16x4f32(wfrag[s], x_lds[i16 * DL_ROW_STRIDE + 4 * s + kk], d, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 16; ++s)
            d = __builtin_amdgcn_mfma_f32_16x16x4f32(wfrag[16 + s], agg_lds[i16 * DL_ROW_STRIDE + 4 * s + kk], d, 0, 0, 0);
    }
    // D: lane l, reg r -> feature 16 ft + 4 (l >> 4) + r of tile row l & 15
    float y[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = d[r] + biasv[r];
    if (p.flags & DL_LN) {
        // two-pass LayerNorm over the 64 features of a row: 4 regs x 4 lane groups x 4 waves
        float s = (y[0] + y[1]) + (y[2] + y[3]);
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (ch == 0 && kk == 0) ln_part[0][ft][i16] = s;
        __syncthreads();
        const float mean = (((ln_part[0][0][i16] + ln_part[0][1][i16]) + ln_part[0][2][i16]) + ln_part[0][3][i16]) * (1.f / 64.f);
        float q = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float dv = y[r] - mean;
            q += dv * dv;
        }
        q += __shfl_xor(q, 16);
        q += __shfl_xor(q, 32);
        if (ch == 0 && kk == 0) ln_part[1][ft][i16] = q;
        __syncthreads();
        const float var = (((ln_part[1][0][i16] + ln_part[1][1][i16]) + ln_part[1][2][i16]) + ln_part[1][3][i16]) * (1.f / 64.f);
        const float rstd = 1.f / sqrtf(var + p.eps);
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = (y[r] - mean) * rstd * lnw[r] + lnb[r];
    }
    if (ch == 0) {
        if (p.flags & DL_RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = fmaxf(y[r], 0.f);
        }
        if (p.flags & DL_RESIDUAL) {
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] += x_lds[i16 * DL_ROW_STRIDE + f0 + r];
        }
        const int row = row0 + i16;
        if (row < p.n_out)
            *reinterpret_cast<float4 *>(p.out + (long long)outer * p.out_so + (long long)row * p.out_sr + f0) =
                make_float4(y[0], y[1], y[2], y[3]);
    }
}

static bool dl_ok16(const ultra_mat *m) {
    return (reinterpret_cast<uintptr_t>(m->ptr) & 15u) == 0 && m->stride_row % 4 == 0 && m->stride_outer % 4 == 0;
}

// Called by ultra_nbf_dense_layer (rspmm_api.hip) with the plan uploaded.
int launch_dense_layer(ultra_plan *p, const ultra_mat *rel, const ultra_mat *x, const ultra_mat *bnd, const int64_t *bnd_rows,
                       const void *weight, const void *bias, const void *ln_w, const void *ln_b, float eps, int flags,
                       const ultra_mat *out, hipStream_t stream) {
    if (!(p->flags & ULTRA_PLAN_DENSE) || p->a16.empty()) {
        set_error("ultra_nbf_dense_layer needs a ULTRA_PLAN_DENSE plan with at most 4 relation types");
        return ULTRA_ERR_UNSUPPORTED;
    }
    if (out->row_len != 64 || p->num_out != p->num_in) {
        set_error("ultra_nbf_dense_layer: hidden dim 64 on a square graph only");
        return ULTRA_ERR_UNSUPPORTED;
    }
    if (!dl_ok16(rel) || !dl_ok16(x) || !dl_ok16(out) || (bnd && !dl_ok16(bnd)) || (reinterpret_cast<uintptr_t>(weight) & 15u)) {
        set_error("ultra_nbf_dense_layer: operands must be 16-byte aligned with strides that are multiples of 4");
        return ULTRA_ERR_UNSUPPORTED;
    }
    if ((uint64_t)p->num_in * (uint64_t)x->stride_row * 4u >= (1ull << 32)) {
        set_error("ultra_nbf_dense_layer: an input slice (rows * stride_row) exceeds 4 GiB");
        return ULTRA_ERR_UNSUPPORTED;
    }
    DenseLayerParams dp;
    dp.a16 = reinterpret_cast<const uint32_t *>(p->d.a16);
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
    dp.n_chunk = p->a16_chunks, dp.n_rt16 = (int)((p->num_out + 15) / 16);
    dp.has_bnd = bnd ? 1 : 0;
    dp.flags = flags;
    dp.eps = eps;
    const long long blocks = (long long)dp.n_rt16 * out->n_outer;
    hipLaunchKernelGGL(dense_layer_kernel, dim3((unsigned)blocks), dim3(512), 0, stream, dp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error(std::string("dense_layer_kernel launch: ") + hipGetErrorString(e));
        return ULTRA_ERR_HIP;
    }
    return ULTRA_OK;
}

}  // namespace ultra
