// Dense epilogues of the NBFNet layer on the MI355X matrix cores (f32-in / f32-accumulate MFMA).
//
//   ultra_conv_update : out = [+x] relu( LayerNorm( W . [x ; agg] + b ) )        layers.py:233-240 (+ models.py:158-160)
//   ultra_readout     : score = w2 . relu( W1[:, :d] . h[t] + qb[sample] ) + b2  models.py:202-209 with the query half of
//                       the concatenated feature folded into a per-sample bias qb = W1[:, d:] . query + b1
//
// Both are skinny GEMMs (K = 128 / 64, N = 64 / 128) over M = batch * num_node rows, i.e. HBM/L2-bound
// epilogues: one wave owns 32 data rows and computes the TRANSPOSED product D[feature][row] with
// v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chains, MI355X_MICROARCH.md), so that a data row's features
// land in one lane pair (lane, lane ^ 32): LayerNorm / the final dot product need one cross-lane
// shuffle instead of a 32-lane reduction.  A-operand = weights, staged once per workgroup in LDS in
// MFMA fragment order (ds_read_b128, conflict free); B-operand = the data rows, loaded straight from
// global memory as 16-byte chunks in a K-permuted order (lane half h takes chunks 2i + h), which also
// leaves the residual input already in registers for the epilogue.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "../../include/ultra_rspmm.h"
#include "plan.hpp"

namespace ultra {

using f32x16 = float __attribute__((ext_vector_type(16)));

struct ConvParams {
    const float *x;
    const float *agg;
    const float *weight;  // (64, 128) row-major
    const float *bias, *ln_w, *ln_b;
    float *out;
    long long rows;
    float eps;
    int flags;
};

enum { CONV_LN = 1, CONV_RELU = 2, CONV_RESIDUAL = 4 };

// feature owned by accumulator register r of feature tile m in lane half h (32x32 C/D layout)
__device__ __forceinline__ int feat_of(int m, int r, int h) { return 32 * m + (r & 3) + 8 * (r >> 2) + 4 * h; }

__global__ void __launch_bounds__(512, 2) conv_update_kernel(const ConvParams p) {
    // [tile m][i][lane][q] : W[32 m + (lane & 31)][8 i + 4 (lane >> 5) + q]
    __shared__ __attribute__((aligned(16))) float lds_w[2 * 16 * 64 * 4];
    __shared__ float lds_vec[3 * 64];
    const int tid = threadIdx.x;
    // 16-byte staging loads: fragment (m, i, lane) = 4 consecutive k of one weight row
    for (int idx4 = tid; idx4 < 2 * 16 * 64; idx4 += blockDim.x) {
        const int l = idx4 & 63, i = (idx4 >> 6) & 15, m = idx4 >> 10;
        reinterpret_cast<float4 *>(lds_w)[idx4] =
            *reinterpret_cast<const float4 *>(p.weight + (32 * m + (l & 31)) * 128 + 8 * i + 4 * (l >> 5));
    }
    if (tid < 64) {
        lds_vec[tid] = p.bias ? p.bias[tid] : 0.f;
        lds_vec[64 + tid] = (p.flags & CONV_LN) ? p.ln_w[tid] : 1.f;
        lds_vec[128 + tid] = (p.flags & CONV_LN) ? p.ln_b[tid] : 0.f;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const float4 *w4 = reinterpret_cast<const float4 *>(lds_w);
    const long long ntile = (p.rows + 31) / 32;
    // persistent waves: the rows of the NEXT tile are requested before the 128 MFMAs of the current one, so
    // their HBM/L2 latency hides under ~8k cycles of matrix work
    const int wpb = blockDim.x >> 6;   // waves per block
    const long long tstride = (long long)gridDim.x * wpb;
    long long tile = (long long)blockIdx.x * wpb + wave;
    float4 bx[8], ba[8];
    if (tile < ntile) {
        const long long r0 = tile * 32 + j;
        const long long rc0 = r0 < p.rows ? r0 : p.rows - 1;
        const float4 *xr = reinterpret_cast<const float4 *>(p.x + rc0 * 64);
        const float4 *ar = reinterpret_cast<const float4 *>(p.agg + rc0 * 64);
#pragma unroll
        for (int i = 0; i < 8; ++i) bx[i] = xr[2 * i + h];
#pragma unroll
        for (int i = 0; i < 8; ++i) ba[i] = ar[2 * i + h];
    }
    for (; tile < ntile; tile += tstride) {
        const long long row = tile * 32 + j;
        const bool valid = row < p.rows;
        float4 bxn[8], ban[8];
        {
            const long long tn = tile + tstride;
            const long long rn = (tn < ntile ? tn : tile) * 32 + j;
            const long long rcn = rn < p.rows ? rn : p.rows - 1;
            const float4 *xr = reinterpret_cast<const float4 *>(p.x + rcn * 64);
            const float4 *ar = reinterpret_cast<const float4 *>(p.agg + rcn * 64);
#pragma unroll
            for (int i = 0; i < 8; ++i) bxn[i] = xr[2 * i + h];
#pragma unroll
            for (int i = 0; i < 8; ++i) ban[i] = ar[2 * i + h];
        }
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc0[r] = 0.f;
            acc1[r] = 0.f;
        }
        // weights for step i + 1 are fetched from LDS while the 8 MFMAs of step i run; the scheduling
        // barrier keeps the compiler from hoisting all 32 fragment reads (128 VGPRs) to the top
        float4 a0 = w4[(0 * 16 + 0) * 64 + lane];
        float4 a1 = w4[(1 * 16 + 0) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float4 b = i < 8 ? bx[i] : ba[i - 8];
            float4 a0n = a0, a1n = a1;
            if (i + 1 < 16) {
                a0n = w4[(0 * 16 + i + 1) * 64 + lane];
                a1n = w4[(1 * 16 + i + 1) * 64 + lane];
            }
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b.w, acc1, 0, 0, 0);
            a0 = a0n;
            a1 = a1n;
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue: bias, LayerNorm over the row's 64 features (32 here + 32 in lane ^ 32), ReLU, residual ----
        float v[2][16];
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            v[0][r] = acc0[r] + lds_vec[feat_of(0, r, h)];
            v[1][r] = acc1[r] + lds_vec[feat_of(1, r, h)];
            s += v[0][r] + v[1][r];
        }
        if (p.flags & CONV_LN) {
            s += __shfl_xor(s, 32);
            const float mean = s * (1.f / 64.f);
            float q = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d0 = v[0][r] - mean, d1 = v[1][r] - mean;
                q += d0 * d0 + d1 * d1;
            }
            q += __shfl_xor(q, 32);
            const float rstd = 1.f / sqrtf(q * (1.f / 64.f) + p.eps);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = feat_of(m, r, h);
                    v[m][r] = (v[m][r] - mean) * rstd * lds_vec[64 + f] + lds_vec[128 + f];
                }
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 y = make_float4(v[m][4 * g + 0], v[m][4 * g + 1], v[m][4 * g + 2], v[m][4 * g + 3]);
                if (p.flags & CONV_RELU) {
                    y.x = fmaxf(y.x, 0.f);
                    y.y = fmaxf(y.y, 0.f);
                    y.z = fmaxf(y.z, 0.f);
                    y.w = fmaxf(y.w, 0.f);
                }
                if (p.flags & CONV_RESIDUAL) {
                    const float4 xi = bx[4 * m + g];  // x[row][32 m + 8 g + 4 h ..]: the chunk this lane already holds
                    y.x += xi.x;
                    y.y += xi.y;
                    y.z += xi.z;
                    y.w += xi.w;
                }
                if (valid) *reinterpret_cast<float4 *>(p.out + row * 64 + 32 * m + 8 * g + 4 * h) = y;
            }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            bx[i] = bxn[i];
            ba[i] = ban[i];
        }
    }
}

struct ReadoutParams {
    const float *hidden;     // (batch, num_node, 64) contiguous
    const int64_t *t_index;  // (batch, n_cand) node ids, or NULL = identity (all-tail)
    const int64_t *triples;  // alternative to t_index: the raw (batch, n_cand, 3) [h, t, r] batch ...
    const int32_t *side;     // ... with side[b] = 1 -> candidates are the tails (column 1), 0 -> the heads (column 0)
    const float *w1;         // (128, 128) row-major; only the first 64 input columns are used here
    const float *qbias;      // (batch, 128) = W1[:, 64:] . query + b1, or NULL: computed here from ...
    const float *query;      // ... query (batch, 64) and
    const float *b1;         // ... mlp.0.bias (128); batch <= READOUT_MAX_INLINE_BATCH then
    const float *w2;         // (128)
    const float *b2;         // (1) = mlp.2.bias
    float *score;            // (batch, n_cand)
    long long batch, num_node, n_cand;
};

constexpr int READOUT_MAX_INLINE_BATCH = 32;

__global__ void __launch_bounds__(512, 2) readout_kernel(const ReadoutParams p) {
    // [tile m (4)][i (8)][lane][q] : W1[32 m + (lane & 31)][8 i + 4 (lane >> 5) + q]
    __shared__ __attribute__((aligned(16))) float lds_w[4 * 8 * 64 * 4];
    __shared__ float lds_w2[128];
    __shared__ float lds_qb[READOUT_MAX_INLINE_BATCH * 128];
    const int tid = threadIdx.x;
    if (!p.qbias) {
        // the query half of mlp.0 (models.py:166-170 concatenates query to every node feature): one 64-term dot
        // product per (sample, hidden unit), computed by every workgroup for itself instead of a GEMM launch
        for (int idx = tid; idx < (int)p.batch * 128; idx += blockDim.x) {
            const int b = idx >> 7, f = idx & 127;
            const float4 *wr = reinterpret_cast<const float4 *>(p.w1 + f * 128 + 64);
            const float4 *qr = reinterpret_cast<const float4 *>(p.query + b * 64);
            float acc = p.b1[f];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float4 w = wr[k], q = qr[k];
                acc += w.x * q.x;
                acc += w.y * q.y;
                acc += w.z * q.z;
                acc += w.w * q.w;
            }
            lds_qb[idx] = acc;
        }
    }
    for (int idx4 = tid; idx4 < 4 * 8 * 64; idx4 += blockDim.x) {
        const int l = idx4 & 63, i = (idx4 >> 6) & 7, m = idx4 >> 9;
        reinterpret_cast<float4 *>(lds_w)[idx4] =
            *reinterpret_cast<const float4 *>(p.w1 + (32 * m + (l & 31)) * 128 + 8 * i + 4 * (l >> 5));
    }
    if (tid < 128) lds_w2[tid] = p.w2[tid];
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const float4 *w4 = reinterpret_cast<const float4 *>(lds_w);
    const long long total = p.batch * p.n_cand;
    const long long ntile = (total + 31) / 32;
    const int wpb = blockDim.x >> 6;
    for (long long tile = (long long)blockIdx.x * wpb + wave; tile < ntile; tile += (long long)gridDim.x * wpb) {
        const long long row = tile * 32 + j;
        const bool valid = row < total;
        const long long rowc = valid ? row : total - 1;
        const long long b = rowc / p.n_cand;
        const long long c = rowc - b * p.n_cand;
        long long node = c;
        if (p.triples)
            node = p.triples[rowc * 3 + (p.side[b] ? 1 : 0)];
        else if (p.t_index)
            node = p.t_index[rowc];
        const float4 *hr = reinterpret_cast<const float4 *>(p.hidden + (b * p.num_node + node) * 64);
        float4 bh[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) bh[i] = hr[2 * i + h];
        f32x16 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        float4 a[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) a[m] = w4[(m * 8 + 0) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float4 an[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) an[m] = (i + 1 < 8) ? w4[(m * 8 + i + 1) * 64 + lane] : a[m];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].x, bh[i].x, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].y, bh[i].y, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].z, bh[i].z, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].w, bh[i].w, acc[m], 0, 0, 0);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) a[m] = an[m];
            __builtin_amdgcn_sched_barrier(0);
        }
        const float *qb = p.qbias ? p.qbias + b * 128 : lds_qb + b * 128;
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = feat_of(m, r, h);
                const float hid = fmaxf(acc[m][r] + qb[f], 0.f);
                s += hid * lds_w2[f];
            }
        s += __shfl_xor(s, 32);
        if (valid && h == 0) p.score[row] = s + p.b2[0];
    }
}

// All layers' relation_projection MLPs (layers.py:80: Linear(64, 64) -> ReLU -> Linear(64, 64) applied to the
// relation representations, models.py:184-185) in ONE launch: workgroup = (layer, 4 row tiles), both weight
// matrices of the layer staged in LDS in MFMA fragment order.  Transposed products as in conv_update: a data row's
// features live in one lane pair, so the hidden activation of the first product is consumed by the second one
// straight from the accumulator registers -- the contraction index of the second product is simply enumerated in
// the order the accumulators hold it (feat_of), and the second weight matrix is staged in that same order.
struct RelProjParams {
    const float *x;        // (rows, 64)
    const float *w0, *b0;  // (n_layer, 64, 64) row-major [out][in], (n_layer, 64)
    const float *w2, *b2;
    float *out;            // (n_layer, rows, 64)
    long long rows;
    int n_layer;
};

__global__ void __launch_bounds__(256) relation_projection_kernel(const RelProjParams p) {
    // lds_w0[m][i][lane][q] = W0[32 m + (lane & 31)][8 i + 4 (lane >> 5) + q]          (k = 8 i + 4 h + q: x chunk 2 i + h)
    // lds_w2[m2][s][lane]   = W2[32 m2 + (lane & 31)][feat_of(s >> 4, s & 15, lane >> 5)]  (k enumerated as the accumulators)
    __shared__ __attribute__((aligned(16))) float lds_w0[2 * 8 * 64 * 4];
    __shared__ __attribute__((aligned(16))) float lds_w2[2 * 32 * 64];
    __shared__ float lds_b[2 * 64];
    const int tid = threadIdx.x;
    const int layer = blockIdx.y;
    const float *w0 = p.w0 + (size_t)layer * 64 * 64, *w2 = p.w2 + (size_t)layer * 64 * 64;
    for (int idx4 = tid; idx4 < 2 * 8 * 64; idx4 += 256) {
        const int l = idx4 & 63, i = (idx4 >> 6) & 7, m = idx4 >> 9;
        reinterpret_cast<float4 *>(lds_w0)[idx4] =
            *reinterpret_cast<const float4 *>(w0 + (32 * m + (l & 31)) * 64 + 8 * i + 4 * (l >> 5));
    }
    for (int idx = tid; idx < 2 * 32 * 64; idx += 256) {
        const int l = idx & 63, st = (idx >> 6) & 31, m2 = idx >> 11;
        lds_w2[idx] = w2[(32 * m2 + (l & 31)) * 64 + feat_of(st >> 4, st & 15, l >> 5)];
    }
    if (tid < 64) {
        lds_b[tid] = p.b0[layer * 64 + tid];
        lds_b[64 + tid] = p.b2[layer * 64 + tid];
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const long long tile = (long long)blockIdx.x * 4 + wave;
    if (tile * 32 >= p.rows) return;
    const long long row = tile * 32 + j;
    const bool valid = row < p.rows;
    const float4 *xr = reinterpret_cast<const float4 *>(p.x + (valid ? row : p.rows - 1) * 64);
    float4 bx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bx[i] = xr[2 * i + h];
    f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const float4 *w04 = reinterpret_cast<const float4 *>(lds_w0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float4 a0 = w04[(0 * 8 + i) * 64 + lane], a1 = w04[(1 * 8 + i) * 64 + lane];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, bx[i].x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, bx[i].x, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, bx[i].y, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, bx[i].y, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, bx[i].z, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, bx[i].z, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, bx[i].w, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, bx[i].w, acc[1], 0, 0, 0);
    }
    // hidden = relu(. + b0), kept in the accumulator layout: it is the B operand of the second product
    float hid[2][16];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) hid[m][r] = fmaxf(acc[m][r] + lds_b[feat_of(m, r, h)], 0.f);
    f32x16 out[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[m][r] = 0.f;
#pragma unroll
    for (int st = 0; st < 32; ++st) {
        const float b = hid[st >> 4][st & 15];
        out[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(lds_w2[(0 * 32 + st) * 64 + lane], b, out[0], 0, 0, 0);
        out[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(lds_w2[(1 * 32 + st) * 64 + lane], b, out[1], 0, 0, 0);
    }
    if (valid) {
        float *orow = p.out + ((size_t)layer * p.rows + row) * 64;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int f = 32 * m + 8 * g + 4 * h;
                float4 y;
                y.x = out[m][4 * g + 0] + lds_b[64 + f + 0];
                y.y = out[m][4 * g + 1] + lds_b[64 + f + 1];
                y.z = out[m][4 * g + 2] + lds_b[64 + f + 2];
                y.w = out[m][4 * g + 3] + lds_b[64 + f + 3];
                *reinterpret_cast<float4 *>(orow + f) = y;
            }
    }
}

static int grid_for(long long ntile, int waves_per_block, int blocks_per_cu = 2) {
    static int cu = 0;   // queried once (kept out of hipGraph capture)
    if (cu == 0) {
        int dev = 0, v = 0;
        cu = 256;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            cu = v;
    }
    long long blocks = (ntile + waves_per_block - 1) / waves_per_block;
    const long long cap = (long long)cu * blocks_per_cu;   // 8 waves per CU = 2 per SIMD (the kernels' VGPR budget), persistent over tiles
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

}  // namespace ultra

using namespace ultra;

extern "C" {

int32_t ultra_conv_update(const void *x, const void *agg, const void *weight, const void *bias, const void *ln_weight,
                          const void *ln_bias, void *out, int64_t rows, int32_t input_dim, int32_t output_dim, float eps,
                          int32_t flags, void *stream) {
    if (input_dim != 64 || output_dim != 64) {
        set_error("ultra_conv_update: only input_dim = output_dim = 64 is built (the ULTRA checkpoints' shape)");
        return ULTRA_ERR_UNSUPPORTED;
    }
    if (!x || !agg || !weight || !out || rows < 0 || ((flags & CONV_LN) && (!ln_weight || !ln_bias))) {
        set_error("ultra_conv_update: NULL operand");
        return ULTRA_ERR_INVALID;
    }
    if (rows == 0) return ULTRA_OK;
    ConvParams p;
    p.x = (const float *)x;
    p.agg = (const float *)agg;
    p.weight = (const float *)weight;
    p.bias = (const float *)bias;
    p.ln_w = (const float *)ln_weight;
    p.ln_b = (const float *)ln_bias;
    p.out = (float *)out;
    p.rows = rows;
    p.eps = eps;
    p.flags = flags;
    // big inputs: one 8-wave workgroup per CU (half the weight staging and workgroup launches of two 4-wave ones:
    // 33.6 -> 31.9 us at 116 k rows); small inputs keep 4-wave workgroups so that the tiles spread over more CUs
    const long long ntile = (rows + 31) / 32;
    const int threads = ntile >= 2048 ? 512 : 256;
    const int grid = grid_for(ntile, threads / 64, threads == 512 ? 1 : 2);
    (void)hipGetLastError();   // drop any stale error left by other users of the runtime
    hipLaunchKernelGGL(conv_update_kernel, dim3(grid), dim3(threads), 0, reinterpret_cast<hipStream_t>(stream), p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error(std::string("conv_update_kernel launch: ") + hipGetErrorString(e));
        return ULTRA_ERR_HIP;
    }
    return ULTRA_OK;
}

static int check_query_bias(const void *qbias, const void *query, const void *b1, int64_t batch) {
    if (qbias) return ULTRA_OK;
    if (!query || !b1) {
        set_error("ultra_readout: pass qbias, or query and b1");
        return ULTRA_ERR_INVALID;
    }
    if (batch > READOUT_MAX_INLINE_BATCH) {
        set_error("ultra_readout: the in-kernel query bias serves batch <= 32; pass a precomputed qbias beyond");
        return ULTRA_ERR_UNSUPPORTED;
    }
    return ULTRA_OK;
}

int32_t ultra_readout(const void *hidden, const int64_t *t_index, const void *w1, const void *qbias, const void *query,
                      const void *b1, const void *w2, const void *b2, void *score, int64_t batch, int64_t num_node,
                      int64_t n_cand, int32_t hidden_dim, int32_t feature_dim, void *stream) {
    if (hidden_dim != 64 || feature_dim != 128) {
        set_error("ultra_readout: only hidden_dim = 64, feature_dim = 128 is built (the ULTRA checkpoints' shape)");
        return ULTRA_ERR_UNSUPPORTED;
    }
    if (!hidden || !w1 || !w2 || !b2 || !score || batch < 0 || n_cand < 0) {
        set_error("ultra_readout: NULL operand");
        return ULTRA_ERR_INVALID;
    }
    if (int rc = check_query_bias(qbias, query, b1, batch)) return rc;
    if (batch * n_cand == 0) return ULTRA_OK;
    ReadoutParams p;
    p.query = (const float *)query;
    p.b1 = (const float *)b1;
    p.hidden = (const float *)hidden;
    p.t_index = t_index;
    p.triples = nullptr;
    p.side = nullptr;
    p.w1 = (const float *)w1;
    p.qbias = (const float *)qbias;
    p.w2 = (const float *)w2;
    p.b2 = (const float *)b2;
    p.score = (float *)score;
    p.batch = batch;
    p.num_node = num_node;
    p.n_cand = n_cand;
    const long long ntile = (batch * n_cand + 31) / 32;
    const int threads = ntile >= 2048 ? 512 : 256;   // as in ultra_conv_update
    const int grid = grid_for(ntile, threads / 64, threads == 512 ? 1 : 2);
    (void)hipGetLastError();   // drop any stale error left by other users of the runtime
    hipLaunchKernelGGL(readout_kernel, dim3(grid), dim3(threads), 0, reinterpret_cast<hipStream_t>(stream), p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error(std::string("readout_kernel launch: ") + hipGetErrorString(e));
        return ULTRA_ERR_HIP;
    }
    return ULTRA_OK;
}

int32_t ultra_readout_batch(const void *hidden, const int64_t *triples, const int32_t *side, const void *w1,
                            const void *qbias, const void *query, const void *b1, const void *w2, const void *b2, void *score,
                            int64_t batch, int64_t num_node, int64_t n_cand, int32_t hidden_dim, int32_t feature_dim,
                            void *stream) {
    if (hidden_dim != 64 || feature_dim != 128) {
        set_error("ultra_readout_batch: only hidden_dim = 64, feature_dim = 128 is built (the ULTRA checkpoints' shape)");
        return ULTRA_ERR_UNSUPPORTED;
    }
    if (!hidden || !triples || !side || !w1 || !w2 || !b2 || !score || batch < 0 || n_cand < 0) {
        set_error("ultra_readout_batch: NULL operand");
        return ULTRA_ERR_INVALID;
    }
    if (int rc = check_query_bias(qbias, query, b1, batch)) return rc;
    if (batch * n_cand == 0) return ULTRA_OK;
    ReadoutParams p;
    p.query = (const float *)query;
    p.b1 = (const float *)b1;
    p.hidden = (const float *)hidden;
    p.t_index = nullptr;
    p.triples = triples;
    p.side = side;
    p.w1 = (const float *)w1;
    p.qbias = (const float *)qbias;
    p.w2 = (const float *)w2;
    p.b2 = (const float *)b2;
    p.score = (float *)score;
    p.batch = batch;
    p.num_node = num_node;
    p.n_cand = n_cand;
    const long long ntile = (batch * n_cand + 31) / 32;
    const int threads = ntile >= 2048 ? 512 : 256;   // as in ultra_conv_update
    const int grid = grid_for(ntile, threads / 64, threads == 512 ? 1 : 2);
    (void)hipGetLastError();   // drop any stale error left by other users of the runtime
    hipLaunchKernelGGL(readout_kernel, dim3(grid), dim3(threads), 0, reinterpret_cast<hipStream_t>(stream), p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error(std::string("readout_kernel launch: ") + hipGetErrorString(e));
        return ULTRA_ERR_HIP;
    }
    return ULTRA_OK;
}

int32_t ultra_relation_projection(const void *x, const void *w0, const void *b0, const void *w2, const void *b2, void *out,
                                  int64_t rows, int32_t n_layer, int32_t dim, void *stream) {
    if (dim != 64) {
        set_error("ultra_relation_projection: only dim = 64 is built (the ULTRA checkpoints' shape)");
        return ULTRA_ERR_UNSUPPORTED;
    }
    if (!x || !w0 || !b0 || !w2 || !b2 || !out || rows < 0 || n_layer < 0) {
        set_error("ultra_relation_projection: NULL operand");
        return ULTRA_ERR_INVALID;
    }
    if (rows == 0 || n_layer == 0) return ULTRA_OK;
    RelProjParams p;
    p.x = (const float *)x;
    p.w0 = (const float *)w0;
    p.b0 = (const float *)b0;
    p.w2 = (const float *)w2;
    p.b2 = (const float *)b2;
    p.out = (float *)out;
    p.rows = rows;
    p.n_layer = n_layer;
    const dim3 grid((unsigned)((rows + 127) / 128), (unsigned)n_layer);
    (void)hipGetLastError();   // drop any stale error left by other users of the runtime
    hipLaunchKernelGGL(relation_projection_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error(std::string("relation_projection_kernel launch: ") + hipGetErrorString(e));
        return ULTRA_ERR_HIP;
    }
    return ULTRA_OK;
}

}  // extern "C"
