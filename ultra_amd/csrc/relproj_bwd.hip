// Backward of the entity model's relation_projection MLPs (layers.py:80, models.py:184-185) for ALL layers in three launches.
//
//     h_l   = relu(x W0_l^T + b0_l)            out_l = h_l W2_l^T + b2_l            l = 0 .. n_layer - 1,  x (rows, 64)
//     gh_l  = (gout_l W2_l) * [h_l > 0]
//     gx    = sum_l gh_l W0_l
//     gW2_l = gout_l^T h_l    gb2_l = sum_rows gout_l        gW0_l = gh_l^T x    gb0_l = sum_rows gh_l
//
// torch ran this as two batched products forward and, backward, four batched GEMMs with K = rows that hipBLASLt serves badly
// (2 x 56 us for 64 x 64 x 3,792), two bias reductions over (6, 3792, 64) (49 + 47 us), the copies that un-stack the gradients
// for the six layers' parameters (12 x 4.6 us) and a dozen elementwise launches: ~ 330 us of a 3.5 ms step at FB15k237's size
// (profiles/r6_01_timeline_eager.txt).  Here:
//
//   relproj_bwd_rows_kernel     workgroup = (64 rows, layer), wave = 16 features: recomputes h (nothing but x was saved), then
//                               gh and this layer's share of gx as two more v_mfma_f32_16x16x4_f32 chains whose A operands are the
//                               TRANSPOSED weight matrices (read column-wise from L2: 64 x 64 floats); h, gh and the gx share go to
//                               scratch.
//   relproj_bwd_weights_kernel  workgroup = (row chunk, layer, which of the two products): dW = A^T B with K = the chunk's rows,
//                               A = gout_l (gW2) or gh_l (gW0), B = h_l or x; the bias gradient is the column sum of A, taken from the
//                               same operand registers.  Partial sums per chunk.
//   relproj_bwd_reduce_kernel   adds the chunks (ascending) and the layers' shares of gx (ascending): a fixed order, no atomics.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "../../include/ultra_nbfnet.h"
#include "../../include/ultra_rspmm.h"
#include "plan.hpp"
#include "device_scope.hpp"

namespace ultra {

using f32x4 = float __attribute__((ext_vector_type(4)));

constexpr int RPB_MAX_LAYERS = 8;
constexpr int RPB_TILES = 4;           // 16-row tiles per workgroup of the rows kernel
constexpr int RPB_STRIDE = 68;         // LDS row stride (floats): 16-byte aligned, rows 4 banks apart
constexpr int RPB_CHUNKS = 16;         // row chunks of the weights kernel (partial sums per chunk)

struct RelProjBwdParams {
    const float *x;                                   // (rows, 64)
    const float *w0[RPB_MAX_LAYERS], *b0[RPB_MAX_LAYERS], *w2[RPB_MAX_LAYERS];
    const float *gout[RPB_MAX_LAYERS];                // per layer (rows, 64)
    float *h, *gh, *gxl;                              // scratch (n_layer, rows, 64) each
    float *part;                                      // scratch (RPB_CHUNKS, 2 n_layer, 64 * 64 + 64)
    float *gx;                                        // (rows, 64)
    float *gw0, *gb0, *gw2, *gb2;                     // stacked (n_layer, 64, 64) / (n_layer, 64)
    long long rows;
    int n_layer;
};

__global__ void __launch_bounds__(256) relproj_bwd_rows_kernel(const RelProjBwdParams p) {
    __shared__ __attribute__((aligned(16))) float x_lds[RPB_TILES][16 * RPB_STRIDE];
    __shared__ __attribute__((aligned(16))) float g_lds[RPB_TILES][16 * RPB_STRIDE];
    __shared__ __attribute__((aligned(16))) float h_lds[RPB_TILES][16 * RPB_STRIDE];
    __shared__ __attribute__((aligned(16))) float gh_lds[RPB_TILES][16 * RPB_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, kk = lane >> 4;
    const int layer = blockIdx.y;
    const long long row_base = (long long)blockIdx.x * (16 * RPB_TILES);
    const float *w0 = p.w0[layer], *w2 = p.w2[layer], *gout = p.gout[layer];
#pragma unroll
    for (int t = 0; t < RPB_TILES; ++t) {
        const long long row = min(row_base + 16 * t + (tid >> 4), p.rows - 1);
        *reinterpret_cast<float4 *>(x_lds[t] + (tid >> 4) * RPB_STRIDE + 4 * (tid & 15)) =
            *reinterpret_cast<const float4 *>(p.x + row * 64 + 4 * (tid & 15));
        *reinterpret_cast<float4 *>(g_lds[t] + (tid >> 4) * RPB_STRIDE + 4 * (tid & 15)) =
            *reinterpret_cast<const float4 *>(gout + row * 64 + 4 * (tid & 15));
    }
    // A operands, lane (i, kk), step s (contraction index 4 s + kk):
    //   a0 : W0[16 wave + i][4 s + kk]        h  = x W0^T     (M = hidden feature, K = input feature)
    //   a2t: W2[4 s + kk][16 wave + i]        gh = gout W2    (M = hidden feature, K = output feature)
    //   a0t: W0[4 s + kk][16 wave + i]        gx = gh W0      (M = input feature,  K = hidden feature)
    float a0[16], a2t[16], a0t[16];
    {
        const float4 *r0 = reinterpret_cast<const float4 *>(w0 + (16 * wave + i16) * 64);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float4 v0 = r0[s];
            a0[s] = kk == 0 ? v0.x : (kk == 1 ? v0.y : (kk == 2 ? v0.z : v0.w));
            a2t[s] = w2[(4 * s + kk) * 64 + 16 * wave + i16];
            a0t[s] = w0[(4 * s + kk) * 64 + 16 * wave + i16];
        }
    }
    const int f0 = 16 * wave + 4 * kk;   // D: lane l, reg r -> M index 16 wave + 4 (l >> 4) + r, tile row l & 15
    float b0v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) b0v[r] = p.b0[layer][f0 + r];
    __syncthreads();
    f32x4 d[RPB_TILES];
    // ---- h = relu(x W0^T + b0) ----
#pragma unroll
    for (int t = 0; t < RPB_TILES; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) d[t][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int t = 0; t < RPB_TILES; ++t)
            d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], x_lds[t][i16 * RPB_STRIDE + 4 * s + kk], d[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < RPB_TILES; ++t) {
        float4 hv;
        hv.x = fmaxf(d[t][0] + b0v[0], 0.f), hv.y = fmaxf(d[t][1] + b0v[1], 0.f);
        hv.z = fmaxf(d[t][2] + b0v[2], 0.f), hv.w = fmaxf(d[t][3] + b0v[3], 0.f);
        *reinterpret_cast<float4 *>(&h_lds[t][i16 * RPB_STRIDE + f0]) = hv;
        const long long row = row_base + 16 * t + i16;
        if (row < p.rows) *reinterpret_cast<float4 *>(p.h + ((size_t)layer * p.rows + row) * 64 + f0) = hv;
    }
    // ---- gh = (gout W2) * [h > 0]  (this lane's own h values: the same D layout) ----
#pragma unroll
    for (int t = 0; t < RPB_TILES; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) d[t][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int t = 0; t < RPB_TILES; ++t)
            d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2t[s], g_lds[t][i16 * RPB_STRIDE + 4 * s + kk], d[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < RPB_TILES; ++t) {
        const float4 hv = *reinterpret_cast<const float4 *>(&h_lds[t][i16 * RPB_STRIDE + f0]);
        float4 gv;
        gv.x = hv.x > 0.f ? d[t][0] : 0.f, gv.y = hv.y > 0.f ? d[t][1] : 0.f;
        gv.z = hv.z > 0.f ? d[t][2] : 0.f, gv.w = hv.w > 0.f ? d[t][3] : 0.f;
        *reinterpret_cast<float4 *>(&gh_lds[t][i16 * RPB_STRIDE + f0]) = gv;
        const long long row = row_base + 16 * t + i16;
        if (row < p.rows) *reinterpret_cast<float4 *>(p.gh + ((size_t)layer * p.rows + row) * 64 + f0) = gv;
    }
    __syncthreads();
    // ---- this layer's share of gx = gh W0 ----
#pragma unroll
    for (int t = 0; t < RPB_TILES; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) d[t][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int t = 0; t < RPB_TILES; ++t)
            d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0t[s], gh_lds[t][i16 * RPB_STRIDE + 4 * s + kk], d[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < RPB_TILES; ++t) {
        const long long row = row_base + 16 * t + i16;
        if (row < p.rows)
            *reinterpret_cast<float4 *>(p.gxl + ((size_t)layer * p.rows + row) * 64 + f0) =
                make_float4(d[t][0], d[t][1], d[t][2], d[t][3]);
    }
}

// dW[f][k] = sum_row A[row][f] B[row][k] over this chunk's rows; db[f] = sum_row A[row][f].
// blockIdx = (chunk, 2 layer + which): which 0 -> (A, B) = (gout_l, h_l): gW2, gb2;  which 1 -> (gh_l, x): gW0, gb0.
// Wave w owns M rows 16 w .. 16 w + 15 and all four 16-column N tiles.
__global__ void __launch_bounds__(256) relproj_bwd_weights_kernel(const RelProjBwdParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, kk = lane >> 4;
    const int chunk = blockIdx.x, layer = blockIdx.y >> 1, which = blockIdx.y & 1;
    const float *A = which == 0 ? p.gout[layer] : p.gh + (size_t)layer * p.rows * 64;
    const float *B = which == 0 ? p.h + (size_t)layer * p.rows * 64 : p.x;
    const long long per = ((p.rows + RPB_CHUNKS - 1) / RPB_CHUNKS + 3) / 4 * 4;      // rows per chunk, whole steps of four
    const long long lo = (long long)chunk * per, hi = min(lo + per, p.rows);
    f32x4 d[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) d[nt][r] = 0.f;
    float bsum = 0.f;
    for (long long r0 = lo; r0 < hi; r0 += 4) {
        const long long row = r0 + kk;
        const bool ok = row < hi;
        const long long rc = ok ? row : p.rows - 1;
        const float a = ok ? A[rc * 64 + 16 * wave + i16] : 0.f;
        bsum += a;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const float b = B[rc * 64 + 16 * nt + i16];      // (times a == 0 past the end: finite operands only)
            d[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d[nt], 0, 0, 0);
        }
    }
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
    float *dst = p.part + ((size_t)chunk * 2 * p.n_layer + blockIdx.y) * (64 * 64 + 64);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(16 * wave + 4 * kk + r) * 64 + 16 * nt + i16] = d[nt][r];
    if (kk == 0) dst[64 * 64 + 16 * wave + i16] = bsum;
}

// blocks [0, n_w): one of the 2 n_layer (64 x 64 + 64) gradient blocks each, chunks added in ascending order;
// blocks [n_w, ...): gx rows, the layers' shares added in ascending order.
__global__ void __launch_bounds__(256) relproj_bwd_reduce_kernel(const RelProjBwdParams p) {
    const int n_w = 2 * p.n_layer * 17;      // (64 * 64 + 64) / 256 = 16.25 -> 17 blocks of 256 entries per gradient block
    if ((int)blockIdx.x < n_w) {
        const int g = blockIdx.x / 17, idx = (blockIdx.x % 17) * 256 + threadIdx.x;
        if (idx >= 64 * 64 + 64) return;
        float s = 0.f;
        for (int c = 0; c < RPB_CHUNKS; ++c) s += p.part[((size_t)c * 2 * p.n_layer + g) * (64 * 64 + 64) + idx];
        const int layer = g >> 1, which = g & 1;
        if (idx < 64 * 64)
            (which == 0 ? p.gw2 : p.gw0)[(size_t)layer * 64 * 64 + idx] = s;
        else
            (which == 0 ? p.gb2 : p.gb0)[(size_t)layer * 64 + idx - 64 * 64] = s;
        return;
    }
    const long long i4 = (long long)(blockIdx.x - n_w) * 256 + threadIdx.x;      // float4 index into gx
    if (i4 >= p.rows * 16) return;
    float4 s = *reinterpret_cast<const float4 *>(p.gxl + i4 * 4);
    for (int l = 1; l < p.n_layer; ++l) {
        const float4 a = *reinterpret_cast<const float4 *>(p.gxl + ((size_t)l * p.rows * 64) + i4 * 4);
        s.x += a.x, s.y += a.y, s.z += a.z, s.w += a.w;
    }
    *reinterpret_cast<float4 *>(p.gx + i4 * 4) = s;
}

}  // namespace ultra

using namespace ultra;

extern "C" {

int64_t ultra_relation_projection_backward_workspace(int64_t rows, int32_t n_layer) {
    if (rows < 0 || n_layer < 0) return 0;
    return ((int64_t)3 * n_layer * rows * 64 + (int64_t)RPB_CHUNKS * 2 * n_layer * (64 * 64 + 64)) * (int64_t)sizeof(float);
}

int32_t ultra_relation_projection_backward(const void *x, const void *const *w0, const void *const *b0, const void *const *w2,
                                           const void *const *grad_out, void *grad_x, void *grad_w0, void *grad_b0, void *grad_w2,
                                           void *grad_b2, void *workspace, int64_t workspace_bytes, int64_t rows, int32_t n_layer,
                                           int32_t dim, void *stream) {
    ULTRA_DEVICE_SCOPE(stream, x);
    if (dim != 64 || n_layer > RPB_MAX_LAYERS) {
        set_error("ultra_relation_projection_backward: only dim = 64 and at most 8 layers are built (the ULTRA checkpoints' shape)");
        return ULTRA_ERR_UNSUPPORTED;
    }
    if (!x || !w0 || !b0 || !w2 || !grad_out || !grad_x || !grad_w0 || !grad_b0 || !grad_w2 || !grad_b2 || !workspace || rows <= 0 ||
        n_layer <= 0) {
        set_error("ultra_relation_projection_backward: NULL operand (or no rows / layers)");
        return ULTRA_ERR_INVALID;
    }
    if (workspace_bytes < ultra_relation_projection_backward_workspace(rows, n_layer)) {
        set_error("ultra_relation_projection_backward: workspace smaller than ultra_relation_projection_backward_workspace()");
        return ULTRA_ERR_INVALID;
    }
    RelProjBwdParams p;
    p.x = (const float *)x;
    for (int l = 0; l < n_layer; ++l) {
        if (!w0[l] || !b0[l] || !w2[l] || !grad_out[l]) {
            set_error("ultra_relation_projection_backward: NULL layer operand");
            return ULTRA_ERR_INVALID;
        }
        p.w0[l] = (const float *)w0[l], p.b0[l] = (const float *)b0[l], p.w2[l] = (const float *)w2[l];
        p.gout[l] = (const float *)grad_out[l];
    }
    float *ws = (float *)workspace;
    const size_t act = (size_t)n_layer * rows * 64;
    p.h = ws, p.gh = ws + act, p.gxl = ws + 2 * act, p.part = ws + 3 * act;
    p.gx = (float *)grad_x;
    p.gw0 = (float *)grad_w0, p.gb0 = (float *)grad_b0, p.gw2 = (float *)grad_w2, p.gb2 = (float *)grad_b2;
    p.rows = rows;
    p.n_layer = n_layer;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    hipLaunchKernelGGL(relproj_bwd_rows_kernel, dim3((unsigned)((rows + 16 * RPB_TILES - 1) / (16 * RPB_TILES)), (unsigned)n_layer),
                       dim3(256), 0, s, p);
    hipLaunchKernelGGL(relproj_bwd_weights_kernel, dim3(RPB_CHUNKS, (unsigned)(2 * n_layer)), dim3(256), 0, s, p);
    const unsigned n_red = (unsigned)(2 * n_layer * 17 + (rows * 16 + 255) / 256);
    hipLaunchKernelGGL(relproj_bwd_reduce_kernel, dim3(n_red), dim3(256), 0, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error(std::string("relation projection backward launch: ") + hipGetErrorString(e));
        return ULTRA_ERR_HIP;
    }
    return ULTRA_OK;
}

}  // extern "C"
