// Stand-alone microbenchmark (no torch): the chain body of dense_order_layer_kernel in isolation -- per source column one
// adjacency byte -> float, one product rel * x, one dependent v_mfma_f32_16x16x4_f32, optionally one ds_bpermute.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_chain_probe.hip -o /tmp/chain_lat && /tmp/chain_lat
// MI355X result (cycles per column): matrix instruction alone 33.6, + cvt 34.1, + cvt + mul 35.8, + bpermute 43.0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
using f32x4 = float __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float byte_of(uint32_t w, int q) { return (float)((w >> (8 * q)) & 0xffu); }
// MODE 0: mfma only (operands fixed); 1: + cvt of adjacency byte; 2: + mul; 3: + bpermute one group ahead (as in the layer kernel)
template <int MODE>
__global__ void k(float *out, long long *cyc, int n, const uint32_t *aw_in, const float *x_in) {
    f32x4 acc = {0, 0, 0, 0};
    const int lane = threadIdx.x & 63, i16 = lane & 15;
    uint32_t aw[16];
    float x[16];
    for (int i = 0; i < 16; ++i) aw[i] = aw_in[(lane + i) & 63], x[i] = x_in[(lane * 16 + i) & 1023];
    const float relv = x_in[lane];
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it) {
        float bv[4], bn[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = MODE >= 3 ? __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (16 * r + i16), __float_as_int(x[0]))) : x[r];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (q + 1 < 16) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    bn[r] = MODE >= 3 ? __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (16 * r + i16), __float_as_int(x[q + 1]))) : x[(q + 1 + r) & 15];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 4 * q + r;
                const float a = MODE >= 1 ? byte_of(aw[c >> 2], c & 3) : relv;
                const float b = MODE >= 2 ? relv * bv[r] : bv[r];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[r] = bn[r];
        }
        for (int i = 0; i < 16; ++i) x[i] += 1.0f;   // keep the operands live and changing
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    float *out, *x; long long *cyc; uint32_t *aw;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8); hipMalloc(&aw, 256); hipMalloc(&x, 4096);
    hipMemset(aw, 1, 256); hipMemset(x, 0, 4096);
    const int n = 512;
    const char *names[] = {"mfma only", "+ cvt", "+ cvt + mul", "+ cvt + mul + bpermute"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(240), dim3(256), 0, 0, out, cyc, n, aw, x); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(240), dim3(256), 0, 0, out, cyc, n, aw, x); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(240), dim3(256), 0, 0, out, cyc, n, aw, x); break;
                default: hipLaunchKernelGGL(k<3>, dim3(240), dim3(256), 0, 0, out, cyc, n, aw, x); break;
            }
            hipDeviceSynchronize();
        }
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-28s %.1f cycles per MFMA\n", names[mode], (double)c / (n * 64));
    }
    return 0;
}
