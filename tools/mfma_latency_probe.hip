// Stand-alone microbenchmark (no torch): issue-to-issue distance of DEPENDENT fp32 matrix instructions on gfx950.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_latency_probe.hip -o /tmp/mfma_lat && /tmp/mfma_lat
// MI355X result: v_mfma_f32_16x16x4_f32 32 cycles, v_mfma_f32_32x32x2_f32 64 cycles, whether the accumulator chain is
// dependent or two chains alternate, with or without a VALU-produced operand; 8 waves per CU (2 per SIMD) share the pipe.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = float __attribute__((ext_vector_type(4)));
using f32x16 = float __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ void k(float *out, long long *cyc, int n, float a0, float b0) {
    f32x4 acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
    f32x16 big = {0}, big2 = {0};
    float a = a0 + threadIdx.x * 1e-9f, b = b0;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc2, 0, 0, 0);
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 16; ++u) big = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, big, 0, 0, 0);
        } else if (MODE == 3) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                big = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, big, 0, 0, 0);
                big2 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, big2, 0, 0, 0);
            }
        } else if (MODE == 4) {   // dependent chain with a VALU-produced B operand each step (like the layer kernel)
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                float bb = b * (a + (float)u);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bb, acc, 0, 0, 0);
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = acc[0] + acc2[1] + big[0] + big2[3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    float *out; long long *cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
    const int n = 1024;
    const char *names[] = {"16x16x4 dependent", "16x16x4 two chains", "32x32x2 dependent", "32x32x2 two chains", "16x16x4 dependent + VALU operand"};
    for (int waves = 1; waves <= 8; waves *= 2)
    for (int mode = 0; mode < 5; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, n, 1.f, 2.f); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, n, 1.f, 2.f); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, n, 1.f, 2.f); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, n, 1.f, 2.f); break;
                default: hipLaunchKernelGGL(k<4>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, n, 1.f, 2.f); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("waves/WG %d  %-34s  %.1f us  -> %.1f ns per MFMA per wave (%.1f counter ticks)\n", waves, names[mode], ms * 1e3, ms * 1e6 / (n * 16), (double)c / (n * 16));
    }
    return 0;
}
