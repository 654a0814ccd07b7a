// Measurement helper: a plain 16-byte-per-lane streaming copy used to (a) measure the achievable HBM
// copy ceiling next to the rspmm numbers and (b) calibrate the FETCH_SIZE / WRITE_SIZE counters on a
// known byte count (MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced reads on gfx950).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "../../include/ultra_rspmm.h"
#include "plan.hpp"

namespace ultra {

__global__ void __launch_bounds__(256) stream_copy_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst,
                                                          long long n16) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

}  // namespace ultra

extern "C" int32_t ultra_stream_copy(void *dst, const void *src, int64_t bytes, void *stream) {
    if (!dst || !src || bytes < 0 || (bytes & 15)) {
        ultra::set_error("ultra_stream_copy: NULL pointer or size not a multiple of 16");
        return ULTRA_ERR_INVALID;
    }
    if (bytes == 0) return ULTRA_OK;
    hipLaunchKernelGGL(ultra::stream_copy_kernel, dim3(2048), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       (const float4 *)src, (float4 *)dst, (long long)(bytes / 16));
    if (hipGetLastError() != hipSuccess) {
        ultra::set_error("stream_copy_kernel launch failed");
        return ULTRA_ERR_HIP;
    }
    return ULTRA_OK;
}
