// Every launching C entry point runs on the device that OWNS ITS STREAM (NULL stream: the calling thread's current
// device), whatever the thread's current device is -- like the reference's cudaSetDevice(input.get_device())
// (rspmm.cu:243, 304).  One process per GPU under torchrun usually never calls torch.cuda.set_device, so rank k would
// otherwise launch on device 0 with device-k pointers.
#pragma once

#include <hip/hip_runtime.h>

namespace ultra {

struct DeviceScope {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceScope(hipStream_t s) {
        // (no device at all: argument validation still runs and reports; a launch would fail on its own)
        if (hipGetDevice(&prev) != hipSuccess) {
            (void)hipGetLastError();
            return;
        }
        int dev = prev;
        if (s != nullptr && hipStreamGetDevice(s, &dev) == hipSuccess && dev != prev) {
            err = hipSetDevice(dev);
            switched = err == hipSuccess;
        }
    }
    ~DeviceScope() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};

}  // namespace ultra

#define ULTRA_DEVICE_SCOPE(stream_void_ptr)                                        \
    ::ultra::DeviceScope _ultra_scope(reinterpret_cast<hipStream_t>(stream_void_ptr)); \
    if (_ultra_scope.err != hipSuccess) {                                          \
        ::ultra::set_error("could not select the stream's device");                \
        return ULTRA_ERR_HIP;                                                      \
    }
