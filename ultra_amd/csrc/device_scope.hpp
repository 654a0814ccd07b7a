// Every launching C entry point runs on the device that owns its operands, whatever the calling thread's current
// device is -- like the reference's cudaSetDevice(input.get_device()) (rspmm.cu:243, 304).  One process per GPU under
// torchrun usually never calls torch.cuda.set_device, so rank k would otherwise launch on device 0 with device-k
// pointers.  A non-NULL stream names its device; the NULL stream (torch's default stream on EVERY device has the handle 0)
// names nothing, so then the device is read off one of the entry's device pointers (hipPointerGetAttributes).
#pragma once

#include <hip/hip_runtime.h>

namespace ultra {

struct DeviceScope {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceScope(hipStream_t s, const void *operand = nullptr) {
        // (no device at all: argument validation still runs and reports; a launch would fail on its own)
        if (hipGetDevice(&prev) != hipSuccess) {
            (void)hipGetLastError();
            return;
        }
        int dev = prev;
        bool known = s != nullptr && hipStreamGetDevice(s, &dev) == hipSuccess;
        if (!known && operand != nullptr) {
            hipPointerAttribute_t attr;
            if (hipPointerGetAttributes(&attr, operand) == hipSuccess && attr.type == hipMemoryTypeDevice)
                dev = attr.device, known = true;
            else
                (void)hipGetLastError();   // (a host or unknown pointer: argument validation reports it)
        }
        if (known && dev != prev) {
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

#define ULTRA_DEVICE_SCOPE(stream_void_ptr, operand_dev_ptr)                                        \
    ::ultra::DeviceScope _ultra_scope(reinterpret_cast<hipStream_t>(stream_void_ptr), operand_dev_ptr); \
    if (_ultra_scope.err != hipSuccess) {                                          \
        ::ultra::set_error("could not select the operands' device");                \
        return ULTRA_ERR_HIP;                                                      \
    }
