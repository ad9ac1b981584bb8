// csrc/host_util.h -- host-side helpers shared by the C-ABI launchers.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/aocl_gpu_codec.h"

namespace aoclgpu {
void set_error(const char* what, hipError_t e);
void set_error_msg(const char* msg);
int check_batch(const aoclgpu_batch* b, bool decompress);
int check_launch(const char* kernel);
// Scratch in HBM for the launch about to be enqueued on `stream` (one pool entry per device and stream); nullptr on failure.
// The caller holds a WsGuard from before this call until its kernel is enqueued (see capi.hip).
struct WsGuard { WsGuard(); ~WsGuard(); WsGuard(const WsGuard&) = delete; WsGuard& operator=(const WsGuard&) = delete; };
void* get_workspace(const WsGuard& held, size_t bytes, void* stream);
// The host-buffer entry points work on aoclgpu_host_device(); the calling thread's current device is put back when they
// return (a caller that drives another GPU with its own HIP code must not find it changed under its feet).
struct DeviceScope {
    int prev = -1; bool ok = false;
    explicit DeviceScope(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; ok = hipSetDevice(dev) == hipSuccess; if (prev == dev) prev = -1; }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope&) = delete; DeviceScope& operator=(const DeviceScope&) = delete;
};
}  // namespace aoclgpu
