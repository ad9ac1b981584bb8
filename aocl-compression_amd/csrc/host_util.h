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
size_t workspace_held(const WsGuard& held, void* stream);   // bytes the pool already holds for this (device, stream)
// The host-buffer entry points work on aoclgpu_host_device(); the calling thread's current device is put back when they
// return (a caller that drives another GPU with its own HIP code must not find it changed under its feet).
struct DeviceScope {
    int prev = -1; bool ok = false;
    explicit DeviceScope(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; ok = hipSetDevice(dev) == hipSuccess; if (prev == dev) prev = -1; }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope&) = delete; DeviceScope& operator=(const DeviceScope&) = delete;
};
// One block of an LZ4 stream (csrc/lz4_kernels.hip lz4_compress_stream_kernel, driven by csrc/lz4frame_api.cpp): bytes
// [off, off + len) of the stream window; see Lz4Link (lz4_kernels.hip) for lowCand / lowPos / ext.
struct Lz4StreamJob { uint32_t off, len, lowCand, lowPos, ext, pad[3]; };
int lz4_stream_blocks(const uint8_t* d_src, uint32_t idxBase, const Lz4StreamJob* d_jobs, uint32_t njobs, uint32_t* d_table, uint8_t* d_dst, uint32_t slot,
                      int32_t* d_len, uint32_t accel, void* stream);
}  // namespace aoclgpu
