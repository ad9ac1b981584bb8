// csrc/common.hip.h -- shared device helpers for the gfx950 block-codec kernels.
//
// Execution model used by every codec kernel in this directory:
//   * one 64-lane wavefront owns one chunk (workgroup = 1 wave, so there is no s_barrier anywhere);
//   * the codec's hash table lives in that workgroup's LDS;
//   * wave-uniform parser state (ip / op / anchor ...) is kept uniform by construction and hoisted
//     into SGPRs with readfirstlane where the compiler cannot prove it;
//   * cross-lane decisions use ballot + ctz, cross-lane data uses readlane / LDS.
// LDS accesses of one wave execute in program order, so cross-lane LDS hand-offs only need a
// compiler-level fence (wavefront scope lowers to no instruction).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/aocl_gpu_codec.h"

namespace aoclgpu {

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ uint32_t ld8(const uint8_t* p) { return *p; }
__device__ __forceinline__ uint32_t ld16(const uint8_t* p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }
__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint64_t ld64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ void st16(uint8_t* p, uint32_t v) { uint16_t w = (uint16_t)v; __builtin_memcpy(p, &w, 2); }
__device__ __forceinline__ void st32(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }

// wave-uniform broadcast of lane 0's value (also tells the compiler the value is an SGPR)
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t lane_get(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }

// compiler-only ordering point between cross-lane LDS / global accesses of ONE wave
__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

__device__ __forceinline__ uint64_t ballot(bool p) { return __ballot(p); }
__device__ __forceinline__ int first_lane(uint64_t m) { return m ? (int)__builtin_ctzll(m) : 64; }

// Chunk geometry of a batch (see aoclgpu_batch in include/aocl_gpu_codec.h)
struct ChunkView {
    const uint8_t* src; uint32_t src_len;
    uint8_t* dst; uint32_t dst_cap;
};

__device__ __forceinline__ ChunkView compress_view(const aoclgpu_batch& b, uint32_t i)
{
    ChunkView v;
    if (b.src_off) {
        v.src = b.src + b.src_off[i]; v.src_len = b.src_len[i];
        v.dst = b.dst + b.dst_off[i]; v.dst_cap = b.dst_cap[i];
    } else {
        const uint64_t o = (uint64_t)i * b.chunk_bytes;
        const uint64_t rem = b.total_bytes > o ? b.total_bytes - o : 0;
        v.src = b.src + o; v.src_len = (uint32_t)(rem < b.chunk_bytes ? rem : b.chunk_bytes);
        v.dst = b.dst + (uint64_t)i * b.slot_bytes;
        v.dst_cap = (uint32_t)(b.slot_bytes > 0xFFFFFFFFull ? 0xFFFFFFFFull : b.slot_bytes);
    }
    return v;
}

__device__ __forceinline__ ChunkView decompress_view(const aoclgpu_batch& b, uint32_t i)
{
    ChunkView v;
    if (b.src_off) {
        v.src = b.src + b.src_off[i]; v.src_len = b.src_len[i];
        v.dst = b.dst + b.dst_off[i]; v.dst_cap = b.dst_cap[i];
    } else {
        const uint64_t o = (uint64_t)i * b.chunk_bytes;
        const uint64_t rem = b.total_bytes > o ? b.total_bytes - o : 0;
        v.src = b.src + (uint64_t)i * b.slot_bytes; v.src_len = b.src_len[i];
        v.dst = b.dst + o; v.dst_cap = (uint32_t)(rem < b.chunk_bytes ? rem : b.chunk_bytes);
    }
    return v;
}

// Cooperative forward copy of n bytes, non-overlapping, arbitrary alignment (global -> global).
// All 64 lanes call it with uniform arguments.
__device__ __forceinline__ void wave_copy(uint8_t* __restrict__ d, const uint8_t* __restrict__ s, uint32_t n, int lane)
{
    if (n <= 64) {
        if ((uint32_t)lane < n) d[lane] = s[lane];
        return;
    }
    // head: bring the destination to 16-byte alignment (<= 15 bytes)
    const uint32_t head = (uint32_t)((16u - ((uintptr_t)d & 15u)) & 15u);
    if ((uint32_t)lane < head) d[lane] = s[lane];
    d += head; s += head; n -= head;
    const uint32_t nvec = n >> 4;
    for (uint32_t i = (uint32_t)lane; i < nvec; i += kWave) {
        uint4 v; __builtin_memcpy(&v, s + ((size_t)i << 4), 16);
        *reinterpret_cast<uint4*>(d + ((size_t)i << 4)) = v;
    }
    const uint32_t done = nvec << 4;
    if ((uint32_t)lane < n - done) d[done + lane] = s[done + lane];
}

}  // namespace aoclgpu
