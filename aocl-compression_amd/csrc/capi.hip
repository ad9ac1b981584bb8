// csrc/capi.hip -- device management, compaction, timing and host-buffer entry points of the C-ABI
// (include/aocl_gpu_codec.h).  Codec launchers live next to their kernels (lz4_kernels.hip, ...).
#include "common.hip.h"
#include "host_util.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace aoclgpu {

static thread_local char g_err[256] = "";

void set_error(const char* what, hipError_t e)
{
    std::snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
}
void set_error_msg(const char* msg) { std::snprintf(g_err, sizeof(g_err), "%s", msg); }

int check_batch(const aoclgpu_batch* b, bool decompress)
{
    if (!b) { set_error_msg("batch is NULL"); return AOCLGPU_ERR_ARG; }
    if (b->nchunks == 0) return AOCLGPU_OK;
    if (!b->dst || !b->out_len || (!b->src && b->total_bytes != 0 && !b->src_off)) {
        set_error_msg("batch has NULL src/dst/out_len"); return AOCLGPU_ERR_ARG;
    }
    if (b->src_off) {
        if (!b->src_len || !b->dst_off || !b->dst_cap) { set_error_msg("explicit batch misses an array"); return AOCLGPU_ERR_ARG; }
    } else {
        if (b->chunk_bytes == 0 || b->chunk_bytes > 0xFFFFFFFFull) { set_error_msg("uniform batch: bad chunk_bytes"); return AOCLGPU_ERR_ARG; }
        if (decompress && !b->src_len) { set_error_msg("uniform decompress batch needs src_len[]"); return AOCLGPU_ERR_ARG; }
        const uint64_t need = (b->total_bytes + b->chunk_bytes - 1) / b->chunk_bytes;
        if (b->nchunks != (need ? need : 1)) { set_error_msg("uniform batch: nchunks != ceil(total/chunk)"); return AOCLGPU_ERR_ARG; }
    }
    return AOCLGPU_OK;
}

int check_launch(const char* kernel)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error(kernel, e); return AOCLGPU_ERR_HIP; }
    return AOCLGPU_OK;
}

// ---------------- compaction: exclusive scan of sizes + gather ----------------
__global__ void __launch_bounds__(1024) scan_sizes_kernel(const int32_t* __restrict__ len, uint32_t n, uint64_t* __restrict__ off)
{
    __shared__ uint64_t part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t lo = t * per, hi = (lo + per < n) ? lo + per : n;
    uint64_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += (len[i] > 0) ? (uint64_t)len[i] : 0;
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint64_t v = (t >= d) ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint64_t run = (t == 0) ? 0 : part[t - 1];
    for (uint32_t i = lo; i < hi; ++i) { off[i] = run; run += (len[i] > 0) ? (uint64_t)len[i] : 0; }
    if (t == 1023) off[n] = part[1023];
}

__global__ void __launch_bounds__(256) gather_slots_kernel(const uint8_t* __restrict__ slots, uint64_t slot_bytes,
                                                           const int32_t* __restrict__ len, const uint64_t* __restrict__ off,
                                                           uint8_t* __restrict__ packed)
{
    const uint32_t c = blockIdx.x;
    const int32_t l = len[c];
    if (l <= 0) return;
    const uint8_t* s = slots + (uint64_t)c * slot_bytes;
    uint8_t* d = packed + off[c];
    uint32_t n = (uint32_t)l;
    const uint32_t head = (uint32_t)((16u - ((uintptr_t)d & 15u)) & 15u);
    const uint32_t h = head < n ? head : n;
    if (threadIdx.x < h) d[threadIdx.x] = s[threadIdx.x];
    d += h; s += h; n -= h;
    const uint32_t nvec = n >> 4;
    for (uint32_t i = threadIdx.x; i < nvec; i += 256) {
        uint4 v; __builtin_memcpy(&v, s + ((size_t)i << 4), 16);
        *reinterpret_cast<uint4*>(d + ((size_t)i << 4)) = v;
    }
    const uint32_t done = nvec << 4;
    if (threadIdx.x < n - done) d[done + threadIdx.x] = s[done + threadIdx.x];
}

namespace {
// Some launches need per-resident-workgroup scratch in HBM (Zstd compressor: sequence store + staged block; Zstd decoder:
// literal buffer; Snappy compressor: hash tables).  One pool entry per (device, stream): launches on the same stream are ordered, launches on different
// streams never share scratch.  A pool that has to grow is freed first (hipFree waits for the device).
struct WsPool { void* p = nullptr; size_t bytes = 0; int dev = -1; void* stream = nullptr; };
WsPool g_pool[16];
std::mutex g_pool_mu;

}  // namespace

void* get_workspace(size_t bytes, void* stream)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    WsPool* e = nullptr;
    for (WsPool& q : g_pool) if (q.p && q.dev == dev && q.stream == stream) { e = &q; break; }
    if (!e) for (WsPool& q : g_pool) if (!q.p) { e = &q; break; }
    if (!e) {                                             // more than 16 (device, stream) pairs: recycle the first entry
        if (hipDeviceSynchronize() != hipSuccess) return nullptr;
        e = &g_pool[0];
    }
    if (e->p && e->dev == dev && e->stream == stream && e->bytes >= bytes) return e->p;
    if (e->p) { (void)hipFree(e->p); e->p = nullptr; }
    if (hipMalloc(&e->p, bytes) != hipSuccess) { e->p = nullptr; return nullptr; }
    e->bytes = bytes; e->dev = dev; e->stream = stream;
    return e->p;
}

}  // namespace aoclgpu

using namespace aoclgpu;

extern "C" const char* aoclgpu_last_error(void) { return g_err; }
extern "C" const char* aoclgpu_version(void) { return "aocl-compression-gpu 0.1 (gfx950)"; }

extern "C" int aoclgpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int aoclgpu_init(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { set_error("hipGetDeviceCount", e); return AOCLGPU_ERR_NO_DEVICE; }
    if (device < 0 || device >= n) { set_error_msg("device index out of range"); return AOCLGPU_ERR_ARG; }
    e = hipSetDevice(device);
    if (e != hipSuccess) { set_error("hipSetDevice", e); return AOCLGPU_ERR_HIP; }
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, device);
    if (e != hipSuccess) { set_error("hipGetDeviceProperties", e); return AOCLGPU_ERR_HIP; }
    if (std::strncmp(p.gcnArchName, "gfx950", 6) != 0) {
        std::snprintf(g_err, sizeof(g_err), "device %d is %s, this library is built for gfx950 only", device, p.gcnArchName);
        return AOCLGPU_ERR_NO_DEVICE;
    }
    return AOCLGPU_OK;
}

extern "C" size_t aoclgpu_compress_bound(aoclgpu_codec codec, size_t n)
{
    switch (codec) {
    case AOCLGPU_LZ4:    return n > 0x7E000000u ? 0 : n + n / 255 + 16;                      /* lz4.h:261 */
    case AOCLGPU_SNAPPY: return 32 + n + n / 6;                                               /* snappy.cc:160-182 */
    case AOCLGPU_ZSTD:   return n + (n >> 8) + (n < (128u << 10) ? (((128u << 10) - n) >> 11) : 0);  /* zstd.h:473 */
    }
    return 0;
}

extern "C" int aoclgpu_compact_slots(const uint8_t* slots, uint64_t slot_bytes, const int32_t* out_len, uint32_t nchunks,
                                     uint8_t* packed, uint64_t* out_off, void* stream)
{
    if (!slots || !out_len || !packed || !out_off) { set_error_msg("compact: NULL argument"); return AOCLGPU_ERR_ARG; }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(scan_sizes_kernel, dim3(1), dim3(1024), 0, s, out_len, nchunks, out_off);
    if (int rc = check_launch("scan_sizes_kernel")) return rc;
    if (nchunks) hipLaunchKernelGGL(gather_slots_kernel, dim3(nchunks), dim3(256), 0, s, slots, slot_bytes, out_len, out_off, packed);
    return check_launch("gather_slots_kernel");
}

typedef int (*batch_fn)(const aoclgpu_batch*, void*);
static batch_fn pick(aoclgpu_codec codec, int op)
{
    switch (codec) {
    case AOCLGPU_LZ4:    return op ? aoclgpu_lz4_decompress_chunks : aoclgpu_lz4_compress_chunks;
    case AOCLGPU_SNAPPY: return op ? aoclgpu_snappy_decompress_chunks : aoclgpu_snappy_compress_chunks;
    case AOCLGPU_ZSTD:   return op ? aoclgpu_zstd_decompress_chunks : aoclgpu_zstd_compress_chunks;
    }
    return nullptr;
}

extern "C" double aoclgpu_time_batch(aoclgpu_codec codec, int op, const aoclgpu_batch* b, int iters, void* stream)
{
    batch_fn fn = pick(codec, op);
    if (!fn || iters <= 0) { set_error_msg("time_batch: bad codec/iters"); return -1.0; }
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { set_error_msg("hipEventCreate failed"); return -1.0; }
    double ms = -1.0;
    hipEventRecord(e0, s);
    int rc = 0;
    for (int i = 0; i < iters && rc == 0; ++i) rc = fn(b, stream);
    hipEventRecord(e1, s);
    hipError_t e = hipEventSynchronize(e1);
    if (rc == 0 && e == hipSuccess) {
        float f = 0.f;
        if (hipEventElapsedTime(&f, e0, e1) == hipSuccess) ms = (double)f / iters;
    } else if (e != hipSuccess) set_error("hipEventSynchronize", e);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms;
}

// ---------------- host-buffer one-shot path ----------------
static int64_t host_oneshot(aoclgpu_codec codec, int op, const void* hsrc, size_t n, void* hdst, size_t cap, uint32_t flags = 0)
{
    batch_fn fn = pick(codec, op);
    if (!fn) { set_error_msg("bad codec"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_ARG; }
    if ((!hsrc && n) || !hdst || n > 0xFFFFFFFFull) { set_error_msg("host one-shot: bad arguments"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_ARG; }
    if (cap > 0xFFFFFFFFull) cap = 0xFFFFFFFFull;
    uint8_t *dsrc = nullptr, *ddst = nullptr; int32_t* dlen = nullptr; uint32_t* dsl = nullptr;
    int64_t result = AOCLGPU_HOST_FAIL + AOCLGPU_ERR_HIP;
    hipError_t e;
#define CK(call) do { e = (call); if (e != hipSuccess) { set_error(#call, e); goto done; } } while (0)
    CK(hipMalloc((void**)&dsrc, n ? n : 1));
    CK(hipMalloc((void**)&ddst, cap ? cap : 1));
    CK(hipMalloc((void**)&dlen, sizeof(int32_t)));
    CK(hipMalloc((void**)&dsl, sizeof(uint32_t)));
    if (n) CK(hipMemcpy(dsrc, hsrc, n, hipMemcpyHostToDevice));
    {
        aoclgpu_batch b; std::memset(&b, 0, sizeof(b));
        b.src = dsrc; b.dst = ddst; b.out_len = dlen; b.nchunks = 1;
        if (const char* ex = getenv("AOCLGPU_EXPERIMENT")) b.reserved = (uint32_t)atoi(ex) & 0xFFu;   // experiment switches of the test tools
        b.reserved |= flags & ~0xFFu;
        if (op == 0) { b.total_bytes = n; b.chunk_bytes = n ? n : 1; b.slot_bytes = cap; }
        else {
            uint32_t sl = (uint32_t)n;
            CK(hipMemcpy(dsl, &sl, sizeof(sl), hipMemcpyHostToDevice));
            b.src_len = dsl; b.total_bytes = cap; b.chunk_bytes = cap ? cap : 1; b.slot_bytes = n;
        }
        int rc = fn(&b, nullptr);
        if (rc != 0) { result = AOCLGPU_HOST_FAIL + rc; goto done; }
        CK(hipDeviceSynchronize());
        int32_t r = 0;
        CK(hipMemcpy(&r, dlen, sizeof(r), hipMemcpyDeviceToHost));
        if (r > 0) CK(hipMemcpy(hdst, ddst, (size_t)r, hipMemcpyDeviceToHost));
        result = r;
    }
#undef CK
done:
    if (dsrc) hipFree(dsrc);
    if (ddst) hipFree(ddst);
    if (dlen) hipFree(dlen);
    if (dsl) hipFree(dsl);
    return result;
}

extern "C" int64_t aoclgpu_compress_host_ex(aoclgpu_codec codec, const void* host_src, size_t n, void* host_dst, size_t cap, uint32_t flags)
{
    return host_oneshot(codec, 0, host_src, n, host_dst, cap, flags);
}
extern "C" int64_t aoclgpu_compress_host(aoclgpu_codec codec, const void* host_src, size_t n, void* host_dst, size_t cap)
{
    return host_oneshot(codec, 0, host_src, n, host_dst, cap);
}
extern "C" int64_t aoclgpu_decompress_host(aoclgpu_codec codec, const void* host_src, size_t n, void* host_dst, size_t cap)
{
    return host_oneshot(codec, 1, host_src, n, host_dst, cap);
}
