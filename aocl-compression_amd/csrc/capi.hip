// csrc/capi.hip -- device management, compaction, timing and host-buffer entry points of the C-ABI
// (include/aocl_gpu_codec.h).  Codec launchers live next to their kernels (lz4_kernels.hip, ...).
#include "common.hip.h"
#include "host_util.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <new>
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
// literal buffer; Snappy compressor: hash tables).  One pool entry per (device, stream): launches on the same stream are
// ordered, launches on different streams never share scratch.
//
// Thread safety: a launcher takes a WsGuard (one process-wide mutex) BEFORE it asks for scratch and keeps it until its
// kernel is enqueued.  An entry is only ever freed under that mutex, and hipFree waits for everything enqueued on the
// device, so (a) no thread can free a pointer another thread has been handed but not yet launched on, and (b) a buffer
// that has to grow is released only after the kernels that use it have finished.
struct WsPool { void* p = nullptr; size_t bytes = 0; int dev = -1; void* stream = nullptr; };
WsPool g_pool[16];
std::mutex g_pool_mu;

}  // namespace

WsGuard::WsGuard() { g_pool_mu.lock(); }
WsGuard::~WsGuard() { g_pool_mu.unlock(); }

void* get_workspace(const WsGuard&, size_t bytes, void* stream)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    WsPool* e = nullptr;
    for (WsPool& q : g_pool) if (q.p && q.dev == dev && q.stream == stream) { e = &q; break; }
    if (!e) for (WsPool& q : g_pool) if (!q.p) { e = &q; break; }
    if (!e) {                                             // more than 16 (device, stream) pairs: recycle the smallest entry
        e = &g_pool[0];
        for (WsPool& q : g_pool) if (q.bytes < e->bytes) e = &q;
    }
    if (e->p && e->dev == dev && e->stream == stream && e->bytes >= bytes) return e->p;
    if (e->p) {
        // the entry may belong to another device: free it there (hipFree synchronises that device first)
        int cur = dev;
        if (e->dev != dev) (void)hipSetDevice(e->dev);
        (void)hipFree(e->p);
        if (e->dev != dev) (void)hipSetDevice(cur);
        e->p = nullptr; e->bytes = 0;
    }
    size_t ask = bytes;
    if (const char* f = getenv("AOCLGPU_FAULT_WORKSPACE_MB")) {   // test hook: requests above this size fail the way a full device fails them
        const long long mb = atoll(f);
        if (mb > 0 && bytes > ((size_t)mb << 20)) ask = (size_t)1 << 60;
    }
    if (hipMalloc(&e->p, ask) != hipSuccess) {
        (void)hipGetLastError();                          // the failure must not surface as the next launch's error (callers fall back)
        e->p = nullptr; return nullptr;
    }
    e->bytes = bytes; e->dev = dev; e->stream = stream;
    return e->p;
}

size_t workspace_held(const WsGuard&, void* stream)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    for (WsPool& q : g_pool) if (q.p && q.dev == dev && q.stream == stream) return q.bytes;
    return 0;
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

// ---------------- memory-system probes (bench.py: the measured ceiling next to the 8 TB/s of the data sheet) ----------------
// probe_copy_kernel: every lane moves 16 bytes per step, consecutive lanes consecutive addresses - the access pattern HBM
// likes best; 2 * bytes / time is what a kernel of this chip can get out of its memory.
// probe_scatter8_kernel: every lane issues dependent-free 8-byte loads at pseudo-random 8-byte-aligned addresses of a large
// buffer (the codecs' candidate-window pattern); run under rocprofv3 --pmc FETCH_SIZE it calibrates that counter for
// scattered loads (bytes fetched per load).
// Variants of the copy (the probe reports the fastest: it asks what the memory system can do, not what one loop shape does):
// TILE = 16-byte loads in flight per lane; NT = non-temporal loads and stores (streamed data is not kept in L2 / MALL);
// BLOCKED = a workgroup walks whole tiles of 256 x TILE x 16 bytes (each wave instruction covers 1 KiB of consecutive bytes,
// a workgroup 4 KiB x TILE) instead of a grid-wide stride between a lane's loads.
typedef uint32_t __attribute__((ext_vector_type(4))) probe_u4;
template <int TILE, bool NT, bool BLOCKED>
__global__ void __launch_bounds__(256) probe_copy_kernel(probe_u4* __restrict__ dst, const probe_u4* __restrict__ src, size_t n16)
{
    if (BLOCKED) {
        const size_t per = (size_t)256 * TILE, ntiles = n16 / per;
        for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
            const size_t o = t * per + threadIdx.x;
            probe_u4 v[TILE];
#pragma unroll
            for (int k = 0; k < TILE; ++k) v[k] = NT ? __builtin_nontemporal_load(src + o + (size_t)k * 256) : src[o + (size_t)k * 256];
#pragma unroll
            for (int k = 0; k < TILE; ++k) { if (NT) __builtin_nontemporal_store(v[k], dst + o + (size_t)k * 256); else dst[o + (size_t)k * 256] = v[k]; }
        }
        for (size_t i = ntiles * per + (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
    } else {
        const size_t stride = (size_t)gridDim.x * blockDim.x;
        size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        for (; i + (TILE - 1) * stride < n16; i += TILE * stride) {
            probe_u4 v[TILE];
#pragma unroll
            for (int k = 0; k < TILE; ++k) v[k] = NT ? __builtin_nontemporal_load(src + i + k * stride) : src[i + k * stride];
#pragma unroll
            for (int k = 0; k < TILE; ++k) { if (NT) __builtin_nontemporal_store(v[k], dst + i + k * stride); else dst[i + k * stride] = v[k]; }
        }
        for (; i < n16; i += stride) dst[i] = src[i];
    }
}
typedef void (*probe_copy_fn)(probe_u4*, const probe_u4*, size_t);
struct ProbeCopyVariant { const char* name; probe_copy_fn fn; int wg_per_cu; };
const ProbeCopyVariant kProbeCopy[] = {
    {"stride x4, 8 WG/CU", probe_copy_kernel<4, false, false>, 8},          // round 4's probe (4.5 TB/s)
    {"blocked x4 nt, 8 WG/CU", probe_copy_kernel<4, true, true>, 8},
    {"blocked x4 nt, 16 WG/CU", probe_copy_kernel<4, true, true>, 16},
    {"blocked x8 nt, 8 WG/CU", probe_copy_kernel<8, true, true>, 8},
    {"blocked x4, 16 WG/CU", probe_copy_kernel<4, false, true>, 16},
    {"stride x4 nt, 16 WG/CU", probe_copy_kernel<4, true, false>, 16},
    {"blocked x2 nt, 32 WG/CU", probe_copy_kernel<2, true, true>, 32},
    {"blocked x1 nt, one tile per WG", probe_copy_kernel<1, true, true>, 0},   // grid = number of tiles: no loop at all
};
constexpr int kProbeCopyVariants = (int)(sizeof(kProbeCopy) / sizeof(kProbeCopy[0]));
template <int UNROLL>
__global__ void __launch_bounds__(256) probe_scatter8_kernel(const uint64_t* __restrict__ src, size_t mask8, uint32_t loads_per_lane, uint64_t* __restrict__ sink)
{   // UNROLL independent loads per trip: twice the loads in flight must not change the rate if the memory system is the limit
    uint64_t x = 0x9E3779B97F4A7C15ull * ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1), acc = 0;
    for (uint32_t k = 0; k < loads_per_lane; k += UNROLL) {
        uint64_t a[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; a[u] = x & mask8; }   // xorshift64: addresses do not depend on loaded data
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += src[a[u]];
    }
    if (acc == 0x1234567887654321ull) *sink = acc;                 // (keeps the loads alive)
}

// ms per copy of `bytes` bytes of the FASTEST variant (each: one warm-up launch, then `iters` launches between two events);
// AOCLGPU_PROBE_COPY_VARIANT=<k> runs variant k only, AOCLGPU_PROBE_TRACE=1 lists every variant's rate on stderr.
static int g_probe_copy_best = -1;
extern "C" const char* aoclgpu_probe_copy_variant(void) { return g_probe_copy_best >= 0 ? kProbeCopy[g_probe_copy_best].name : ""; }
extern "C" double aoclgpu_probe_copy(void* dst, const void* src, size_t bytes, int iters, void* stream)
{
    if (!dst || !src || bytes < 16 || iters <= 0) { set_error_msg("probe_copy: bad arguments"); return -1.0; }
    hipStream_t s = (hipStream_t)stream;
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { set_error_msg("probe_copy: device query failed"); return -1.0; }
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess) { set_error_msg("hipEventCreate failed"); return -1.0; }
    if (hipEventCreate(&e1) != hipSuccess) { hipEventDestroy(e0); set_error_msg("hipEventCreate failed"); return -1.0; }
    const char* only = getenv("AOCLGPU_PROBE_COPY_VARIANT");
    const bool trace = getenv("AOCLGPU_PROBE_TRACE") != nullptr;
    double best = -1.0;
    for (int k = 0; k < kProbeCopyVariants; ++k) {
        if (only && atoi(only) != k) continue;
        const ProbeCopyVariant& v = kProbeCopy[k];
        const size_t n16 = bytes / 16;
        const size_t tiles = n16 / 256 ? n16 / 256 : 1;
        const dim3 g(v.wg_per_cu ? (unsigned)prop.multiProcessorCount * v.wg_per_cu : (unsigned)(tiles > 0x7FFFFFFFu ? 0x7FFFFFFFu : tiles)), t(256);
        hipLaunchKernelGGL(v.fn, g, t, 0, s, (probe_u4*)dst, (const probe_u4*)src, n16);       // warm-up
        hipEventRecord(e0, s);
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(v.fn, g, t, 0, s, (probe_u4*)dst, (const probe_u4*)src, n16);
        hipEventRecord(e1, s);
        double ms = -1.0;
        if (hipEventSynchronize(e1) == hipSuccess && hipGetLastError() == hipSuccess) {
            float f = 0.f;
            if (hipEventElapsedTime(&f, e0, e1) == hipSuccess) ms = (double)f / iters;
        }
        if (trace) fprintf(stderr, "probe_copy[%d] %-32s %8.3f ms  %7.1f GB/s\n", k, v.name, ms, ms > 0 ? 2.0 * bytes / ms / 1e6 : 0.0);
        if (ms > 0 && (best < 0 || ms < best)) { best = ms; g_probe_copy_best = k; }
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return best;
}

extern "C" double aoclgpu_probe_scatter8(const void* src, size_t bytes, uint32_t loads_per_lane, uint64_t* total_loads, void* stream)
{
    if (!src || bytes < 64 || !loads_per_lane) { set_error_msg("probe_scatter8: bad arguments"); return -1.0; }
    hipStream_t s = (hipStream_t)stream;
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { set_error_msg("probe_scatter8: device query failed"); return -1.0; }
    uint64_t* sink = nullptr;
    if (hipMalloc((void**)&sink, 8) != hipSuccess) { set_error_msg("probe_scatter8: hipMalloc failed"); return -1.0; }
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess) { (void)hipFree(sink); set_error_msg("hipEventCreate failed"); return -1.0; }
    if (hipEventCreate(&e1) != hipSuccess) { hipEventDestroy(e0); (void)hipFree(sink); set_error_msg("hipEventCreate failed"); return -1.0; }
    const dim3 g((unsigned)prop.multiProcessorCount * 8), t(256);
    // addresses: a power-of-two part of the buffer (no division in the address chain); bit 31 of loads_per_lane asks for eight
    // independent loads per trip instead of four
    size_t n8 = 1; while (n8 * 2 <= bytes / 8) n8 *= 2;
    const bool wide = (loads_per_lane >> 31) != 0;
    loads_per_lane = (loads_per_lane & 0x7FFFFFFFu) / 8 * 8;
    if (!loads_per_lane) loads_per_lane = 8;
    hipEventRecord(e0, s);
    if (wide) hipLaunchKernelGGL(probe_scatter8_kernel<8>, g, t, 0, s, (const uint64_t*)src, n8 - 1, loads_per_lane, sink);
    else hipLaunchKernelGGL(probe_scatter8_kernel<4>, g, t, 0, s, (const uint64_t*)src, n8 - 1, loads_per_lane, sink);
    hipEventRecord(e1, s);
    double ms = -1.0;
    if (hipEventSynchronize(e1) == hipSuccess && hipGetLastError() == hipSuccess) {
        float f = 0.f;
        if (hipEventElapsedTime(&f, e0, e1) == hipSuccess) ms = (double)f;
    }
    if (total_loads) *total_loads = (uint64_t)g.x * t.x * loads_per_lane;
    hipEventDestroy(e0); hipEventDestroy(e1);
    (void)hipFree(sink);
    return ms;
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
// The reference's one-shot calls are re-entrant, so every call borrows a HostCtx (device buffers that only grow, one
// non-blocking stream, one pinned result word the kernel writes straight into host memory) from a process-wide pool and
// hands it back; concurrent callers get different contexts and overlap on the device.  Per call: one H2D of the input,
// one kernel, one stream synchronisation, one D2H of exactly the bytes produced.
namespace {
struct HostCtx {
    int dev = -1; hipStream_t stream = nullptr;
    uint8_t* dsrc = nullptr; size_t srcCap = 0;
    uint8_t* ddst = nullptr; size_t dstCap = 0;
    int32_t* hres = nullptr;          // pinned, device-visible: [0] = out_len written by the kernel, [1] = src_len read by it
    bool busy = false;
};
std::mutex g_host_mu;
std::vector<HostCtx*> g_host_ctx;
std::atomic<int> g_host_device{-1};

int host_device()
{
    int d = g_host_device.load();
    if (d >= 0) return d;
    const char* e = getenv("AOCLGPU_DEVICE");             // which GPU the host-buffer API uses (default 0); documented in INTEGRATION.md
    d = e ? atoi(e) : 0;
    if (d < 0) d = 0;
    g_host_device.store(d);
    return d;
}

HostCtx* host_acquire(int dev)
{
    {
        std::lock_guard<std::mutex> lk(g_host_mu);
        for (HostCtx* c : g_host_ctx) if (!c->busy && c->dev == dev) { c->busy = true; return c; }
    }
    HostCtx* c = new (std::nothrow) HostCtx();
    if (!c) return nullptr;
    c->dev = dev; c->busy = true;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipHostMalloc((void**)&c->hres, 64, hipHostMallocMapped) != hipSuccess) {
        if (c->stream) (void)hipStreamDestroy(c->stream);
        delete c; return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_host_mu);
    g_host_ctx.push_back(c);
    return c;
}
// A context keeps its buffers for the next call of the same size class; what a single huge call needed is given back at
// once (a co-resident process should not find several GiB of HBM pinned by a call that returned long ago).
constexpr size_t kHostCtxKeepBytes = (size_t)256 << 20;
void host_release(HostCtx* c)
{
    if (c->srcCap > kHostCtxKeepBytes) { (void)hipFree(c->dsrc); c->dsrc = nullptr; c->srcCap = 0; }
    if (c->dstCap > kHostCtxKeepBytes) { (void)hipFree(c->ddst); c->ddst = nullptr; c->dstCap = 0; }
    std::lock_guard<std::mutex> lk(g_host_mu); c->busy = false;
}

bool host_grow(uint8_t** p, size_t* cap, size_t need)
{
    if (*cap >= need && *p) return true;
    if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }
    size_t want = need < 4096 ? 4096 : need + need / 8;   // a little slack: callers usually repeat the same sizes
    if (hipMalloc((void**)p, want) != hipSuccess) {
        want = need ? need : 1;
        if (hipMalloc((void**)p, want) != hipSuccess) { *p = nullptr; return false; }
    }
    *cap = want;
    return true;
}
}  // namespace

// Frees the device buffers of every idle host context (aocl_llc_destroy calls it); contexts in use keep theirs.
namespace aoclgpu { size_t rap_host_cache_trim(); }             // rap.hip: the RAP host calls' cached buffer set
extern "C" void aoclgpu_host_trim(void)
{
    (void)aoclgpu::rap_host_cache_trim();
    {
        std::lock_guard<std::mutex> lk(g_host_mu);
        for (HostCtx* c : g_host_ctx) {
            if (c->busy) continue;
            DeviceScope ds(c->dev);
            if (c->dsrc) { (void)hipFree(c->dsrc); c->dsrc = nullptr; c->srcCap = 0; }
            if (c->ddst) { (void)hipFree(c->ddst); c->ddst = nullptr; c->dstCap = 0; }
        }
    }
    // the launchers' scratch pool too (hash tables, per-chunk workspaces - up to 32 GiB after a large Zstd level-3 batch);
    // hipFree waits for the work that still uses an entry
    WsGuard held;
    for (WsPool& q : g_pool) {
        if (!q.p) continue;
        DeviceScope ds(q.dev);
        (void)hipFree(q.p);
        q.p = nullptr; q.bytes = 0;
    }
}

extern "C" int aoclgpu_set_host_device(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) { set_error_msg("set_host_device: no such device"); return AOCLGPU_ERR_ARG; }
    g_host_device.store(device);
    return AOCLGPU_OK;
}
extern "C" int aoclgpu_host_device(void) { return host_device(); }

static int64_t host_oneshot(aoclgpu_codec codec, int op, const void* hsrc, size_t n, void* hdst, size_t cap, uint32_t flags = 0)
{
    batch_fn fn = pick(codec, op);
    if (!fn) { set_error_msg("bad codec"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_ARG; }
    if ((!hsrc && n) || !hdst || n > 0xFFFFFFFFull) { set_error_msg("host one-shot: bad arguments"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_ARG; }
    if (cap > 0xFFFFFFFFull) cap = 0xFFFFFFFFull;
    const int dev = host_device();
    DeviceScope devScope(dev);
    hipError_t e = hipSuccess;
    if (!devScope.ok) { set_error_msg("host one-shot: hipSetDevice failed"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_NO_DEVICE; }
    HostCtx* c = host_acquire(dev);
    if (!c) { set_error_msg("host one-shot: no context (stream / pinned memory allocation failed)"); return AOCLGPU_HOST_FAIL + AOCLGPU_ERR_HIP; }
    int64_t result = AOCLGPU_HOST_FAIL + AOCLGPU_ERR_HIP;
#define CK(call) do { e = (call); if (e != hipSuccess) { set_error(#call, e); goto done; } } while (0)
    if (!host_grow(&c->dsrc, &c->srcCap, n ? n : 1) || !host_grow(&c->ddst, &c->dstCap, cap ? cap : 1)) {
        set_error_msg("host one-shot: hipMalloc failed"); goto done;
    }
    if (n) CK(hipMemcpyAsync(c->dsrc, hsrc, n, hipMemcpyHostToDevice, c->stream));
    {
        aoclgpu_batch b; std::memset(&b, 0, sizeof(b));
        b.src = c->dsrc; b.dst = c->ddst; b.out_len = c->hres; b.nchunks = 1;
        b.reserved = flags;
        c->hres[0] = 0;
        if (op == 0) { b.total_bytes = n; b.chunk_bytes = n ? n : 1; b.slot_bytes = cap; }
        else {
            c->hres[1] = (int32_t)(uint32_t)n;
            b.src_len = (const uint32_t*)(c->hres + 1); b.total_bytes = cap; b.chunk_bytes = cap ? cap : 1; b.slot_bytes = n;
        }
        const int rc = fn(&b, c->stream);
        if (rc != 0) { result = AOCLGPU_HOST_FAIL + rc; goto done; }
        CK(hipStreamSynchronize(c->stream));
        int32_t r = ((volatile int32_t*)c->hres)[0];
        if (r < 0 && op == 1 && codec == AOCLGPU_LZ4 && !(flags & (AOCLGPU_FLAG_RAP_PARTITIONS | AOCLGPU_FLAG_LZ4_LINKED))) {
            // a malformed LZ4 block: the value the reference returns for it (lz4_kernels.hip, lz4_error_value_kernel)
            if (aoclgpu_lz4_error_value(c->dsrc, (uint32_t)n, (uint32_t)cap, 64u, (int32_t*)c->hres, c->stream) == 0) {
                CK(hipStreamSynchronize(c->stream));
                const int32_t v = ((volatile int32_t*)c->hres)[0];
                if (v < 0) r = v;                              // (a walk that finds nothing wrong leaves the decoder's own verdict)
            }
        }
        if (r > 0) {
            CK(hipMemcpyAsync(hdst, c->ddst, (size_t)r, hipMemcpyDeviceToHost, c->stream));
            CK(hipStreamSynchronize(c->stream));
        }
        result = r;
    }
#undef CK
done:
    host_release(c);
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
