// csrc/exchange.cpp -- BASELINE config 5 as a C product: the exchange of compressed segments between the GPUs of a node over
// RCCL / xGMI behind the C ABI (include/aocl_gpu_codec.h), so that a C host program - the reference's world - shards a
// stream over 8 GPUs without any Python: compress the rank's chunks, aoclgpu_compact_slots into the rank's place of the gather
// buffer, aoclgpu_allgather_words for the per-chunk sizes, aoclgpu_exchange_segments, aoclgpu_rap_assemble_slabs.
// The reference's counterpart is the serial join on the master thread (lz4.c:2736-2906, snappy.cc:2567-2654,
// zstd_compress.c:5499-5517); here the join's input arrives over the links.
//
// RCCL is bound at run time (dlopen of librccl.so.1, the soname PyTorch-ROCm's bundled copy also carries, so a process that
// already loaded one keeps exactly one RCCL): the library has no link-time dependency on it, and a single-GPU user never
// loads it.  xGMI is point-to-point (7 links per GPU): ONE group of exact-size ncclSend / ncclRecv pairs lets every pair of
// ranks talk over its own link at the same time, where a ring all-gather of segments padded to the largest one would move
// world x max(segment) bytes over one link per hop (RCCL has no all-gatherv).
#include "../../include/aocl_gpu_codec.h"
#include "host_util.h"
#include <dlfcn.h>
#include <mutex>
#include <stdio.h>
#include <string.h>

namespace {
using aoclgpu::set_error_msg;

struct NcclId { char internal[128]; };              // ncclUniqueId (rccl.h): an opaque 128-byte value
typedef void* Comm;                                 // ncclComm_t
typedef int (*fn_getid)(NcclId*);
typedef int (*fn_init)(Comm*, int, NcclId, int);
typedef int (*fn_destroy)(Comm);
typedef int (*fn_void)(void);
typedef int (*fn_send)(const void*, size_t, int, int, Comm, hipStream_t);
typedef int (*fn_recv)(void*, size_t, int, int, Comm, hipStream_t);
typedef int (*fn_allgather)(const void*, void*, size_t, int, Comm, hipStream_t);
typedef const char* (*fn_errstr)(int);
typedef int (*fn_count)(Comm, int*);
enum { kNcclUint8 = 1, kNcclUint32 = 3 };           // ncclDataType_t (rccl.h)

struct Rccl {
    void* h = nullptr;
    fn_getid getid = nullptr; fn_init init = nullptr; fn_destroy destroy = nullptr; fn_void gstart = nullptr, gend = nullptr;
    fn_send send = nullptr; fn_recv recv = nullptr; fn_allgather allgather = nullptr; fn_errstr errstr = nullptr;
    fn_count count = nullptr;                       // optional: ncclCommCount (aoclgpu_comm_count reports an error without it)
    bool ok = false;
};
Rccl g_rccl;
std::once_flag g_once;

const Rccl& rccl()
{
    std::call_once(g_once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) { g_rccl.h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (g_rccl.h) break; }
        if (!g_rccl.h) return;
        g_rccl.getid = (fn_getid)dlsym(g_rccl.h, "ncclGetUniqueId"); g_rccl.init = (fn_init)dlsym(g_rccl.h, "ncclCommInitRank");
        g_rccl.destroy = (fn_destroy)dlsym(g_rccl.h, "ncclCommDestroy");
        g_rccl.gstart = (fn_void)dlsym(g_rccl.h, "ncclGroupStart"); g_rccl.gend = (fn_void)dlsym(g_rccl.h, "ncclGroupEnd");
        g_rccl.send = (fn_send)dlsym(g_rccl.h, "ncclSend"); g_rccl.recv = (fn_recv)dlsym(g_rccl.h, "ncclRecv");
        g_rccl.allgather = (fn_allgather)dlsym(g_rccl.h, "ncclAllGather"); g_rccl.errstr = (fn_errstr)dlsym(g_rccl.h, "ncclGetErrorString");
        g_rccl.count = (fn_count)dlsym(g_rccl.h, "ncclCommCount");
        g_rccl.ok = g_rccl.getid && g_rccl.init && g_rccl.destroy && g_rccl.gstart && g_rccl.gend && g_rccl.send && g_rccl.recv && g_rccl.allgather;
    });
    return g_rccl;
}

int fail(const char* what, int rc)
{
    char msg[200];
    const Rccl& r = g_rccl;
    snprintf(msg, sizeof(msg), "%s: %s", what, (r.errstr && rc) ? r.errstr(rc) : "RCCL (librccl.so.1) is not available");
    set_error_msg(msg);
    return AOCLGPU_ERR_HIP;
}
}  // namespace

extern "C" int aoclgpu_comm_unique_id(uint8_t id[AOCLGPU_COMM_ID_BYTES])
{
    const Rccl& r = rccl();
    if (!r.ok || !id) return fail("comm_unique_id", 0);
    NcclId v; memset(&v, 0, sizeof(v));
    if (int rc = r.getid(&v)) return fail("ncclGetUniqueId", rc);
    memcpy(id, &v, sizeof(v));
    return AOCLGPU_OK;
}

extern "C" int aoclgpu_comm_create(const uint8_t id[AOCLGPU_COMM_ID_BYTES], int nranks, int rank, void** comm)
{
    const Rccl& r = rccl();
    if (!r.ok || !id || !comm || nranks < 1 || rank < 0 || rank >= nranks) return fail("comm_create: bad arguments", 0);
    NcclId v; memcpy(&v, id, sizeof(v));
    Comm c = nullptr;
    if (int rc = r.init(&c, nranks, v, rank)) return fail("ncclCommInitRank", rc);
    *comm = c;
    return AOCLGPU_OK;
}

extern "C" int aoclgpu_comm_destroy(void* comm)
{
    const Rccl& r = rccl();
    if (!r.ok || !comm) return fail("comm_destroy", 0);
    if (int rc = r.destroy((Comm)comm)) return fail("ncclCommDestroy", rc);
    return AOCLGPU_OK;
}

extern "C" int aoclgpu_comm_count(void* comm, int* nranks)
{
    const Rccl& r = rccl();
    if (!r.ok || !r.count || !comm || !nranks) return fail("comm_count", 0);
    int n = 0;
    if (int rc = r.count((Comm)comm, &n)) return fail("ncclCommCount", rc);
    *nranks = n;
    return AOCLGPU_OK;
}

extern "C" int aoclgpu_exchange_segments(void* comm, void* gather_buf, const uint64_t* seg_bytes, const uint64_t* seg_off,
                                         int nranks, int rank, void* stream)
{
    if (!gather_buf || !seg_bytes || !seg_off || nranks < 1 || rank < 0 || rank >= nranks) { set_error_msg("exchange_segments: bad arguments"); return AOCLGPU_ERR_ARG; }
    if (nranks == 1) return AOCLGPU_OK;
    const Rccl& r = rccl();
    if (!r.ok || !comm) return fail("exchange_segments", 0);
    uint8_t* base = (uint8_t*)gather_buf;
    hipStream_t s = (hipStream_t)stream;
    int rc = r.gstart();
    if (rc) return fail("ncclGroupStart", rc);
    for (int d = 1; d < nranks && !rc; ++d) {                             // peer distance d: send to rank + d, receive from rank - d
        const int to = (rank + d) % nranks, from = (rank - d + nranks) % nranks;
        if (seg_bytes[rank]) rc = r.send(base + seg_off[rank], (size_t)seg_bytes[rank], kNcclUint8, to, (Comm)comm, s);
        if (!rc && seg_bytes[from]) rc = r.recv(base + seg_off[from], (size_t)seg_bytes[from], kNcclUint8, from, (Comm)comm, s);
    }
    const int rc2 = r.gend();
    if (rc) return fail("ncclSend / ncclRecv", rc);
    if (rc2) return fail("ncclGroupEnd", rc2);
    return AOCLGPU_OK;
}

extern "C" int aoclgpu_allgather_words(void* comm, const void* mine_dev, void* all_dev, size_t nwords, int nranks, int rank, void* stream)
{
    if (!mine_dev || !all_dev || nranks < 1 || rank < 0 || rank >= nranks) { set_error_msg("allgather_words: bad arguments"); return AOCLGPU_ERR_ARG; }
    if (nranks == 1) {
        if (mine_dev != all_dev && hipMemcpyAsync(all_dev, mine_dev, nwords * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) { set_error_msg("allgather_words: copy failed"); return AOCLGPU_ERR_HIP; }
        return AOCLGPU_OK;
    }
    const Rccl& r = rccl();
    if (!r.ok || !comm) return fail("allgather_words", 0);
    if (int rc = r.allgather(mine_dev, all_dev, nwords, kNcclUint32, (Comm)comm, (hipStream_t)stream)) return fail("ncclAllGather", rc);
    return AOCLGPU_OK;
}
