// csrc/aocl_api.cpp -- the reference's unified API (api/aocl_compression.h:170-231) on top of the GPU engine.
//
// Mirrors api/api.cpp:45-189 + the LZ4 / SNAPPY / ZSTD wrappers of api/codec.cpp:115-157,262-308,352-436:
// same return codes (wrapper result < 0 -> ERR_COMPRESSION_FAILED), same measureStats fields (ns,
// bytes*1000/ns), same argument checks.  Buffers are HOST buffers as in the reference; one call moves the
// input to HBM, runs ONE chunk (= exactly the bytes the reference's one-shot native call emits) and moves
// the result back.  Bulk work should use the device-resident batch entry points of aocl_gpu_codec.h instead.
// There is no CPU fallback: without a gfx950 device every call fails with ERR_COMPRESSION_FAILED.
#include "../../include/aocl_compression.h"
#include "../../include/aocl_gpu_codec.h"
#include "../../include/aocl_native_api.h"
#include <atomic>
#include <time.h>
#include <stdlib.h>
#include <string.h>

extern "C" int aoclgpu_zstd_level_table(void);     // zstd_api.cpp

namespace {

const int64_t kCodecError = -1;   // api/codec.h CODEC_ERROR

uint64_t now_ns()
{   // utils/utils.h:244-247: clock_gettime(CLOCK_REALTIME), nanoseconds
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

// numThreads > 1: 0 (default) = GPU geometry, the codec's chunk (128 KiB for Zstd) is the partition; 1 = the reference's
// own partition geometry for Zstd (few partitions of at least a window each: byte-identical to its threaded build for the
// same thread count, one wavefront per partition).  aoclgpu_set_rap_geometry / AOCLGPU_RAP_GEOMETRY=reference.
std::atomic<int> g_rap_geometry{-1};
int rap_geometry()
{
    int m = g_rap_geometry.load();
    if (m < 0) { const char* e = getenv("AOCLGPU_RAP_GEOMETRY"); m = (e && !strcmp(e, "reference")) ? 1 : 0; g_rap_geometry.store(m); }
    return m;
}

// The codec type arrives as an enum a caller may have filled with anything (the reference's own tests pass out-of-range
// values, gtest/api/api_gtest.cpp): it is read as the integer it is, never through the enum type.
int raw_type(const aocl_compression_type& t) { int v; memcpy(&v, &t, sizeof(v)); return v; }

bool gpu_codec(int t, aoclgpu_codec* out)
{
    switch (t) {
    case LZ4: *out = AOCLGPU_LZ4; return true;
    case LZ4HC: *out = AOCLGPU_LZ4; return true;      // decode only: LZ4HC streams are LZ4 blocks (api/codec.cpp:179-188)
    case SNAPPY: *out = AOCLGPU_SNAPPY; return true;
    case ZSTD: *out = AOCLGPU_ZSTD; return true;
    default: return false;
    }
}

int64_t wrap_compress(int t, const aocl_compression_desc* h)
{
    aoclgpu_codec c;
    if (!gpu_codec(t, &c)) return kCodecError;
    if (t == LZ4HC) return kCodecError;               // no GPU match finder for the HC levels (api/codec.cpp:169-177)
    if (!h->outBuf || (!h->inBuf && h->inSize != 0)) return kCodecError;
    if (h->numThreads > 1) {
        // The reference's AOCL_ENABLE_THREADS build partitions the input inside this call and returns a RAP stream
        // (lz4.c:2655-2909, snappy.cc:2505-2654, zstd_compress.c:5388-5534; callers add aocl_get_rap_frame_bound_mt()
        // to outSize).  Here `numThreads > 1` asks for the same with GPU chunks as the partitions.
        if (h->inSize > 0x7FFFFFFFu) return kCodecError;
        uint32_t flags = 0;
        if (t == ZSTD) {
            // every partition is compressed with the level's parameters for ITS size, as each thread of the reference does
            // with the caller's parameters (zstd_compress.c:5447-5468); levels with a GPU parse: 1-4 (0 = the default, 3)
            const int level = h->level == 0 ? 3 : (int)h->level;
            if (level < 1 || level > 4) return kCodecError;
            flags = AOCLGPU_ZSTD_LEVEL((uint32_t)level) | (aoclgpu_zstd_level_table() ? 0u : AOCLGPU_FLAG_ZSTD_UPSTREAM_TABLE);
            if (rap_geometry() == 1) {
                // the reference's own geometry: min(numThreads, partitions of window size) frames (threads/threads.c:46-118)
                ZSTD_parameters zp = ZSTD_getParams((int)h->level, h->inSize, 0);
                if (h->optVar && zp.cParams.windowLog > h->optVar) {        // api/codec.cpp:402-408: the window AND the chain table follow optVar
                    zp.cParams.windowLog = (unsigned)h->optVar;
                    zp.cParams.chainLog = (unsigned)h->optVar + ((zp.cParams.strategy == ZSTD_btlazy2) || (zp.cParams.strategy == ZSTD_btopt) ||
                                                                  (zp.cParams.strategy == ZSTD_btultra));
                }
                const uint64_t wf = h->inSize < (100ull << 20) ? 1 : (h->inSize > (400ull << 20) ? 4 : h->inSize / (100ull << 20));
                const uint64_t chunk = ((uint64_t)1 << zp.cParams.windowLog) * wf;
                uint64_t parts = 1;
                if (h->inSize >= chunk) {
                    parts = h->inSize / chunk;
                    if (h->inSize % chunk >= (wf > 1 ? chunk >> 1 : ((uint64_t)1 << zp.cParams.windowLog) >> 1)) parts++;
                    if (parts > (uint64_t)h->numThreads) parts = (uint64_t)h->numThreads;
                }
                const int64_t r = aoclgpu_zstd_rap_compress_host_parts(h->inBuf, h->inSize, h->outBuf, h->outSize, (uint32_t)parts,
                                                                       (uint32_t)zp.cParams.strategy, zp.cParams.windowLog, zp.cParams.hashLog,
                                                                       zp.cParams.chainLog, zp.cParams.minMatch < 4 ? 4u : zp.cParams.minMatch,
                                                                       zp.cParams.targetLength);
                return r > 0 ? r : kCodecError;
            }
        }
        int64_t r = aoclgpu_rap_compress_host_ex(c, h->inBuf, h->inSize, h->outBuf, h->outSize, 0, flags);
        return r > 0 ? r : kCodecError;
    }
    if (t == LZ4) {                                   // api/codec.cpp:126-134: LZ4_compress_default takes int sizes
        if (h->inSize > 0x7E000000u) return kCodecError;
        size_t cap = h->outSize > 0x7FFFFFFFu ? 0x7FFFFFFFu : h->outSize;
        int64_t r = aoclgpu_compress_host(c, h->inBuf, h->inSize, h->outBuf, cap);
        return r > 0 ? r : kCodecError;
    }
    if (t == SNAPPY) {                                // api/codec.cpp:268-290
        if (h->outSize < aoclgpu_compress_bound(c, h->inSize)) return kCodecError;
        int64_t r = aoclgpu_compress_host(c, h->inBuf, h->inSize, h->outBuf, h->outSize);
        return r > 0 ? r : kCodecError;
    }
    // ZSTD: the call sequence of the reference's wrapper (api/codec.cpp:388-422) on this library's own Zstd entry points:
    // parameters from the level table, optVar clamps windowLog, ZSTD_compress_advanced.  Levels 0-4 and the negative ones (fast and double-fast,
    // windows below the input size included) are compressed; greedy and above come back as
    // ZSTD_error_parameter_unsupported -> CODEC_ERROR, never as different bytes.
    ZSTD_parameters zp = ZSTD_getParams((int)h->level, h->inSize, 0);
    zp.fParams.contentSizeFlag = 1;
    if (h->optVar && zp.cParams.windowLog > h->optVar) {
        zp.cParams.windowLog = (unsigned)h->optVar;
        zp.cParams.chainLog = (unsigned)h->optVar + ((zp.cParams.strategy == ZSTD_btlazy2) || (zp.cParams.strategy == ZSTD_btopt) ||
                                                      (zp.cParams.strategy == ZSTD_btultra));
    }
    static ZSTD_CCtx* const kCtx = ZSTD_createCCtx();   // stateless in this library: one shared handle satisfies the signature
    const size_t r = ZSTD_compress_advanced(kCtx, h->outBuf, h->outSize, h->inBuf, h->inSize, NULL, 0, zp);
    return ZSTD_isError(r) ? kCodecError : (int64_t)r;
}

int64_t wrap_decompress(int t, const aocl_compression_desc* h)
{
    aoclgpu_codec c;
    if (!gpu_codec(t, &c)) return kCodecError;
    if (!h->inBuf || !h->outBuf) return kCodecError;
    // RAP streams (threads/threads.c:183-300): split at the frame's partition table and decode the partitions as one batch
    static const unsigned char kRap[8] = {'A', 'O', 'C', 'L', '_', 'L', 'L', 'C'}, kSkip[4] = {0x50, 0x2A, 0x4D, 0x18};
    const unsigned char* p = (const unsigned char*)h->inBuf;
    const bool rap = (h->inSize >= 16 && memcmp(p, kRap, 8) == 0) ||
                     (t == ZSTD && h->inSize >= 24 && memcmp(p, kSkip, 4) == 0 && memcmp(p + 8, kRap, 8) == 0);
    int64_t r = rap ? aoclgpu_rap_decompress_host(c, h->inBuf, h->inSize, h->outBuf, h->outSize)
              : t == ZSTD ? aoclgpu_zstd_decompress_host_frames(h->inBuf, h->inSize, h->outBuf, h->outSize)
                          : aoclgpu_decompress_host(c, h->inBuf, h->inSize, h->outBuf, h->outSize);
    return r >= 0 ? r : kCodecError;                  // api/codec.cpp:143-151
}

}  // namespace

// Trace lines of the reference's logger (utils/utils.h:92-208, built with AOCL_ENABLE_LOG_FEATURE): the level comes from the
// environment at every statement (AOCL_ENABLE_LOG = ERR | INFO | DEBUG | TRACE), errors go to stderr, everything else to stdout,
// one line "[LEVEL] : file : function : line : text".  Always compiled in here; silent without the variable.
namespace {
enum { kLogErr = 1, kLogInfo = 2, kLogDebug = 3, kLogTrace = 4 };
int log_max_level()
{
    const char* e = getenv("AOCL_ENABLE_LOG");
    if (!e) return 0;
    return !strcmp(e, "ERR") ? 1 : !strcmp(e, "INFO") ? 2 : !strcmp(e, "DEBUG") ? 3 : !strcmp(e, "TRACE") ? 4 : 0;
}
const char* codec_name(int t)
{   // aocl_codec[].codec_name (api/codec.h:165-193)
    static const char* const names[] = {"lz4", "lz4hc", "lzma", "bzip2", "snappy", "zlib", "zstd"};
    return (t >= 0 && t < (int)(sizeof(names) / sizeof(names[0]))) ? names[t] : "?";
}
}  // namespace
#define AOCL_LOG(level, ...) do { if ((level) <= log_max_level()) { \
        FILE* f_ = (level) == kLogErr ? stderr : stdout; \
        fprintf(f_, "[%s] : %s : %s : %d : ", (level) == kLogErr ? "ERR" : (level) == kLogInfo ? "INFO" : (level) == kLogDebug ? "DEBUG" : "TRACE", __FILE__, __func__, __LINE__); \
        fprintf(f_, __VA_ARGS__); fprintf(f_, "\n"); } } while (0)

extern "C" void aoclgpu_set_rap_geometry(int reference) { g_rap_geometry.store(reference ? 1 : 0); }
extern "C" int aoclgpu_rap_geometry(void) { return rap_geometry(); }

extern "C" int64_t aocl_llc_compress(aocl_compression_desc* handle, aocl_compression_type codec_type)
{
    if (!handle) return ERR_COMPRESSION_FAILED;
    AOCL_LOG(kLogTrace, "Enter");
    AOCL_LOG(kLogInfo, "Calling compression method: %s", codec_name(raw_type(codec_type)));   // api/api.cpp:54-57
    const uint64_t t0 = now_ns();
    const int64_t ret = wrap_compress(raw_type(codec_type), handle);
    const uint64_t t1 = now_ns();
    if (handle->measureStats == 1) {                  // api/api.cpp:70-75
        handle->cSize = (uint64_t)ret;
        handle->cTime = t1 - t0;
        handle->cSpeed = (float)((handle->inSize * 1000.0) / (double)handle->cTime);
    }
    AOCL_LOG(kLogTrace, "Exit");
    return ret < 0 ? ERR_COMPRESSION_FAILED : ret;
}

extern "C" int64_t aocl_llc_decompress(aocl_compression_desc* handle, aocl_compression_type codec_type)
{
    if (!handle) return ERR_COMPRESSION_FAILED;
    AOCL_LOG(kLogTrace, "Enter");
    AOCL_LOG(kLogInfo, "Calling decompression method: %s", codec_name(raw_type(codec_type)));  // api/api.cpp:95-98
    const uint64_t t0 = now_ns();
    const int64_t ret = wrap_decompress(raw_type(codec_type), handle);
    const uint64_t t1 = now_ns();
    if (handle->measureStats == 1) {                  // api/api.cpp:111-116
        handle->dSize = (uint64_t)ret;
        handle->dTime = t1 - t0;
        handle->dSpeed = (float)((handle->dSize * 1000.0) / (double)handle->dTime);
    }
    AOCL_LOG(kLogTrace, "Exit");
    return ret < 0 ? ERR_COMPRESSION_FAILED : ret;
}

extern "C" int32_t aocl_llc_setup(aocl_compression_desc* handle, aocl_compression_type codec_type)
{
    const int ct = raw_type(codec_type);
    AOCL_LOG(kLogTrace, "Enter");
    if (ct < (int)LZ4 || ct >= (int)AOCL_COMPRESSOR_ALGOS_NUM) {
        AOCL_LOG(kLogErr, "setup failed !! compression method is not supported.");
        return ERR_UNSUPPORTED_METHOD;                // api/api.cpp:133-138
    }
    if (handle) AOCL_LOG(kLogInfo, "All optimizations are turned %s", handle->optOff ? "off" : "on");   // :140-141
    AOCL_LOG(kLogInfo, "Calling setup method for: %s", codec_name(ct));
    aoclgpu_codec c;
    if (!gpu_codec(ct, &c)) {                         // api/api.cpp:155-161 (no backend in this build)
        AOCL_LOG(kLogErr, "setup failed !! compression method is excluded from this library build.");
        AOCL_LOG(kLogTrace, "Exit");
        return ERR_EXCLUDED_METHOD;
    }
    if (handle) {
        // utils/utils.cpp:148-175 derives optLevel from CPUID; host ISA levels are meaningless for the GPU
        // backend, so it reports 0 ("no host SIMD path selected").  workBuf: LZ4/Snappy keep no context
        // (api/codec.cpp:118-122,262-266); the Zstd tables live on the device per launch.
        handle->optLevel = 0;
        // LZ4 / Snappy keep no context (workBuf NULL, api/codec.cpp:118-122,262-266); Zstd's setup hands back a context block
        // (api/codec.cpp:354-376; gtest/api/api_gtest.cpp:771-778 checks it is not NULL) - it is freed by aocl_llc_destroy
        handle->workBuf = NULL;
        if (ct == (int)ZSTD) {
            void** ctx = (void**)malloc(2 * sizeof(void*));
            if (ctx) { ctx[0] = ZSTD_createCCtx(); ctx[1] = ZSTD_createDCtx(); }
            handle->workBuf = (char*)ctx;
        }
        // the reference's setup runs the codec's dispatcher (api/codec.cpp:354-376); for Zstd that also selects the level
        // table (optOff -> upstream parameters, zstd_compress.c:7390-7425)
        if (ct == (int)ZSTD) (void)aocl_setup_zstd_encode(handle->optOff, 0, handle->inSize, handle->level, handle->optVar);
    }
    if (aoclgpu_init(aoclgpu_host_device()) != AOCLGPU_OK) {   // AOCLGPU_DEVICE / aoclgpu_set_host_device
        AOCL_LOG(kLogErr, "setup failed !! no gfx950 device for the GPU backend.");
        AOCL_LOG(kLogTrace, "Exit");
        return ERR_EXCLUDED_METHOD;
    }
    AOCL_LOG(kLogTrace, "Exit");
    return 0;
}

extern "C" void aocl_llc_destroy(aocl_compression_desc* handle, aocl_compression_type codec_type)
{
    if (!handle) return;
    AOCL_LOG(kLogTrace, "Enter");
    AOCL_LOG(kLogInfo, "Calling destroy method for: %s", codec_name(raw_type(codec_type)));   // api/api.cpp:172-175
    if (raw_type(codec_type) == (int)ZSTD && handle->workBuf) {          // api/codec.cpp:378-394
        void** ctx = (void**)handle->workBuf;
        ZSTD_freeCCtx((ZSTD_CCtx*)ctx[0]); ZSTD_freeDCtx((ZSTD_DCtx*)ctx[1]);
        free(ctx);
        aocl_destroy_zstd_encode(); aocl_destroy_zstd_decode();
    }
    handle->workBuf = NULL;
    aoclgpu_host_trim();                                  // staging buffers of idle host contexts go back to the device
    AOCL_LOG(kLogTrace, "Exit");
}

extern "C" const char* aocl_llc_version(void)
{
    return AOCL_COMPRESSION_LIBRARY_VERSION " Build gfx950";   // api/api.cpp:186-189
}
