// csrc/zstd_api.cpp -- the reference's Zstd native entry points on top of the GPU engine (include/aocl_native_api.h):
// one-shot calls, the context family the reference's own wrapper uses (api/codec.cpp:354-436: ZSTD_createCCtx /
// ZSTD_getParams / ZSTD_CCtx_setParameter / ZSTD_compress_advanced / ZSTD_decompressDCtx) and the parameter tables.
// Host buffers, no CPU fallback.
//
// What has a GPU backend: the "fast" and "double-fast" strategies (the parameters of levels 1-4 and of every negative level,
// clevels.h:24-130, 155-262, or any explicit parameter set of those strategies: hashLog / chainLog <= 24, minMatch 3-7, fast
// with any targetLength), with or without a content checksum, windows above or below the input size.  Inputs of at most 128 KiB with a
// level's own parameters are one block per frame on the chunk kernels; everything else - above all inputs LARGER than
// 128 KiB - is ONE frame of blocks, byte-identical to the reference's ZSTD_compress_advanced, written by one wavefront
// (aoclgpu_zstd_compress_frames): correct and slow, ~20 MiB/s for a lone input.  The opt-out for bulk callers is
// aoclgpu_zstd_set_large_input_mode(1) / AOCLGPU_ZSTD_LARGE_INPUTS=pieces: 128 KiB pieces as concatenated frames side by
// side (a valid stream for every Zstd decoder, not the reference's bytes).  Other strategies, dictionaries and frames without a
// content size return ZSTD_error_parameter_unsupported instead of silently compressing differently.
//
// Level tables as in the reference (zstd_compress.c:72, 7390-7463, clevels.h:262): a process starts on the UPSTREAM table;
// the first compress-side entry point (or aocl_setup_zstd_encode) runs the dispatcher once, which installs the AOCL table
// unless optimisations are off (optOff / AOCL_DISABLE_OPT=ON); later setups are ignored until aocl_destroy_zstd_encode
// clears the "set up" flag - which leaves the table where it was.
#include "../../include/aocl_native_api.h"
#include "../../include/aocl_gpu_codec.h"
#include <atomic>
#include <mutex>
#include <new>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {
size_t zerr(int code) { return (size_t)0 - (size_t)code; }
enum { kErrGeneric = 1, kErrPrefix = 10, kErrFrameParamUnsupported = 14, kErrFrameWindowTooLarge = 16, kErrCorrupt = 20, kErrParamUnsupported = 40, kErrParamOutOfBound = 42,
       kErrMemory = 64, kErrDstTooSmall = 70, kErrSrcSize = 72, kErrDstNull = 74, kErrMax = 120 };   // zstd_errors.h:58-100

// {windowLog, chainLog, hashLog, searchLog, minMatch, targetLength, strategy} for level rows 0..22 of the four size
// classes (> 256 KB, <= 256 KB, <= 128 KB, <= 16 KB): [0] = ZSTD_defaultCParameters (clevels.h:24-130),
// [1] = AOCL_ZSTD_defaultCParameters (clevels.h:155-262, the table the reference switches to once its dispatcher has
// run with optimisations on: zstd_compress.c:7390-7425)
const uint16_t kLevelRows[2][4][23][7] = {
  {
    {
      {19,12,13,1,6,1,1}, {19,13,14,1,7,0,1}, {20,15,16,1,6,0,1}, {21,16,17,1,5,0,2}, {21,18,18,1,5,0,2}, {21,18,19,3,5,2,3},
      {21,18,19,3,5,4,4}, {21,19,20,4,5,8,4}, {21,19,20,4,5,16,5}, {22,20,21,4,5,16,5}, {22,21,22,5,5,16,5}, {22,21,22,6,5,16,5},
      {22,22,23,6,5,32,5}, {22,22,22,4,5,32,6}, {22,22,23,5,5,32,6}, {22,23,23,6,5,32,6}, {22,22,22,5,5,48,7}, {23,23,22,5,4,64,7},
      {23,23,22,6,3,64,8}, {23,24,22,7,3,256,9}, {25,25,23,7,3,256,9}, {26,26,24,7,3,512,9}, {27,27,25,9,3,999,9},
    },
    {
      {18,12,13,1,5,1,1}, {18,13,14,1,6,0,1}, {18,14,14,1,5,0,2}, {18,16,16,1,4,0,2}, {18,16,17,3,5,2,3}, {18,17,18,5,5,2,3},
      {18,18,19,3,5,4,4}, {18,18,19,4,4,4,4}, {18,18,19,4,4,8,5}, {18,18,19,5,4,8,5}, {18,18,19,6,4,8,5}, {18,18,19,5,4,12,6},
      {18,19,19,7,4,12,6}, {18,18,19,4,4,16,7}, {18,18,19,4,3,32,7}, {18,18,19,6,3,128,7}, {18,19,19,6,3,128,8}, {18,19,19,8,3,256,8},
      {18,19,19,6,3,128,9}, {18,19,19,8,3,256,9}, {18,19,19,10,3,512,9}, {18,19,19,12,3,512,9}, {18,19,19,13,3,999,9},
    },
    {
      {17,12,12,1,5,1,1}, {17,12,13,1,6,0,1}, {17,13,15,1,5,0,1}, {17,15,16,2,5,0,2}, {17,17,17,2,4,0,2}, {17,16,17,3,4,2,3},
      {17,16,17,3,4,4,4}, {17,16,17,3,4,8,5}, {17,16,17,4,4,8,5}, {17,16,17,5,4,8,5}, {17,16,17,6,4,8,5}, {17,17,17,5,4,8,6},
      {17,18,17,7,4,12,6}, {17,18,17,3,4,12,7}, {17,18,17,4,3,32,7}, {17,18,17,6,3,256,7}, {17,18,17,6,3,128,8}, {17,18,17,8,3,256,8},
      {17,18,17,10,3,512,8}, {17,18,17,5,3,256,9}, {17,18,17,7,3,512,9}, {17,18,17,9,3,512,9}, {17,18,17,11,3,999,9},
    },
    {
      {14,12,13,1,5,1,1}, {14,14,15,1,5,0,1}, {14,14,15,1,4,0,1}, {14,14,15,2,4,0,2}, {14,14,14,4,4,2,3}, {14,14,14,3,4,4,4},
      {14,14,14,4,4,8,5}, {14,14,14,6,4,8,5}, {14,14,14,8,4,8,5}, {14,15,14,5,4,8,6}, {14,15,14,9,4,8,6}, {14,15,14,3,4,12,7},
      {14,15,14,4,3,24,7}, {14,15,14,5,3,32,8}, {14,15,15,6,3,64,8}, {14,15,15,7,3,256,8}, {14,15,15,5,3,48,9}, {14,15,15,6,3,128,9},
      {14,15,15,7,3,256,9}, {14,15,15,8,3,256,9}, {14,15,15,8,3,512,9}, {14,15,15,9,3,512,9}, {14,15,15,10,3,999,9},
    },
  },
  {
    {
      {19,12,13,1,6,1,1}, {23,13,14,1,7,0,1}, {23,15,16,1,6,0,1}, {23,15,17,1,5,0,2}, {23,16,18,1,5,0,2}, {21,18,19,3,5,2,3},
      {21,18,19,3,5,4,4}, {21,19,20,4,5,8,4}, {21,19,20,4,5,16,5}, {22,20,21,4,5,16,5}, {22,21,22,5,5,16,5}, {22,21,22,6,5,16,5},
      {22,22,23,6,5,32,5}, {22,22,22,4,5,32,6}, {22,22,23,5,5,32,6}, {22,23,23,6,5,32,6}, {22,22,22,5,5,48,7}, {23,23,22,5,4,64,7},
      {23,23,22,6,3,64,8}, {23,24,22,7,3,256,9}, {25,25,23,7,3,256,9}, {26,26,24,7,3,512,9}, {27,27,25,9,3,999,9},
    },
    {
      {18,12,13,1,5,1,1}, {18,13,14,1,6,0,1}, {18,14,14,1,5,0,2}, {18,16,16,1,4,0,2}, {18,16,17,3,5,2,3}, {18,17,18,5,5,2,3},
      {18,18,19,3,5,4,4}, {18,18,19,4,4,4,4}, {18,18,19,4,4,8,5}, {18,18,19,5,4,8,5}, {18,18,19,6,4,8,5}, {18,18,19,5,4,12,6},
      {18,19,19,7,4,12,6}, {18,18,19,4,4,16,7}, {18,18,19,4,3,32,7}, {18,18,19,6,3,128,7}, {18,19,19,6,3,128,8}, {18,19,19,8,3,256,8},
      {18,19,19,6,3,128,9}, {18,19,19,8,3,256,9}, {18,19,19,10,3,512,9}, {18,19,19,12,3,512,9}, {18,19,19,13,3,999,9},
    },
    {
      {17,12,12,1,5,1,1}, {17,12,13,1,6,0,1}, {17,13,15,1,5,0,1}, {17,14,16,2,5,0,2}, {17,15,17,2,4,0,2}, {17,16,17,3,4,2,3},
      {17,16,17,3,4,4,4}, {17,16,17,3,4,8,5}, {17,16,17,4,4,8,5}, {17,16,17,5,4,8,5}, {17,16,17,6,4,8,5}, {17,17,17,5,4,8,6},
      {17,18,17,7,4,12,6}, {17,18,17,3,4,12,7}, {17,18,17,4,3,32,7}, {17,18,17,6,3,256,7}, {17,18,17,6,3,128,8}, {17,18,17,8,3,256,8},
      {17,18,17,10,3,512,8}, {17,18,17,5,3,256,9}, {17,18,17,7,3,512,9}, {17,18,17,9,3,512,9}, {17,18,17,11,3,999,9},
    },
    {
      {14,12,13,1,5,1,1}, {14,14,15,1,5,0,1}, {14,14,15,1,4,0,1}, {14,13,15,2,4,0,2}, {14,14,14,4,4,2,3}, {14,14,14,3,4,4,4},
      {14,14,14,4,4,8,5}, {14,14,14,6,4,8,5}, {14,14,14,8,4,8,5}, {14,15,14,5,4,8,6}, {14,15,14,9,4,8,6}, {14,15,14,3,4,12,7},
      {14,15,14,4,3,24,7}, {14,15,14,5,3,32,8}, {14,15,15,6,3,64,8}, {14,15,15,7,3,256,8}, {14,15,15,5,3,48,9}, {14,15,15,6,3,128,9},
      {14,15,15,7,3,256,9}, {14,15,15,8,3,256,9}, {14,15,15,8,3,512,9}, {14,15,15,9,3,512,9}, {14,15,15,10,3,999,9},
    },
  },
};
std::atomic<int> g_aocl_table{0};      // AOCL_ZSTD_defaultCParameters_used: starts on the upstream table (clevels.h:262)
std::atomic<int> g_setup_ok{0};        // setup_ok_zstd_encode (zstd_compress.c:72)
std::mutex g_setup_mu;

bool disable_opt_env()
{   // get_disable_opt_flags (utils/utils.cpp:207-219)
    const char* e = getenv("AOCL_DISABLE_OPT");
    return e && strcmp(e, "ON") == 0;
}
void dispatcher_setup(int optOff)
{   // aocl_setup_zstd_encode / aocl_setup_native (zstd_compress.c:7429-7457): once, until aocl_destroy_zstd_encode
    std::lock_guard<std::mutex> lk(g_setup_mu);
    if (g_setup_ok.load()) return;
    const bool off = optOff ? true : disable_opt_env();
    g_aocl_table.store(off ? 0 : 1);   // aocl_register_zstd_compress_fmv (:7390-7425), a build with AOCL_ZSTD_OPT
    g_setup_ok.store(1);
}
void setup_native() { dispatcher_setup(0); }       // AOCL_SETUP_NATIVE() at the top of every compress-side entry point

unsigned highbit(uint32_t v) { unsigned r = 0; while (v >>= 1) r++; return r; }

ZSTD_compressionParameters cparams_for(int level, unsigned long long srcSizeHint, size_t dictSize)
{   // ZSTD_getCParams_internal + ZSTD_adjustCParams_internal (zstd_compress.c:7295-7331, 1539-1633), mode ZSTD_cpm_unknown
    const unsigned long long kUnknown = 0ULL - 1;
    const bool unknown = srcSizeHint == kUnknown;
    const unsigned long long rSize = (unknown && dictSize == 0) ? kUnknown : srcSizeHint + dictSize + ((unknown && dictSize > 0) ? 500 : 0);
    const unsigned tableID = (rSize <= (256u << 10)) + (rSize <= (128u << 10)) + (rSize <= (16u << 10));
    int row = level == 0 ? 3 : (level < 0 ? 0 : (level > 22 ? 22 : level));
    const uint16_t* r = kLevelRows[g_aocl_table.load() ? 1 : 0][tableID][row];
    ZSTD_compressionParameters cp = { r[0], r[1], r[2], r[3], r[4], r[5], (ZSTD_strategy)r[6] };
    if (level < 0) { const int c = level < -(1 << 17) ? -(1 << 17) : level; cp.targetLength = (unsigned)(-c); }
    const unsigned long long maxWindowResize = 1ULL << 30;
    if (srcSizeHint <= maxWindowResize && dictSize <= maxWindowResize) {
        const uint32_t tSize = (uint32_t)(srcSizeHint + dictSize);
        const unsigned srcLog = tSize < (1u << 6) ? 6 : highbit(tSize - 1) + 1;
        if (cp.windowLog > srcLog) cp.windowLog = srcLog;
    }
    if (!unknown) {
        unsigned dawLog = cp.windowLog;                                   // ZSTD_dictAndWindowLog
        if (dictSize) {
            const unsigned long long ws = 1ULL << cp.windowLog, both = dictSize + ws;
            if (ws >= dictSize + srcSizeHint) dawLog = cp.windowLog;
            else if (both >= (1ULL << 31)) dawLog = 31;
            else dawLog = highbit((uint32_t)both - 1) + 1;
        }
        const unsigned cycleLog = cp.chainLog - ((unsigned)cp.strategy >= (unsigned)ZSTD_btlazy2 ? 1 : 0);
        if (cp.hashLog > dawLog + 1) cp.hashLog = dawLog + 1;
        if (cycleLog > dawLog) cp.chainLog -= cycleLog - dawLog;
    }
    if (cp.windowLog < 10) cp.windowLog = 10;
    return cp;
}

bool params_in_bounds(const ZSTD_compressionParameters& c)
{   // ZSTD_checkCParams (zstd_compress.c:1402-1413) with the 64-bit bounds of zstd.h:1201-1225
    return c.windowLog >= 10 && c.windowLog <= 31 && c.chainLog >= 6 && c.chainLog <= 30 && c.hashLog >= 6 && c.hashLog <= 30 &&
           c.searchLog >= 1 && c.searchLog <= 30 && c.minMatch >= 3 && c.minMatch <= 7 && c.targetLength <= (1u << 17) &&
           (int)c.strategy >= 1 && (int)c.strategy <= 9;
}

// which level's GPU parse (1..4) these parameters describe for an n-byte input; 0 = none.  Levels 1-2 are the "fast"
// strategy, 3-4 "double-fast" (for the sizes where the level table says so).
int gpu_level_for(const ZSTD_compressionParameters& c, size_t n)
{
    if (c.strategy != ZSTD_fast && c.strategy != ZSTD_dfast) return 0;
    if (c.strategy == ZSTD_fast && c.targetLength != 0) return 0;       // (a target length leaves literals raw and, above 1, widens the step: frame path)
    if (n == 0) return 1;                                                // an empty frame: no parameter shows in the output
    // the parameters must be those of a level for THIS input size (inputs above 128 KiB are then cut into 128 KiB pieces,
    // each compressed with that level's parameters for a piece: the documented multi-frame deviation)
    for (int level = 1; level <= 4; ++level) {
        const ZSTD_compressionParameters w = cparams_for(level, n, 0);
        if (w.strategy != c.strategy || w.hashLog != c.hashLog || w.minMatch != c.minMatch) continue;
        if (c.strategy == ZSTD_dfast && w.chainLog != c.chainLog) continue;
        return level;
    }
    return 0;
}

// Inputs above 128 KiB: 0 (default) = ONE frame of blocks, the reference's bytes, one wavefront's work
// from the first block to the last (every block needs the hash table the previous one left); 1 = 128 KiB pieces as
// concatenated frames compressed side by side (a valid stream every Zstd decoder reads, ~1000x the throughput on large
// inputs, not the reference's bytes).  AOCLGPU_ZSTD_LARGE_INPUTS=pieces in the environment selects 1 at start-up.
std::atomic<int> g_large_input_mode{-1};
int large_input_mode()
{
    int m = g_large_input_mode.load();
    if (m < 0) {
        const char* e = getenv("AOCLGPU_ZSTD_LARGE_INPUTS");
        m = (e && !strcmp(e, "pieces")) ? 1 : 0;
        g_large_input_mode.store(m);
    }
    return m;
}

size_t compress_with(const ZSTD_compressionParameters& c, const ZSTD_frameParameters& f, void* dst, size_t cap, const void* src, size_t n)
{
    if (!params_in_bounds(c)) return zerr(kErrParamOutOfBound);
    if (!f.contentSizeFlag) return zerr(kErrParamUnsupported);
    const bool covers = ((unsigned long long)1 << c.windowLog) >= n;     // single-segment frame
    const int level = gpu_level_for(c, n);
    // what the frame kernel parses: fast with any step (targetLength + 1 above 1: the levels below -1, zstd_fast.c:440),
    // double-fast; tables of at most 2^24 entries
    const bool frameable = n < ((size_t)1 << 31) && c.hashLog <= 24 &&
                           (c.strategy == ZSTD_fast || (c.strategy == ZSTD_dfast && c.chainLog <= 24));
    int64_t r;
    if (n <= (128u << 10) && level && covers && !f.checksumFlag) {
        // one block with a level's own parameters: the chunk kernels (hash table in LDS for "fast")
        const uint32_t flags = AOCLGPU_ZSTD_LEVEL(level) | (g_aocl_table.load() ? 0u : AOCLGPU_FLAG_ZSTD_UPSTREAM_TABLE);
        r = aoclgpu_compress_host_ex(AOCLGPU_ZSTD, src, n, dst, cap, flags);
    } else if (n > (128u << 10) && large_input_mode() == 1 && level && c.windowLog >= 17 && !f.checksumFlag) {
        // the caller asked for pieces: 128 KiB each, one frame per piece, side by side (the documented multi-frame stream)
        const uint32_t flags = AOCLGPU_ZSTD_LEVEL(level) | (g_aocl_table.load() ? 0u : AOCLGPU_FLAG_ZSTD_UPSTREAM_TABLE);
        r = aoclgpu_zstd_compress_host_frames(src, n, dst, cap, flags);
    } else if (frameable) {
        // any size, any such parameters: one frame of blocks, exactly the reference's.  minMatch 3 runs the 4-byte hash
        // (ZSTD_compressBlock_fast's switch, zstd_fast.c:764-790, and ZSTD_compressBlock_doubleFast's, zstd_double_fast.c:
        // "default: includes case 3")
        r = aoclgpu_zstd_compress_host_frame(src, n, dst, cap, (uint32_t)c.strategy, c.windowLog, c.hashLog, c.chainLog,
                                             c.minMatch < 4 ? 4u : c.minMatch, c.targetLength, f.checksumFlag ? 1u : 0u);
        if (r == 0) return zerr(kErrDstTooSmall);
    } else return zerr(kErrParamUnsupported);
    if (r > 0) return (size_t)r;
    if (r == 0) return zerr(cap < ZSTD_compressBound(n) ? kErrDstTooSmall : kErrParamUnsupported);   // 0 with room: no GPU parse for a piece
    return zerr(kErrGeneric);
}
}  // namespace

struct ZSTD_CCtx_s { int level; ZSTD_frameParameters f; };
struct ZSTD_DCtx_s { int unused; };

extern "C" {

// for the unified API's threaded branch (aocl_api.cpp): run the dispatcher as a compress entry point does, and tell which
// level table is installed (1 = AOCL, 0 = upstream)
int aoclgpu_zstd_level_table(void) { setup_native(); return g_aocl_table.load(); }

void aoclgpu_zstd_set_large_input_mode(int pieces) { g_large_input_mode.store(pieces ? 1 : 0); }
int aoclgpu_zstd_large_input_mode(void) { return large_input_mode(); }

unsigned ZSTD_versionNumber(void) { return 10505; }
const char* ZSTD_versionString(void) { return "1.5.5"; }
int ZSTD_minCLevel(void) { return -(1 << 17); }
int ZSTD_maxCLevel(void) { return 22; }
int ZSTD_defaultCLevel(void) { return 3; }
size_t ZSTD_compressBound(size_t n)
{   // zstd.h:228-230 ZSTD_COMPRESSBOUND
    if (n >= (sizeof(size_t) == 8 ? 0xFF00FF00FF00FF00ull : 0xFF00FF00u)) return zerr(kErrSrcSize);
    return n + (n >> 8) + (n < (128u << 10) ? (((128u << 10) - n) >> 11) : 0);
}
unsigned ZSTD_isError(size_t code) { return code > zerr(kErrMax); }                       // zstd_errors.h:100
const char* ZSTD_getErrorName(size_t code)
{   // error_private.c:21-67
    if (!ZSTD_isError(code)) return "No error detected";
    switch ((int)((size_t)0 - code)) {
    case kErrPrefix: return "Unknown frame descriptor";
    case kErrFrameParamUnsupported: return "Unsupported frame parameter";
    case kErrFrameWindowTooLarge: return "Frame requires too much memory for decoding";
    case kErrCorrupt: return "Data corruption detected";
    case 22: return "Restored data doesn't match checksum";
    case 24: return "Header of Literals' block doesn't respect format specification";
    case 30: return "Dictionary is corrupted";
    case 32: return "Dictionary mismatch";
    case kErrParamUnsupported: return "Unsupported parameter";
    case kErrParamOutOfBound: return "Parameter is out of bound";
    case kErrMemory: return "Allocation error : not enough memory";
    case kErrDstTooSmall: return "Destination buffer is too small";
    case kErrSrcSize: return "Src size is incorrect";
    case kErrDstNull: return "Operation on NULL destination buffer";
    default: return "Error (generic)";
    }
}

ZSTD_compressionParameters ZSTD_getCParams(int level, unsigned long long srcSizeHint, size_t dictSize)
{   // zstd_compress.c:7336-7340
    if (srcSizeHint == 0) srcSizeHint = 0ULL - 1;
    return cparams_for(level, srcSizeHint, dictSize);
}
ZSTD_parameters ZSTD_getParams(int level, unsigned long long srcSizeHint, size_t dictSize)
{   // zstd_compress.c:7346-7368
    ZSTD_parameters p; memset(&p, 0, sizeof(p));
    p.cParams = ZSTD_getCParams(level, srcSizeHint, dictSize);
    p.fParams.contentSizeFlag = 1;
    return p;
}

ZSTD_CCtx* ZSTD_createCCtx(void)
{
    ZSTD_CCtx* c = new (std::nothrow) ZSTD_CCtx_s();
    if (c) { c->level = 3; memset(&c->f, 0, sizeof(c->f)); c->f.contentSizeFlag = 1; }
    return c;
}
size_t ZSTD_freeCCtx(ZSTD_CCtx* c) { delete c; return 0; }
ZSTD_DCtx* ZSTD_createDCtx(void) { return new (std::nothrow) ZSTD_DCtx_s(); }
size_t ZSTD_freeDCtx(ZSTD_DCtx* d) { delete d; return 0; }

size_t ZSTD_CCtx_setParameter(ZSTD_CCtx* c, ZSTD_cParameter param, int value)
{   // zstd_compress.c:774-860: the parameters the one-shot calls of this library can honour are stored, the rest refused
    if (!c) return zerr(kErrGeneric);
    switch ((int)param) {
    case ZSTD_c_compressionLevel: c->level = value; return 0;
    case ZSTD_c_contentSizeFlag: c->f.contentSizeFlag = value != 0; return 0;
    case ZSTD_c_checksumFlag: c->f.checksumFlag = value != 0; return 0;
    case ZSTD_c_dictIDFlag: c->f.noDictIDFlag = value == 0; return 0;
    default: return zerr(kErrParamUnsupported);
    }
}

size_t ZSTD_compress_advanced(ZSTD_CCtx* cctx, void* dst, size_t dstCapacity, const void* src, size_t srcSize,
                              const void* dict, size_t dictSize, ZSTD_parameters params)
{   // zstd_compress.c:5388-5408
    setup_native();
    if (!cctx || (!src && srcSize > 0)) return zerr(kErrGeneric);
    if (!dst) return zerr(kErrDstNull);
    if (dict && dictSize) return zerr(kErrParamUnsupported);
    return compress_with(params.cParams, params.fParams, dst, dstCapacity, src, srcSize);
}

size_t ZSTD_compressCCtx(ZSTD_CCtx* cctx, void* dst, size_t dstCapacity, const void* src, size_t srcSize, int level)
{   // zstd_compress.c:5519-5531
    setup_native();
    if (!cctx || (!src && srcSize > 0)) return zerr(kErrGeneric);
    if (!dst) return zerr(kErrDstNull);
    const ZSTD_parameters p = ZSTD_getParams(level, srcSize, 0);
    return compress_with(p.cParams, p.fParams, dst, dstCapacity, src, srcSize);
}

size_t ZSTD_compress(void* dst, size_t dstCapacity, const void* src, size_t srcSize, int level)
{   // zstd_compress.c:5534-5560
    setup_native();
    if (!src && srcSize > 0) return zerr(kErrGeneric);
    if (!dst) return zerr(kErrDstNull);
    const ZSTD_parameters p = ZSTD_getParams(level, srcSize, 0);
    return compress_with(p.cParams, p.fParams, dst, dstCapacity, src, srcSize);
}

size_t ZSTD_decompress(void* dst, size_t dstCapacity, const void* src, size_t compressedSize)
{   // zstd_decompress.c:1252
    if ((!dst && dstCapacity) || !src) return zerr(kErrGeneric);
    const int64_t r = aoclgpu_zstd_decompress_host_frames(src, compressedSize, dst ? dst : (void*)"", dstCapacity);
    if (r >= 0) return (size_t)r;
    if (r <= AOCLGPU_HOST_FAIL / 2) return zerr(kErrGeneric);         // the GPU path itself failed (no device, allocation, launch)
    // the decoder reports the reference's own error code for the check that failed (zstd_kernels.hip zstd_decompress_wave / decode_block)
    switch ((int)-r) {
    case kErrPrefix: case kErrFrameParamUnsupported: case kErrFrameWindowTooLarge: case kErrCorrupt: case 22 /* checksum_wrong */:
    case 24 /* literals_headerWrong */: case 30 /* dictionary_corrupted */: case 32 /* dictionary_wrong */: case kErrDstTooSmall: case kErrSrcSize:
        return zerr((int)-r);
    default: return zerr(kErrCorrupt);
    }
}
size_t ZSTD_decompressDCtx(ZSTD_DCtx* dctx, void* dst, size_t dstCapacity, const void* src, size_t srcSize)
{   // zstd_decompress.c:1246-1250
    if (!dctx) return zerr(kErrGeneric);
    return ZSTD_decompress(dst, dstCapacity, src, srcSize);
}

unsigned long long ZSTD_getFrameContentSize(const void* src, size_t srcSize)
{   // zstd_decompress.c:601-624 via ZSTD_getFrameHeader (zstd_decompress.c:440-560); zstd frames only
    const unsigned long long kUnknown = 0ULL - 1, kError = 0ULL - 2;
    const uint8_t* p = (const uint8_t*)src;
    if (!p || srcSize < 5) return kError;
    const uint32_t magic = p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24);
    if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) return srcSize >= 8 ? 0 : kError;     // skippable frame (:609)
    if (magic != 0xFD2FB528u) return kError;
    const uint32_t fhd = p[4], dictID = fhd & 3, single = (fhd >> 5) & 1, fcsID = fhd >> 6;
    static const size_t did[4] = {0, 1, 2, 4}, fcs[4] = {0, 2, 4, 8};
    size_t pos = 5 + (single ? 0 : 1) + did[dictID];
    const size_t fl = (fcsID == 0) ? (single ? 1 : 0) : fcs[fcsID];
    if (fhd & 0x08) return kError;                                    // reserved bit
    if (srcSize < pos + fl) return kError;
    if (!single && (uint32_t)(p[5] >> 3) + 10 > 31) return kError;    // windowLog above ZSTD_WINDOWLOG_MAX (zstd_decompress.c:516-521)
    if (fl == 0) return kUnknown;
    unsigned long long v = 0;
    for (size_t i = 0; i < fl; ++i) v |= (unsigned long long)p[pos + i] << (8 * i);
    if (fl == 2) v += 256;
    return v;
}

// ---- frame inspection (host arithmetic on headers; zstd_decompress.c:601-790) ----
static size_t first_frame_size(const uint8_t* p, size_t n, unsigned long long* content /* may be NULL */)
{   // ZSTD_findFrameSizeInfo: compressed size of the frame (or skippable frame) at p; error code when it is not complete
    if (content) *content = 0ULL - 1;
    if (!p || n < 8) return zerr(kErrSrcSize);
    uint32_t magic; memcpy(&magic, p, 4);
    if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
        uint32_t sz; memcpy(&sz, p + 4, 4);
        if ((uint64_t)sz + 8 > n) return zerr(kErrSrcSize);
        if (content) *content = 0;
        return 8 + (size_t)sz;
    }
    if (magic != 0xFD2FB528u) return zerr(kErrPrefix);
    const uint32_t fhd = p[4], fcsId = fhd >> 6, single = (fhd >> 5) & 1, hasCk = (fhd >> 2) & 1, did = fhd & 3;
    size_t pos = 5 + (single ? 0 : 1) + (did == 3 ? 4 : did);
    const size_t fl = fcsId == 0 ? (single ? 1 : 0) : (fcsId == 1 ? 2 : (fcsId == 2 ? 4 : 8));
    if (pos + fl > n) return zerr(kErrSrcSize);
    if (fhd & 0x08) return zerr(kErrFrameParamUnsupported);           // reserved bit (ZSTD_getFrameHeader_advanced, zstd_decompress.c:500-502)
    if (!single && (uint32_t)(p[5] >> 3) + 10 > 31) return zerr(kErrFrameWindowTooLarge);   // (:516-521)
    if (content && fl) {
        unsigned long long v = 0;
        for (size_t i = 0; i < fl; ++i) v |= (unsigned long long)p[pos + i] << (8 * i);
        *content = fl == 2 ? v + 256 : v;
    }
    pos += fl;
    for (;;) {
        if (n - pos < 3) return zerr(kErrSrcSize);
        const uint32_t bh = p[pos] | ((uint32_t)p[pos + 1] << 8) | ((uint32_t)p[pos + 2] << 16);
        pos += 3;
        if (((bh >> 1) & 3) == 3) return zerr(kErrCorrupt);
        const size_t body = ((bh >> 1) & 3) == 1 ? 1 : (bh >> 3);
        if (body > n - pos) return zerr(kErrSrcSize);
        pos += body;
        if (bh & 1) break;
    }
    if (hasCk) { if (n - pos < 4) return zerr(kErrSrcSize); pos += 4; }
    return pos;
}
size_t ZSTD_findFrameCompressedSize(const void* src, size_t srcSize) { return first_frame_size((const uint8_t*)src, srcSize, nullptr); }
unsigned long long ZSTD_getDecompressedSize(const void* src, size_t srcSize)
{   // zstd_decompress.c:692-698: 0 when unknown or on error
    const unsigned long long r = ZSTD_getFrameContentSize(src, srcSize);
    return r >= 0ULL - 2 ? 0 : r;
}
unsigned long long ZSTD_decompressBound(const void* src, size_t srcSize)
{   // zstd_decompress.c:795-812: sum over the frames of the content size, or of (blocks x block size) when it is absent
    const uint8_t* p = (const uint8_t*)src;
    unsigned long long bound = 0;
    while (srcSize > 0) {
        unsigned long long content;
        const size_t fs = first_frame_size(p, srcSize, &content);
        if (ZSTD_isError(fs)) return 0ULL - 2;
        if (content == 0ULL - 1) {                                     // no content size: count the blocks
            unsigned long long blocks = 0; size_t pos = 5 + ((p[4] >> 5) & 1 ? 0 : 1) + ((p[4] & 3) == 3 ? 4 : (p[4] & 3));
            unsigned long long blockMax = 128u << 10;
            if (!((p[4] >> 5) & 1)) { const uint32_t wb = p[5], wl = (wb >> 3) + 10; unsigned long long w = 1ULL << wl; w += (w >> 3) * (wb & 7); if (w < blockMax) blockMax = w; }
            for (;;) { const uint32_t bh = p[pos] | ((uint32_t)p[pos + 1] << 8) | ((uint32_t)p[pos + 2] << 16); pos += 3 + (((bh >> 1) & 3) == 1 ? 1 : (bh >> 3)); blocks++; if (bh & 1) break; }
            content = blocks * blockMax;
        }
        bound += content;
        p += fs; srcSize -= fs;
    }
    return bound;
}

// ---- context-parameter API (zstd.h:1021,1057,1202): ZSTD_compress2 takes the level and frame flags stored in the context ----
size_t ZSTD_CCtx_reset(ZSTD_CCtx* c, ZSTD_ResetDirective reset)
{
    if (!c) return zerr(kErrGeneric);
    if ((int)reset & 2) { c->level = 3; memset(&c->f, 0, sizeof(c->f)); c->f.contentSizeFlag = 1; }
    return 0;
}
size_t ZSTD_DCtx_reset(ZSTD_DCtx* d, ZSTD_ResetDirective) { return d ? 0 : zerr(kErrGeneric); }
size_t ZSTD_compress2(ZSTD_CCtx* cctx, void* dst, size_t dstCapacity, const void* src, size_t srcSize)
{   // zstd_compress.c:6333-6360
    setup_native();
    if (!cctx || (!src && srcSize > 0)) return zerr(kErrGeneric);
    if (!dst) return zerr(kErrDstNull);
    const ZSTD_parameters p = ZSTD_getParams(cctx->level, srcSize, 0);
    return compress_with(p.cParams, cctx->f, dst, dstCapacity, src, srcSize);
}

// the reference's dispatcher hooks (zstd.h:423-452): there is no host ISA variant to pick here; the dispatcher's one
// visible effect - which level table ZSTD_getCParams reads - follows zstd_compress.c:7390-7463 (see the top of this file)
char* aocl_setup_zstd_encode(int optOff, int, size_t, size_t, size_t) { dispatcher_setup(optOff); return nullptr; }
char* aocl_setup_zstd_decode(int, int, size_t, size_t, size_t) { return nullptr; }
void aocl_destroy_zstd_encode(void) { std::lock_guard<std::mutex> lk(g_setup_mu); g_setup_ok.store(0); }   // (the table stays, :7459-7463)
void aocl_destroy_zstd_decode(void) {}

}  // extern "C"
