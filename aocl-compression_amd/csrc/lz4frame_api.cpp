// csrc/lz4frame_api.cpp -- LZ4 frame format (reference: algos/lz4/lz4frame.c, lz4frame.h, built with LZ4_FRAME_FORMAT_SUPPORT,
// CMakeLists.txt:311-325) on top of the GPU block codec.  Host buffers in, host buffers out, no CPU codec anywhere:
// the blocks of a frame are LZ4 blocks compressed / decoded by lz4_kernels.hip, this file only does the framing
// (header, block sizes, end mark) and the xxHash32 checksums of lz4frame.c:259-266, 758-761, 1004-1010.
//
//  * LZ4F_compressFrame (lz4frame.c:373-460), byte-identical to the reference's frame for the fast compressor
//    (compressionLevel < 3; negative levels are accelerations, :768).  LZ4F_blockIndependent - and any input that fits one
//    block, which the reference switches to independent itself (:394-395): every block on its own, all blocks of the frame
//    in one launch, with the reference's `srcSize - 1` output limit (LZ4F_makeBlock :740-763: a block that does not shrink is
//    stored).  LZ4F_blockLinked (the default): every block continues from the hash table of the one before and may match
//    into it (LZ4_compress_fast_continue in prefix mode), so the blocks of a frame are ONE wavefront's sequential work
//    (lz4_compress_linked_kernel) - ~35 MiB/s however large the input.  Callers who prefer throughput to the reference's
//    bytes switch that off (aoclgpu_lz4f_set_block_linking(0) or AOCLGPU_LZ4F_LINKED=independent): such frames are written
//    with independent blocks, flagged so in the FLG byte - valid frames every LZ4F decoder reads, a little larger.
//    LZ4HC levels (>= 3) return LZ4F_ERROR_compressionLevel_invalid.
//  * LZ4F_compressBegin / Update / flush / End: the reference's buffering rules, one launch per call (end of this file).
//  * LZ4F_decompress (lz4frame.c:1398-1860): a whole frame per call is decoded straight into the caller's buffer;
//    a frame that arrives in pieces (or a destination too small for all of it) is collected in the context, decoded when
//    its end mark has arrived, and handed out over the following calls - same calling convention as the reference (0 when
//    a frame is done, otherwise a hint > 0), but output only starts once the frame is complete.  Independent blocks are
//    decoded in one launch, one wavefront per block; linked blocks by one wavefront in order, each block copying from what
//    was decoded before it.  Header, block and content checksums are verified.  Skippable frames are skipped.
#include "../../include/aocl_native_api.h"
#include "../../include/aocl_gpu_codec.h"
#include <hip/hip_runtime.h>
#include "host_util.h"
#include <atomic>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <exception>
#include <vector>

namespace {

enum {   // lz4frame.h:511-533
    E_OK = 0, E_GENERIC, E_maxBlockSize_invalid, E_blockMode_invalid, E_contentChecksumFlag_invalid, E_compressionLevel_invalid,
    E_headerVersion_wrong, E_blockChecksum_invalid, E_reservedFlag_set, E_allocation_failed, E_srcSize_tooLarge,
    E_dstMaxSize_tooSmall, E_frameHeader_incomplete, E_frameType_unknown, E_frameSize_wrong, E_srcPtr_wrong,
    E_decompressionFailed, E_headerChecksum_invalid, E_contentChecksum_invalid, E_frameDecoding_alreadyStarted, E_maxCode
};
const char* const kErrNames[] = {
    "OK_NoError", "ERROR_GENERIC", "ERROR_maxBlockSize_invalid", "ERROR_blockMode_invalid", "ERROR_contentChecksumFlag_invalid",
    "ERROR_compressionLevel_invalid", "ERROR_headerVersion_wrong", "ERROR_blockChecksum_invalid", "ERROR_reservedFlag_set",
    "ERROR_allocation_failed", "ERROR_srcSize_tooLarge", "ERROR_dstMaxSize_tooSmall", "ERROR_frameHeader_incomplete",
    "ERROR_frameType_unknown", "ERROR_frameSize_wrong", "ERROR_srcPtr_wrong", "ERROR_decompressionFailed",
    "ERROR_headerChecksum_invalid", "ERROR_contentChecksum_invalid", "ERROR_frameDecoding_alreadyStarted", "ERROR_maxCode" };
size_t err(int code) { return (size_t)0 - (size_t)code; }                    // lz4frame.c:283-288

const uint32_t kMagic = 0x184D2204u, kSkipMagic = 0x184D2A50u;              // lz4frame.c:225-226
const size_t kMaxHeader = 19, kBH = 4, kBF = 4;                              // maxFHSize, BHSize, BFSize (:228-232)
const uint32_t kStored = 0x80000000u;                                        // LZ4F_BLOCKUNCOMPRESSED_FLAG

uint32_t rd32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
uint64_t rd64(const uint8_t* p) { return rd32(p) | ((uint64_t)rd32(p + 4) << 32); }
void wr32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
void wr64(uint8_t* p, uint64_t v) { wr32(p, (uint32_t)v); wr32(p + 4, (uint32_t)(v >> 32)); }

// xxHash32 (the published algorithm; the reference's copy is algos/lz4/xxhash.c) - checksums only, host side
uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
uint32_t xxh32(const void* data, size_t len, uint32_t seed)
{
    const uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
    const uint8_t* p = (const uint8_t*)data; const uint8_t* const end = p + len;
    uint32_t h;
    if (len >= 16) {
        uint32_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        const uint8_t* const limit = end - 16;
        do {
            v1 = rotl(v1 + rd32(p) * P2, 13) * P1; v2 = rotl(v2 + rd32(p + 4) * P2, 13) * P1;
            v3 = rotl(v3 + rd32(p + 8) * P2, 13) * P1; v4 = rotl(v4 + rd32(p + 12) * P2, 13) * P1;
            p += 16;
        } while (p <= limit);
        h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
    } else h = seed + P5;
    h += (uint32_t)len;
    while (p + 4 <= end) { h = rotl(h + rd32(p) * P3, 17) * P4; p += 4; }
    while (p < end) { h = rotl(h + *p * P5, 11) * P1; p++; }
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
    return h;
}

size_t block_size(int id)
{   // LZ4F_getBlockSize (lz4frame.c:290-298): 0 = default = 64 KB
    if (id == 0) id = 4;
    if (id < 4 || id > 7) return 0;
    return (size_t)1 << (8 + 2 * id);
}
int optimal_bsid(int requested, size_t srcSize)
{   // LZ4F_optimalBSID (lz4frame.c:304-316)
    int proposed = 4; size_t maxBlock = 64u << 10;
    while (requested > proposed) {
        if (srcSize <= maxBlock) return proposed;
        proposed++; maxBlock <<= 2;
    }
    return requested;
}

struct DevBuf {   // device allocations released on scope exit
    std::vector<void*> p;
    ~DevBuf() { for (void* q : p) if (q) (void)hipFree(q); }
    void* get(size_t bytes) { void* q = nullptr; if (hipMalloc(&q, bytes ? bytes : 1) != hipSuccess) return nullptr; p.push_back(q); return q; }
};

// upload the explicit-mode geometry arrays of a batch; returns false on failure
bool upload_geometry(DevBuf& D, aoclgpu_batch& b, const std::vector<uint64_t>& so, const std::vector<uint32_t>& sl,
                     const std::vector<uint64_t>& dofs, const std::vector<uint32_t>& dc)
{
    const size_t n = so.size();
    uint64_t* d_so = (uint64_t*)D.get(n * 8); uint32_t* d_sl = (uint32_t*)D.get(n * 4);
    uint64_t* d_do = (uint64_t*)D.get(n * 8); uint32_t* d_dc = (uint32_t*)D.get(n * 4);
    int32_t* d_ol = (int32_t*)D.get(n * 4);
    if (!d_so || !d_sl || !d_do || !d_dc || !d_ol) return false;
    if (hipMemcpy(d_so, so.data(), n * 8, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_sl, sl.data(), n * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_do, dofs.data(), n * 8, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_dc, dc.data(), n * 4, hipMemcpyHostToDevice) != hipSuccess) return false;
    b.src_off = d_so; b.src_len = d_sl; b.dst_off = d_do; b.dst_cap = d_dc; b.out_len = d_ol; b.nchunks = (uint32_t)n;
    return true;
}


// xxHash32 over data that arrives in pieces (content checksum of the streaming compressor)
struct Xxh32Stream {
    uint64_t total; uint32_t v[4]; uint8_t mem[16]; uint32_t memsize;
    void reset()
    {
        const uint32_t P1 = 2654435761u, P2 = 2246822519u;
        total = 0; v[0] = P1 + P2; v[1] = P2; v[2] = 0; v[3] = 0u - P1; memsize = 0;
    }
    static uint32_t round(uint32_t acc, uint32_t in) { return rotl(acc + in * 2246822519u, 13) * 2654435761u; }
    void update(const void* data, size_t len)
    {
        const uint8_t* p = (const uint8_t*)data; const uint8_t* const end = p + len;
        total += len;
        if (memsize + len < 16) { memcpy(mem + memsize, p, len); memsize += (uint32_t)len; return; }
        if (memsize) {
            memcpy(mem + memsize, p, 16 - memsize);
            for (int i = 0; i < 4; ++i) v[i] = round(v[i], rd32(mem + 4 * i));
            p += 16 - memsize; memsize = 0;
        }
        while (p + 16 <= end) { for (int i = 0; i < 4; ++i) v[i] = round(v[i], rd32(p + 4 * i)); p += 16; }
        if (p < end) { memcpy(mem, p, (size_t)(end - p)); memsize = (uint32_t)(end - p); }
    }
    uint32_t digest() const
    {
        const uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
        uint32_t h = total >= 16 ? rotl(v[0], 1) + rotl(v[1], 7) + rotl(v[2], 12) + rotl(v[3], 18) : v[2] + P5;
        h += (uint32_t)total;
        const uint8_t* p = mem; const uint8_t* const end = mem + memsize;
        while (p + 4 <= end) { h = rotl(h + rd32(p) * P3, 17) * P4; p += 4; }
        while (p < end) { h = rotl(h + *p * P5, 11) * P1; p++; }
        h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
        return h;
    }
};

// LZ4F_blockLinked in one-shot frames: 1 (default) = honoured, the reference's bytes, serial; 0 = independent blocks instead
std::atomic<int> g_block_linking{-1};
bool block_linking()
{
    int m = g_block_linking.load();
    if (m < 0) {
        const char* e = getenv("AOCLGPU_LZ4F_LINKED");
        m = (e && !strcmp(e, "independent")) ? 0 : 1;
        g_block_linking.store(m);
    }
    return m != 0;
}

struct BlockRef { const uint8_t* p; uint32_t len; };

// LZ4F_makeBlock (lz4frame.c:740-763) for a list of blocks, all compressed in ONE launch: each block on its own, output limit
// len - 1, stored when it does not shrink; writes {size, data, [checksum]} per block at op.  Returns the bytes written or an
// error code (test with LZ4F_isError).
size_t make_blocks(const std::vector<BlockRef>& blks, size_t bs, bool blockCrc, uint8_t* op, int acceleration = 1)
{
    const size_t nb = blks.size();
    if (!nb) return 0;
    uint8_t* const op0 = op;
    DevBuf D;
    const size_t slot = (bs + 15) & ~(size_t)15;
    size_t total = 0;
    for (const BlockRef& b : blks) total += b.len;
    uint8_t* d_src = (uint8_t*)D.get(total); uint8_t* d_dst = (uint8_t*)D.get(nb * slot);
    if (!d_src || !d_dst) return err(E_allocation_failed);
    std::vector<uint64_t> so(nb), dofs(nb); std::vector<uint32_t> sl(nb), dc(nb);
    size_t off = 0;
    for (size_t i = 0; i < nb; ++i) {
        // consecutive blocks of one buffer travel in one copy
        size_t j = i, run = blks[i].len;
        while (j + 1 < nb && blks[j + 1].p == blks[j].p + blks[j].len) { ++j; run += blks[j].len; }
        if (hipMemcpy(d_src + off, blks[i].p, run, hipMemcpyHostToDevice) != hipSuccess) return err(E_allocation_failed);
        for (size_t k = i; k <= j; ++k) { so[k] = off; sl[k] = blks[k].len; dofs[k] = k * slot; dc[k] = blks[k].len - 1; off += blks[k].len; }
        i = j;
    }
    aoclgpu_batch b; memset(&b, 0, sizeof(b));
    b.src = d_src; b.dst = d_dst;
    if (!upload_geometry(D, b, so, sl, dofs, dc)) return err(E_allocation_failed);
    b.reserved = AOCLGPU_LZ4_ACCELERATION(acceleration);
    if (aoclgpu_lz4_compress_chunks(&b, nullptr) != AOCLGPU_OK) return err(E_GENERIC);
    std::vector<int32_t> ol(nb);
    std::vector<uint8_t> slots(nb * slot);
    if (hipMemcpy(ol.data(), b.out_len, nb * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(slots.data(), d_dst, nb * slot, hipMemcpyDeviceToHost) != hipSuccess) return err(E_GENERIC);
    for (size_t i = 0; i < nb; ++i) {
        uint32_t c = ol[i] > 0 ? (uint32_t)ol[i] : 0;
        if (c == 0) { c = sl[i]; wr32(op, c | kStored); memcpy(op + kBH, blks[i].p, c); }                       // stored block
        else { wr32(op, c); memcpy(op + kBH, slots.data() + i * slot, c); }
        if (blockCrc) wr32(op + kBH + c, xxh32(op + kBH, c, 0));
        op += kBH + c + (blockCrc ? kBF : 0);
    }
    return (size_t)(op - op0);
}

// The blocks of a frame with LZ4F_blockLinked, one-shot (lz4frame.c:373-419): compressed in order by one wavefront that keeps
// the hash table from block to block (aoclgpu_lz4_compress_chunks with AOCLGPU_FLAG_LZ4_LINKED).  Same output layout as
// make_blocks.
size_t make_linked_blocks(const uint8_t* src, size_t n, size_t bs, bool blockCrc, int acceleration, uint8_t* op)
{
    if (!n) return 0;
    uint8_t* const op0 = op;
    const size_t nb = (n + bs - 1) / bs, slot = (bs + 15) & ~(size_t)15;
    DevBuf D;
    uint8_t* d_src = (uint8_t*)D.get(n); uint8_t* d_dst = (uint8_t*)D.get(nb * slot); int32_t* d_len = (int32_t*)D.get(nb * 4);
    if (!d_src || !d_dst || !d_len) return err(E_allocation_failed);
    if (hipMemcpy(d_src, src, n, hipMemcpyHostToDevice) != hipSuccess) return err(E_allocation_failed);
    aoclgpu_batch b; memset(&b, 0, sizeof(b));
    b.src = d_src; b.dst = d_dst; b.out_len = d_len; b.nchunks = (uint32_t)nb; b.total_bytes = n; b.chunk_bytes = bs; b.slot_bytes = slot;
    b.reserved = AOCLGPU_FLAG_LZ4_LINKED | AOCLGPU_LZ4_ACCELERATION(acceleration);
    if (aoclgpu_lz4_compress_chunks(&b, nullptr) != AOCLGPU_OK) return err(E_GENERIC);
    std::vector<int32_t> ol(nb);
    std::vector<uint8_t> slots(nb * slot);
    if (hipMemcpy(ol.data(), d_len, nb * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(slots.data(), d_dst, nb * slot, hipMemcpyDeviceToHost) != hipSuccess) return err(E_GENERIC);
    for (size_t i = 0; i < nb; ++i) {
        const uint32_t len = (uint32_t)(n - i * bs < bs ? n - i * bs : bs);
        uint32_t c = ol[i] > 0 ? (uint32_t)ol[i] : 0;
        if (c == 0) { c = len; wr32(op, c | kStored); memcpy(op + kBH, src + i * bs, c); }                    // stored block
        else { wr32(op, c); memcpy(op + kBH, slots.data() + i * slot, c); }
        if (blockCrc) wr32(op + kBH + c, xxh32(op + kBH, c, 0));
        op += kBH + c + (blockCrc ? kBF : 0);
    }
    return (size_t)(op - op0);
}

// frame header (lz4frame.c:670-699).  Returns its size.
size_t write_header(uint8_t* dst, const LZ4F_preferences_t& prefs, bool independent = true)
{
    uint8_t* op = dst;
    wr32(op, kMagic); op += 4;
    uint8_t* const hs = op;
    *op++ = (uint8_t)((1 << 6) | ((independent ? 1 : 0) << 5) | ((prefs.frameInfo.blockChecksumFlag & 1) << 4) | ((prefs.frameInfo.contentSize > 0) << 3) |
                      ((prefs.frameInfo.contentChecksumFlag & 1) << 2) | (prefs.frameInfo.dictID > 0));
    *op++ = (uint8_t)(((int)prefs.frameInfo.blockSizeID & 7) << 4);
    if (prefs.frameInfo.contentSize) { wr64(op, prefs.frameInfo.contentSize); op += 8; }
    if (prefs.frameInfo.dictID) { wr32(op, prefs.frameInfo.dictID); op += 4; }
    *op = (uint8_t)(xxh32(hs, (size_t)(op - hs), 0) >> 8); op++;
    return (size_t)(op - dst);
}

// LZ4F_compressBound_internal (lz4frame.c:324-349)
size_t bound_internal(size_t srcSize, const LZ4F_preferences_t* prefs, size_t alreadyBuffered)
{
    const bool worst = prefs == nullptr;
    const int bsid = prefs ? (int)prefs->frameInfo.blockSizeID : 0;
    const size_t bs = block_size(bsid);
    if (!bs) return err(E_maxBlockSize_invalid);
    const unsigned flush = (prefs ? prefs->autoFlush : 0) | (srcSize == 0);
    const size_t maxBuffered = bs - 1, buffered = alreadyBuffered < maxBuffered ? alreadyBuffered : maxBuffered;
    const size_t maxSrc = srcSize + buffered, full = maxSrc / bs, partial = maxSrc & (bs - 1), last = flush ? partial : 0;
    const size_t nb = full + (last > 0);
    const size_t blockCrc = (worst || prefs->frameInfo.blockChecksumFlag) ? kBF : 0;
    const size_t frameEnd = kBH + ((worst || prefs->frameInfo.contentChecksumFlag) ? kBF : 0);
    return (kBH + blockCrc) * nb + bs * full + last + frameEnd;
}

struct FrameHeader { int bsid; bool independent, blockChecksum, contentChecksum, hasSize, hasDict; uint64_t contentSize; uint32_t dictID; size_t bytes; };

// LZ4F_decodeHeader (lz4frame.c:1190-1290) for a regular frame; returns 0 or an error code
size_t parse_header(const uint8_t* p, size_t n, FrameHeader* h)
{
    if (n < 7) return err(E_frameHeader_incomplete);
    if (rd32(p) != kMagic) return err(E_frameType_unknown);
    const uint32_t flg = p[4], bd = p[5];
    if (((flg >> 6) & 3) != 1) return err(E_headerVersion_wrong);
    if ((flg >> 1) & 1) return err(E_reservedFlag_set);
    h->independent = (flg >> 5) & 1; h->blockChecksum = (flg >> 4) & 1; h->hasSize = (flg >> 3) & 1;
    h->contentChecksum = (flg >> 2) & 1; h->hasDict = flg & 1;
    h->bytes = 7 + (h->hasSize ? 8 : 0) + (h->hasDict ? 4 : 0);
    if (n < h->bytes) return err(E_frameHeader_incomplete);
    if ((bd >> 7) & 1) return err(E_reservedFlag_set);
    h->bsid = (bd >> 4) & 7;
    if (h->bsid < 4) return err(E_maxBlockSize_invalid);
    if (bd & 15) return err(E_reservedFlag_set);
    if (((xxh32(p + 4, h->bytes - 5, 0) >> 8) & 0xFF) != p[h->bytes - 1]) return err(E_headerChecksum_invalid);
    h->contentSize = h->hasSize ? rd64(p + 6) : 0;
    h->dictID = h->hasDict ? rd32(p + 6 + (h->hasSize ? 8 : 0)) : 0;
    return 0;
}

}  // namespace

struct LZ4F_dctx_s {
    unsigned version; LZ4F_frameInfo_t info; bool haveInfo;
    // piecewise decoding: the frame is collected here until its end mark (and checksum) has arrived, decoded in one go,
    // and handed out as the caller's buffers allow
    std::vector<uint8_t> in, out;
    size_t outPos;
};

extern "C" {
void aoclgpu_lz4f_set_block_linking(int linked) { g_block_linking.store(linked ? 1 : 0); }
int aoclgpu_lz4f_block_linking(void) { return block_linking() ? 1 : 0; }

unsigned LZ4F_isError(size_t code) { return code > err(E_maxCode); }          // lz4frame.c:270-273
const char* LZ4F_getErrorName(size_t code)
{
    if (LZ4F_isError(code)) return kErrNames[(size_t)0 - code];
    return "Unspecified error code";
}
unsigned LZ4F_getVersion(void) { return 100; }                                // LZ4F_VERSION
int LZ4F_compressionLevel_max(void) { return 12; }                            // LZ4HC_CLEVEL_MAX

size_t LZ4F_compressFrameBound(size_t srcSize, const LZ4F_preferences_t* prefs)
{   // lz4frame.c:324-361 with autoFlush = 1
    const int bsid = prefs ? (int)prefs->frameInfo.blockSizeID : 0;
    const size_t bs = block_size(bsid);
    if (!bs) return err(E_maxBlockSize_invalid);
    const size_t full = srcSize / bs, last = srcSize & (bs - 1), nb = full + (last > 0);
    const size_t blockCrc = (prefs && prefs->frameInfo.blockChecksumFlag) ? kBF : 0;
    const size_t frameEnd = kBH + ((prefs && prefs->frameInfo.contentChecksumFlag) ? kBF : 0);
    return kMaxHeader + (kBH + blockCrc) * nb + bs * full + last + frameEnd;
}

static size_t LZ4F_compressFrame_impl(void* dstBuffer, size_t dstCapacity, const void* srcBuffer, size_t srcSize, const LZ4F_preferences_t* prefsPtr)
{
    LZ4F_preferences_t prefs;
    if (prefsPtr) prefs = *prefsPtr; else memset(&prefs, 0, sizeof(prefs));
    if (!dstBuffer || (!srcBuffer && srcSize)) return err(E_GENERIC);
    if (prefs.compressionLevel >= 3) return err(E_compressionLevel_invalid);                                  // only the fast compressor
    const int acceleration = prefs.compressionLevel < 0 ? -prefs.compressionLevel + 1 : 1;                    // LZ4F_compressBlock :768
    if (prefs.frameInfo.contentSize != 0) prefs.frameInfo.contentSize = srcSize;                              // :389-390
    prefs.frameInfo.blockSizeID = (LZ4F_blockSizeID_t)optimal_bsid((int)prefs.frameInfo.blockSizeID, srcSize);
    if (!block_size((int)prefs.frameInfo.blockSizeID)) return err(E_maxBlockSize_invalid);
    if (dstCapacity < LZ4F_compressFrameBound(srcSize, &prefs)) return err(E_dstMaxSize_tooSmall);           // :400-401
    if (prefs.frameInfo.blockSizeID == 0) prefs.frameInfo.blockSizeID = LZ4F_max64KB;                         // :612
    const size_t bs = block_size((int)prefs.frameInfo.blockSizeID);
    if (srcSize > ((size_t)0xFFFFFFFFu - 64) * 1) return err(E_srcSize_tooLarge);
    uint8_t* const dst = (uint8_t*)dstBuffer; uint8_t* op = dst;
    // one block => independent (:394-395); linked blocks unless the caller switched them off
    const bool linked = prefs.frameInfo.blockMode == LZ4F_blockLinked && srcSize > bs && block_linking() && srcSize < 0x7E000000ull;
    op += write_header(op, prefs, !linked);
    // ---- blocks: one launch for the whole frame ----
    if (linked) {
        const size_t w = make_linked_blocks((const uint8_t*)srcBuffer, srcSize, bs, prefs.frameInfo.blockChecksumFlag != 0, acceleration, op);
        if (LZ4F_isError(w)) return w;
        op += w;
    } else {
        std::vector<BlockRef> blks;
        for (size_t o = 0; o < srcSize; o += bs) blks.push_back(BlockRef{(const uint8_t*)srcBuffer + o, (uint32_t)(srcSize - o < bs ? srcSize - o : bs)});
        const size_t w = make_blocks(blks, bs, prefs.frameInfo.blockChecksumFlag != 0, op, acceleration);
        if (LZ4F_isError(w)) return w;
        op += w;
    }
    wr32(op, 0); op += 4;                                                                                     // end mark (:1002)
    if (prefs.frameInfo.contentChecksumFlag) { wr32(op, xxh32(srcBuffer, srcSize, 0)); op += 4; }
    return (size_t)(op - dst);
}

size_t LZ4F_createDecompressionContext(LZ4F_dctx** dctxPtr, unsigned version)
{
    if (!dctxPtr) return err(E_GENERIC);
    LZ4F_dctx* d = new (std::nothrow) LZ4F_dctx();
    if (!d) { *dctxPtr = nullptr; return err(E_allocation_failed); }
    d->version = version; d->haveInfo = false; d->outPos = 0; memset(&d->info, 0, sizeof(d->info)); *dctxPtr = d;
    return 0;
}
size_t LZ4F_freeDecompressionContext(LZ4F_dctx* dctx) { delete dctx; return 0; }
void LZ4F_resetDecompressionContext(LZ4F_dctx* dctx) { if (dctx) { dctx->haveInfo = false; dctx->in.clear(); dctx->out.clear(); dctx->outPos = 0; } }

size_t LZ4F_headerSize(const void* src, size_t srcSize)
{   // lz4frame.c:1150-1180
    if (!src) return err(E_srcPtr_wrong);
    if (srcSize < 5) return err(E_frameHeader_incomplete);
    const uint8_t* p = (const uint8_t*)src;
    if ((rd32(p) & 0xFFFFFFF0u) == kSkipMagic) return 8;
    if (rd32(p) != kMagic) return err(E_frameType_unknown);
    return 7 + ((p[4] >> 3) & 1 ? 8 : 0) + (p[4] & 1 ? 4 : 0);
}

static size_t LZ4F_getFrameInfo_impl(LZ4F_dctx* dctx, LZ4F_frameInfo_t* info, const void* src, size_t* srcSizePtr)
{   // lz4frame.c:1310-1360: decodes the header, consumes it, hints at the next block header
    if (!dctx || !info || !src || !srcSizePtr) return err(E_GENERIC);
    FrameHeader h;
    const size_t r = parse_header((const uint8_t*)src, *srcSizePtr, &h);
    if (LZ4F_isError(r)) { *srcSizePtr = 0; return r; }
    memset(info, 0, sizeof(*info));
    info->blockSizeID = (LZ4F_blockSizeID_t)h.bsid; info->blockMode = h.independent ? LZ4F_blockIndependent : LZ4F_blockLinked;
    info->contentChecksumFlag = h.contentChecksum ? LZ4F_contentChecksumEnabled : LZ4F_noContentChecksum;
    info->frameType = LZ4F_frame; info->contentSize = h.contentSize; info->dictID = h.dictID;
    info->blockChecksumFlag = h.blockChecksum ? LZ4F_blockChecksumEnabled : LZ4F_noBlockChecksum;
    dctx->info = *info; dctx->haveInfo = true;
    dctx->in.assign((const uint8_t*)src, (const uint8_t*)src + h.bytes);      // the header is consumed here: LZ4F_decompress continues behind it
    dctx->out.clear(); dctx->outPos = 0;
    *srcSizePtr = h.bytes;
    return kBH;
}

// Upper bound of what a block of `len` compressed bytes can decode to: the block size of the frame, but never more than
// LZ4's worst-case expansion (a sequence byte adds at most 255 output bytes) - a frame made of thousands of tiny blocks
// must not make the decoder reserve blockSize for each of them.
static inline size_t block_cap(size_t bs, uint32_t len, bool stored)
{
    if (stored) return len;
    const size_t e = (size_t)len * 255u;
    return e < bs ? e : bs;
}

// one whole frame (or skippable frame) at srcBuffer -> dstBuffer; 0 and the consumed / produced sizes, or an error code
static size_t decode_whole(void* dstBuffer, size_t* dstSizePtr, const void* srcBuffer, size_t* srcSizePtr)
{
    const uint8_t* const src = (const uint8_t*)srcBuffer; const size_t n = *srcSizePtr, cap = *dstSizePtr;
    *srcSizePtr = 0; *dstSizePtr = 0;
    if (n >= 8 && (rd32(src) & 0xFFFFFFF0u) == kSkipMagic) {                                                  // skippable frame (:1236-1246)
        const size_t sz = rd32(src + 4);
        if (n < 8 + sz) return err(E_frameSize_wrong);
        *srcSizePtr = 8 + sz; return 0;
    }
    FrameHeader h;
    { const size_t r = parse_header(src, n, &h); if (LZ4F_isError(r)) return r; }
    const size_t bs = block_size(h.bsid);
    // ---- walk the block headers up to the end mark ----
    struct Blk { size_t off; uint32_t len; bool stored; };
    std::vector<Blk> blks;
    size_t ip = h.bytes;
    for (;;) {
        if (n - ip < kBH) return err(E_frameSize_wrong);
        const uint32_t v = rd32(src + ip); ip += kBH;
        if (v == 0) break;
        const uint32_t len = v & 0x7FFFFFFFu;
        if (len > bs) return err(E_maxBlockSize_invalid);                                                       // :1541-1542
        if (n - ip < (size_t)len + (h.blockChecksum ? kBF : 0)) return err(E_frameSize_wrong);
        if (h.blockChecksum && xxh32(src + ip, len, 0) != rd32(src + ip + len)) return err(E_blockChecksum_invalid);
        blks.push_back(Blk{ip, len, (v & kStored) != 0});
        ip += len + (h.blockChecksum ? kBF : 0);
    }
    uint32_t wantCrc = 0;
    if (h.contentChecksum) { if (n - ip < kBF) return err(E_frameSize_wrong); wantCrc = rd32(src + ip); ip += kBF; }
    const size_t nb = blks.size();
    size_t produced = 0;
    if (nb) {
        if (!dstBuffer) return err(E_dstMaxSize_tooSmall);
        DevBuf D;
        const size_t payload = ip - h.bytes;
        uint8_t* d_src = (uint8_t*)D.get(payload);
        if (!d_src || hipMemcpy(d_src, src + h.bytes, payload, hipMemcpyHostToDevice) != hipSuccess) return err(E_allocation_failed);
        std::vector<uint64_t> so(nb), dofs(nb); std::vector<uint32_t> sl(nb), dc(nb);
        std::vector<int32_t> ol(nb);
        aoclgpu_batch b; memset(&b, 0, sizeof(b));
        b.src = d_src;
        if (h.independent) {
            // every block into its own slot, then the slots are joined in order (blocks need not be full)
            size_t slotBytes = 0;
            for (size_t i = 0; i < nb; ++i) {
                so[i] = blks[i].off - h.bytes; sl[i] = blks[i].stored ? 1 : blks[i].len;
                dofs[i] = slotBytes; dc[i] = (uint32_t)(blks[i].stored ? 16 : block_cap(bs, blks[i].len, false));
                slotBytes += ((size_t)dc[i] + 15) & ~(size_t)15;
            }
            uint8_t* d_slots = (uint8_t*)D.get(slotBytes);
            if (!d_slots) return err(E_allocation_failed);
            // stored blocks: a one-byte dummy block for the launch (decodes to nothing useful), overwritten below
            b.dst = d_slots;
            if (!upload_geometry(D, b, so, sl, dofs, dc)) return err(E_allocation_failed);
            if (aoclgpu_lz4_decompress_chunks(&b, nullptr) != AOCLGPU_OK) return err(E_GENERIC);
            if (hipMemcpy(ol.data(), b.out_len, nb * 4, hipMemcpyDeviceToHost) != hipSuccess) return err(E_GENERIC);
            for (size_t i = 0; i < nb; ++i) {
                if (blks[i].stored) ol[i] = (int32_t)blks[i].len;
                else if (ol[i] < 0) return err(E_decompressionFailed);
                if (produced + (size_t)ol[i] > cap) return err(E_dstMaxSize_tooSmall);
                if (blks[i].stored) memcpy((uint8_t*)dstBuffer + produced, src + blks[i].off, blks[i].len);
                else if (hipMemcpy((uint8_t*)dstBuffer + produced, d_slots + dofs[i], (size_t)ol[i], hipMemcpyDeviceToHost) != hipSuccess) return err(E_GENERIC);
                produced += (size_t)ol[i];
            }
        } else {
            // linked blocks: one wavefront, in order, into one contiguous device buffer
            size_t sumCap = 0;
            for (size_t i = 0; i < nb; ++i) {
                so[i] = blks[i].off - h.bytes; sl[i] = blks[i].len | (blks[i].stored ? kStored : 0u); dofs[i] = 0;
                dc[i] = (uint32_t)block_cap(bs, blks[i].len, blks[i].stored); sumCap += dc[i];
            }
            const size_t total = sumCap < cap ? sumCap : cap;
            if (total > 0x7FFFFFFFu) return err(E_srcSize_tooLarge);
            uint8_t* d_out = (uint8_t*)D.get(total ? total : 1);
            if (!d_out) return err(E_allocation_failed);
            b.dst = d_out; b.reserved = AOCLGPU_FLAG_LZ4_LINKED; b.total_bytes = total;
            if (!upload_geometry(D, b, so, sl, dofs, dc)) return err(E_allocation_failed);
            if (aoclgpu_lz4_decompress_chunks(&b, nullptr) != AOCLGPU_OK) return err(E_GENERIC);
            if (hipMemcpy(ol.data(), b.out_len, nb * 4, hipMemcpyDeviceToHost) != hipSuccess) return err(E_GENERIC);
            for (size_t i = 0; i < nb; ++i) { if (ol[i] < 0) return err(sumCap > cap ? E_dstMaxSize_tooSmall : E_decompressionFailed); produced += (size_t)ol[i]; }
            if (produced > cap) return err(E_dstMaxSize_tooSmall);
            if (hipMemcpy(dstBuffer, d_out, produced, hipMemcpyDeviceToHost) != hipSuccess) return err(E_GENERIC);
        }
    }
    if (h.hasSize && h.contentSize != produced) return err(E_frameSize_wrong);                                 // :1722-1723
    if (h.contentChecksum && xxh32(dstBuffer, produced, 0) != wantCrc) return err(E_contentChecksum_invalid);
    *srcSizePtr = ip; *dstSizePtr = produced;
    return 0;
}

}  // extern "C"

// Length of the frame that starts at p, if all of it is within the n bytes given: 0 = not complete yet (*need = bytes
// still missing for the next header / block, at least 1), or an error code.  *upper = upper bound of the decoded size.
static size_t frame_extent(const uint8_t* p, size_t n, size_t* need, size_t* upper)
{
    *need = 1; *upper = 0;
    if (n < 4) { *need = 4 - n; return 0; }
    if ((rd32(p) & 0xFFFFFFF0u) == kSkipMagic) {
        if (n < 8) { *need = 8 - n; return 0; }
        const size_t tot = 8 + (size_t)rd32(p + 4);
        if (n < tot) { *need = tot - n; return 0; }
        return tot;
    }
    if (rd32(p) != kMagic) return err(E_frameType_unknown);
    if (n < 7) { *need = 7 - n; return 0; }
    const size_t hb = 7 + ((p[4] >> 3) & 1 ? 8 : 0) + (p[4] & 1 ? 4 : 0);
    if (n < hb) { *need = hb - n; return 0; }
    FrameHeader h;
    { const size_t r = parse_header(p, n, &h); if (LZ4F_isError(r)) return r; }
    const size_t bs = block_size(h.bsid);
    size_t ip = h.bytes;
    for (;;) {
        if (n - ip < kBH) { *need = kBH - (n - ip); return 0; }
        const uint32_t v = rd32(p + ip); ip += kBH;
        if (v == 0) break;
        const size_t len = (v & 0x7FFFFFFFu) + (h.blockChecksum ? kBF : 0);
        if ((v & 0x7FFFFFFFu) > bs) return err(E_maxBlockSize_invalid);
        if (n - ip < len) { *need = len - (n - ip); return 0; }
        ip += len; *upper += block_cap(bs, v & 0x7FFFFFFFu, (v & kStored) != 0);
    }
    if (h.contentChecksum) { if (n - ip < kBF) { *need = kBF - (n - ip); return 0; } ip += kBF; }
    return ip;
}

static size_t LZ4F_decompress_impl(LZ4F_dctx* dctx, void* dstBuffer, size_t* dstSizePtr, const void* srcBuffer, size_t* srcSizePtr,
                                   const LZ4F_decompressOptions_t* opts)
{   // lz4frame.c:1398-1860.  Returns 0 when a frame is completely decoded and delivered, otherwise a hint (> 0) of how many
    // source bytes the next call should bring (output still pending: 1), or an error code.
    (void)opts;
    if (!dctx || !dstSizePtr || !srcSizePtr || (!srcBuffer && *srcSizePtr)) return err(E_GENERIC);
    const size_t n = *srcSizePtr, cap = *dstSizePtr;
    size_t consumed = 0;
    if (dctx->outPos >= dctx->out.size()) {                                  // nothing decoded is waiting: collect / decode
        dctx->out.clear(); dctx->outPos = 0;
        size_t need = 1, upper = 0;
        if (dctx->in.empty()) {
            // the usual call: a whole frame in srcBuffer, room for all of its content in dstBuffer
            const size_t ext = frame_extent((const uint8_t*)srcBuffer, n, &need, &upper);
            if (LZ4F_isError(ext)) { *srcSizePtr = 0; *dstSizePtr = 0; return ext; }
            if (ext) {
                size_t sn = n, dn = cap;
                const size_t r = decode_whole(dstBuffer, &dn, srcBuffer, &sn);
                if (r == 0) { *srcSizePtr = sn; *dstSizePtr = dn; return 0; }
                if (r != err(E_dstMaxSize_tooSmall)) { *srcSizePtr = 0; *dstSizePtr = 0; return r; }
            }
        }
        // piecewise: append, and decode into the context once the frame is complete
        const size_t before = dctx->in.size();
        if (n) dctx->in.insert(dctx->in.end(), (const uint8_t*)srcBuffer, (const uint8_t*)srcBuffer + n);
        const size_t ext = frame_extent(dctx->in.data(), dctx->in.size(), &need, &upper);
        if (LZ4F_isError(ext)) { dctx->in.clear(); *srcSizePtr = 0; *dstSizePtr = 0; return ext; }
        if (!ext) { *srcSizePtr = n; *dstSizePtr = 0; return need; }
        consumed = ext - before;                                             // bytes behind the frame stay with the caller
        dctx->in.resize(ext);
        dctx->out.resize(upper ? upper : 1);
        size_t sn = ext, dn = upper;
        const size_t r = decode_whole(dctx->out.data(), &dn, dctx->in.data(), &sn);
        dctx->in.clear();
        if (r != 0) { dctx->out.clear(); *srcSizePtr = 0; *dstSizePtr = 0; return r; }
        dctx->out.resize(dn); dctx->outPos = 0; dctx->haveInfo = false;
    }
    const size_t left = dctx->out.size() - dctx->outPos, give = left < cap ? left : cap;
    if (give) memcpy(dstBuffer, dctx->out.data() + dctx->outPos, give);
    dctx->outPos += give;
    *srcSizePtr = consumed; *dstSizePtr = give;
    if (dctx->outPos >= dctx->out.size()) { dctx->out.clear(); dctx->outPos = 0; return 0; }
    return 1;
}

// ---- streaming compressor (lz4frame.c:570-1030): same buffering rules as the reference (a block is cut when blockSize bytes
//      are there, or at every call with autoFlush), so for LZ4F_blockIndependent the bytes equal the reference's for the same
//      sequence of calls; every LZ4F_compressUpdate compresses the blocks it completes in one launch ----
// State of the LZ4_stream_t a reference cctx owns, as far as it decides bytes (lz4.c:3179-3254, :972-1046, LZ4_prepareTable
// :810-853, LZ4_saveDict :3283-3311): the index of the next byte (currentOffset), the table's type, and WHERE the dictionary
// is - the reference picks prefix or external-dictionary mode by comparing addresses, so the emulation keeps addresses too:
// the caller's own for blocks compressed in place, offsets into an imaginary tmpBuff (kTmpSpace) for buffered data.
struct Lz4StreamState {
    uint32_t cur = 0; int type = 0;                          // type: 0 clearedTable, 2 byU32, 3 byU16 (lz4.h tableType_t)
    uint64_t dict = 0; uint32_t dictSize = 0;                // dictionary = [dict, dict + dictSize); 0 / 0 = none
};
constexpr uint64_t kTmpSpace = 0xA0C1ull << 48;              // tmpBuff + x (no user buffer lives there)

struct LZ4F_cctx_s {
    unsigned version; LZ4F_preferences_t prefs; int stage; size_t bs;
    std::vector<uint8_t> tmp; size_t tmpSize; unsigned long long total; Xxh32Stream xxh;
    // linked blocks (LZ4F_blockLinked): the stream the blocks of the frame continue
    bool linked = false; Lz4StreamState lz; uint32_t* d_table = nullptr; bool tableZero = false;
    std::vector<uint8_t> hist;                               // the last <= 64 KB of the stream
    size_t tmpInOff = 0, maxBufferSize = 0;                  // cctx->tmpIn - cctx->tmpBuff, cctx->maxBufferSize (lz4frame.c:640-655)
    ~LZ4F_cctx_s() { if (d_table) { aoclgpu::DeviceScope scope(aoclgpu_host_device()); (void)hipFree(d_table); } }
};

namespace {
// LZ4_prepareTable (lz4.c:810-853)
void lz_prepare(Lz4StreamState& z, bool& clearTable, uint32_t inputSize, int want)
{
    if (z.type != 0) {
        if (z.type != want || (want == 3 && z.cur + inputSize >= 0xFFFFu) || (want == 2 && z.cur > (1u << 30)) || inputSize >= 4096) {
            clearTable = true; z.cur = 0; z.type = 0;
        }
    }
    if (z.cur != 0 && want == 2) z.cur += 65536;
    z.dict = 0; z.dictSize = 0;
}
// what an independent block leaves in the stream state (LZ4_compress_fast_extState_fastReset, lz4.c:1408-1445, limited output)
void lz_after_independent_block(Lz4StreamState& z, bool& clearTable, uint32_t n)
{
    const int want = n < 65547 ? 3 : 2;
    lz_prepare(z, clearTable, n, want);
    z.cur += n; z.type = want; z.dictSize = n;               // (dictSize += inputSize, :1037; nobody reads it before the next reset)
}
// LZ4_saveDict(stream, tmpBuff, 64 KB) (lz4.c:3283-3311)
uint32_t lz_save_dict(Lz4StreamState& z)
{
    const uint32_t ds = z.dictSize < 65536 ? z.dictSize : 65536;
    z.dict = kTmpSpace; z.dictSize = ds;
    return ds;
}
struct StreamBlock { const uint8_t* p; uint32_t len; uint64_t addr; };   // addr: where the reference sees the block (user pointer or tmpBuff offset)
}  // namespace

extern "C" {

size_t LZ4F_createCompressionContext(LZ4F_cctx** cctxPtr, unsigned version)
{
    if (!cctxPtr) return err(E_GENERIC);
    LZ4F_cctx* c = new (std::nothrow) LZ4F_cctx();
    if (!c) { *cctxPtr = nullptr; return err(E_allocation_failed); }
    c->version = version; c->stage = 0; c->tmpSize = 0; c->total = 0; c->bs = 0; *cctxPtr = c;
    return 0;
}
size_t LZ4F_freeCompressionContext(LZ4F_cctx* cctx) { delete cctx; return 0; }

size_t LZ4F_compressBound(size_t srcSize, const LZ4F_preferences_t* prefs)
{   // lz4frame.c:724-730
    if (prefs && prefs->autoFlush) return bound_internal(srcSize, prefs, 0);
    return bound_internal(srcSize, prefs, (size_t)-1);
}

static size_t LZ4F_compressBegin_impl(LZ4F_cctx* cctx, void* dstBuffer, size_t dstCapacity, const LZ4F_preferences_t* prefsPtr)
{   // lz4frame.c:597-700
    if (!cctx || !dstBuffer) return err(E_GENERIC);
    if (dstCapacity < kMaxHeader) return err(E_dstMaxSize_tooSmall);
    if (prefsPtr) cctx->prefs = *prefsPtr; else memset(&cctx->prefs, 0, sizeof(cctx->prefs));
    if (cctx->prefs.compressionLevel >= 3) return err(E_compressionLevel_invalid);                            // only the fast compressor (negative: acceleration)
    if (cctx->prefs.frameInfo.blockSizeID == 0) cctx->prefs.frameInfo.blockSizeID = LZ4F_max64KB;
    cctx->bs = block_size((int)cctx->prefs.frameInfo.blockSizeID);
    if (!cctx->bs) return err(E_maxBlockSize_invalid);
    cctx->tmp.resize(cctx->bs); cctx->tmpSize = 0; cctx->total = 0; cctx->xxh.reset();
    cctx->linked = cctx->prefs.frameInfo.blockMode == LZ4F_blockLinked && block_linking();
    {   // buffer management (lz4frame.c:640-655): the size of tmpBuff decides when the buffered data wraps around
        const size_t need = cctx->prefs.autoFlush ? (cctx->prefs.frameInfo.blockMode == LZ4F_blockLinked ? 65536u : 0u)
                                                  : cctx->bs + (cctx->prefs.frameInfo.blockMode == LZ4F_blockLinked ? 131072u : 0u);
        if (cctx->maxBufferSize < need) cctx->maxBufferSize = need;
    }
    cctx->tmpInOff = 0;
    if (cctx->linked) {                                                   // LZ4F_initStream -> LZ4_resetStream_fast (lz4frame.c:570-590, :661-664)
        aoclgpu::DeviceScope scope(aoclgpu_host_device());                // the stream's table lives on the device its kernels run on
        if (!cctx->d_table) {                                            // (the stream state may already have a history: independent frames)
            if (hipMalloc((void**)&cctx->d_table, 16384) != hipSuccess) { cctx->d_table = nullptr; return err(E_allocation_failed); }
            if (hipMemset(cctx->d_table, 0, 16384) != hipSuccess) return err(E_GENERIC);
            cctx->tableZero = true;
        }
        bool clear = false;
        lz_prepare(cctx->lz, clear, 0, 2);
        if ((clear || cctx->lz.cur == 0) && !cctx->tableZero) {          // (a table that starts at index 0 is a zeroed table)
            if (hipMemset(cctx->d_table, 0, 16384) != hipSuccess) return err(E_GENERIC);
            cctx->tableZero = true;
        }
        cctx->hist.clear();                                              // (whatever an earlier frame left is more than 64 KB of indices away)
    }
    cctx->stage = 1;
    return write_header((uint8_t*)dstBuffer, cctx->prefs, !cctx->linked);
}

namespace {
// LZ4_compress_fast_continue on every block of `blks` in order (lz4.c:3179-3254): mode and limits of each block from the
// stream state and the addresses, the blocks themselves on the GPU in one launch (lz4_compress_stream_kernel), output as
// LZ4F_makeBlock writes it.  Returns the bytes written or an error code.
size_t make_stream_blocks(LZ4F_cctx* c, const std::vector<StreamBlock>& blks, uint8_t* op)
{
    const size_t nb = blks.size();
    if (!nb) return 0;
    uint8_t* const op0 = op;
    Lz4StreamState& z = c->lz;
    const int level = c->prefs.compressionLevel;
    const uint32_t accel = level < 0 ? (uint32_t)(-level + 1) : 1u;
    size_t total = c->hist.size();
    for (const StreamBlock& b : blks) total += b.len;
    {   // LZ4_renormDictT (lz4.c:3127-3146): indices are rebased before they pass 2 GB
        aoclgpu::DeviceScope scope(aoclgpu_host_device());
        size_t fresh = total - c->hist.size();
        if ((uint64_t)z.cur + fresh > 0x7FFF0000ull) {
            std::vector<uint32_t> t(4096);
            if (hipMemcpy(t.data(), c->d_table, 16384, hipMemcpyDeviceToHost) != hipSuccess) return err(E_GENERIC);
            const uint32_t delta = z.cur - 65536;
            for (uint32_t& v : t) v = v < delta ? 0 : v - delta;
            if (hipMemcpy(c->d_table, t.data(), 16384, hipMemcpyHostToDevice) != hipSuccess) return err(E_GENERIC);
            z.cur = 65536;
            if (z.dictSize > 65536) { z.dict += z.dictSize - 65536; z.dictSize = 65536; }
        }
    }
    const uint32_t idxBase = z.cur - (uint32_t)c->hist.size();
    std::vector<aoclgpu::Lz4StreamJob> jobs(nb);
    uint32_t off = (uint32_t)c->hist.size();
    for (size_t i = 0; i < nb; ++i) {
        const uint64_t src = blks[i].addr; const uint32_t n = blks[i].len;
        uint64_t dictEnd = z.dict + z.dictSize;
        if (z.dictSize - 1 < 3 && dictEnd != src) { z.dictSize = 0; z.dict = src; dictEnd = src; }         // tiny dictionaries (:3198-3205)
        {   const uint64_t srcEnd = src + n;                                                                 // overlapping input / dictionary (:3207-3215)
            if (srcEnd > z.dict && srcEnd < dictEnd) {
                uint64_t ds = dictEnd - srcEnd;
                if (ds > 65536) ds = 65536;
                if (ds < 4) ds = 0;
                z.dictSize = (uint32_t)ds; z.dict = dictEnd - ds;
            }
        }
        const bool prefix = dictEnd == src;
        const bool dictSmall = z.dictSize < 65536 && z.dictSize < z.cur;
        aoclgpu::Lz4StreamJob& j = jobs[i];
        memset(&j, 0, sizeof(j));
        j.off = off; j.len = n; j.ext = prefix ? 0u : 1u;
        j.lowCand = dictSmall ? z.cur - z.dictSize : 0u;
        {   // start of the dictionary as a position of the window (never below the window: 64 KB of history are there, a catch-up
            // cannot pass them - the match it extends is at most 65535 bytes behind a position of the block)
            const uint32_t dictIdx = z.cur - z.dictSize;
            j.lowPos = dictIdx > idxBase ? dictIdx - idxBase : 0u;
        }
        if (prefix) z.dictSize += n; else { z.dict = src; z.dictSize = n; }                                  // (:1037, :3251-3252)
        z.cur += n; z.type = 2;
        off += n;
    }
    aoclgpu::DeviceScope scope(aoclgpu_host_device());
    DevBuf D;
    const size_t slot = (c->bs + 15) & ~(size_t)15;
    uint8_t* d_src = (uint8_t*)D.get(total); uint8_t* d_dst = (uint8_t*)D.get(nb * slot); int32_t* d_len = (int32_t*)D.get(nb * 4);
    aoclgpu::Lz4StreamJob* d_jobs = (aoclgpu::Lz4StreamJob*)D.get(nb * sizeof(aoclgpu::Lz4StreamJob));
    if (!d_src || !d_dst || !d_len || !d_jobs) return err(E_allocation_failed);
    if (!c->hist.empty() && hipMemcpy(d_src, c->hist.data(), c->hist.size(), hipMemcpyHostToDevice) != hipSuccess) return err(E_GENERIC);
    for (size_t i = 0; i < nb; ++i)
        if (hipMemcpy(d_src + jobs[i].off, blks[i].p, blks[i].len, hipMemcpyHostToDevice) != hipSuccess) return err(E_GENERIC);
    if (hipMemcpy(d_jobs, jobs.data(), nb * sizeof(aoclgpu::Lz4StreamJob), hipMemcpyHostToDevice) != hipSuccess) return err(E_GENERIC);
    c->tableZero = false;
    if (aoclgpu::lz4_stream_blocks(d_src, idxBase, d_jobs, (uint32_t)nb, c->d_table, d_dst, (uint32_t)slot, d_len, accel, nullptr) != AOCLGPU_OK) return err(E_GENERIC);
    std::vector<int32_t> ol(nb);
    std::vector<uint8_t> slots(nb * slot);
    if (hipMemcpy(ol.data(), d_len, nb * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(slots.data(), d_dst, nb * slot, hipMemcpyDeviceToHost) != hipSuccess) return err(E_GENERIC);
    const bool blockCrc = c->prefs.frameInfo.blockChecksumFlag != 0;
    for (size_t i = 0; i < nb; ++i) {
        uint32_t cs = ol[i] > 0 ? (uint32_t)ol[i] : 0;
        if (cs == 0) { cs = blks[i].len; wr32(op, cs | kStored); memcpy(op + kBH, blks[i].p, cs); }      // stored block
        else { wr32(op, cs); memcpy(op + kBH, slots.data() + i * slot, cs); }
        if (blockCrc) wr32(op + kBH + cs, xxh32(op + kBH, cs, 0));
        op += kBH + cs + (blockCrc ? kBF : 0);
    }
    // the last 64 KB of the stream for the next call
    for (size_t i = 0; i < nb; ++i) c->hist.insert(c->hist.end(), blks[i].p, blks[i].p + blks[i].len);
    if (c->hist.size() > 65536) c->hist.erase(c->hist.begin(), c->hist.end() - 65536);
    return (size_t)(op - op0);
}
// Independent blocks do not use the stream's table here, but a reference cctx compresses them with the same LZ4_stream_t: what
// they leave behind (index, table type) decides how a later frame with linked blocks on the same cctx starts.
void track_independent_blocks(LZ4F_cctx* c, const std::vector<BlockRef>& blks)
{
    bool clr = false;
    for (const BlockRef& b : blks) lz_after_independent_block(c->lz, clr, b.len);
    if (clr && c->d_table && !c->tableZero) {
        aoclgpu::DeviceScope scope(aoclgpu_host_device());
        if (hipMemset(c->d_table, 0, 16384) == hipSuccess) c->tableZero = true;
    }
}
}  // namespace

static size_t LZ4F_compressUpdate_impl(LZ4F_cctx* cctx, void* dstBuffer, size_t dstCapacity, const void* srcBuffer, size_t srcSize,
                                        const LZ4F_compressOptions_t* opts)
{   // lz4frame.c:825-930
    (void)opts;
    if (!cctx || cctx->stage != 1 || !dstBuffer || (!srcBuffer && srcSize)) return err(E_GENERIC);
    if (dstCapacity < bound_internal(srcSize, &cctx->prefs, cctx->tmpSize)) return err(E_dstMaxSize_tooSmall);
    const size_t bs = cctx->bs;
    const uint8_t* sp = (const uint8_t*)srcBuffer; const uint8_t* const se = sp + srcSize;
    if (cctx->linked) {
        // lz4frame.c:849-923 with the stream behind it: which block is compressed where (in the caller's buffer or in tmpBuff,
        // behind the saved dictionary) decides prefix / external-dictionary mode and with it the bytes
        std::vector<StreamBlock> sb;
        std::vector<uint8_t> hold;                                        // the buffered block, completed (kept until the launch is done)
        int last = 0;                                                     // 1 fromTmpBuffer, 2 fromSrcBuffer
        if (cctx->tmpSize > 0) {
            const size_t toCopy = bs - cctx->tmpSize;
            if (toCopy > srcSize) { memcpy(cctx->tmp.data() + cctx->tmpSize, sp, srcSize); sp = se; cctx->tmpSize += srcSize; }
            else {
                memcpy(cctx->tmp.data() + cctx->tmpSize, sp, toCopy); sp += toCopy;
                hold.assign(cctx->tmp.begin(), cctx->tmp.begin() + bs);
                sb.push_back(StreamBlock{hold.data(), (uint32_t)bs, kTmpSpace + cctx->tmpInOff});
                last = 1; cctx->tmpInOff += bs; cctx->tmpSize = 0;
            }
        }
        while ((size_t)(se - sp) >= bs) { sb.push_back(StreamBlock{sp, (uint32_t)bs, (uint64_t)(uintptr_t)sp}); sp += bs; last = 2; }
        if (cctx->prefs.autoFlush && sp < se) { sb.push_back(StreamBlock{sp, (uint32_t)(se - sp), (uint64_t)(uintptr_t)sp}); sp = se; last = 2; }
        const size_t w = make_stream_blocks(cctx, sb, (uint8_t*)dstBuffer);
        if (LZ4F_isError(w)) return w;
        if (last == 2) {                                                  // preserve the dictionary (:895-904)
            if (opts && opts->stableSrc) cctx->tmpInOff = 0;
            else { const uint32_t ds = lz_save_dict(cctx->lz); if (!ds) return err(E_GENERIC); cctx->tmpInOff = ds; }
        }
        if (cctx->tmpInOff + bs > cctx->maxBufferSize && !cctx->prefs.autoFlush) cctx->tmpInOff = lz_save_dict(cctx->lz);   // keep tmpIn within limits (:906-912)
        if (sp < se) { memcpy(cctx->tmp.data(), sp, (size_t)(se - sp)); cctx->tmpSize = (size_t)(se - sp); }
        if (cctx->prefs.frameInfo.contentChecksumFlag) cctx->xxh.update(srcBuffer, srcSize);
        cctx->total += srcSize;
        return w;
    }
    std::vector<BlockRef> blks;
    if (cctx->tmpSize > 0) {                                              // complete the buffered block first
        const size_t toCopy = bs - cctx->tmpSize;
        if (toCopy > srcSize) { memcpy(cctx->tmp.data() + cctx->tmpSize, sp, srcSize); sp = se; cctx->tmpSize += srcSize; }
        else {
            memcpy(cctx->tmp.data() + cctx->tmpSize, sp, toCopy); sp += toCopy;
            blks.push_back(BlockRef{cctx->tmp.data(), (uint32_t)bs});
            cctx->tmpSize = 0;
        }
    }
    while ((size_t)(se - sp) >= bs) { blks.push_back(BlockRef{sp, (uint32_t)bs}); sp += bs; }
    if (cctx->prefs.autoFlush && sp < se) { blks.push_back(BlockRef{sp, (uint32_t)(se - sp)}); sp = se; }
    const size_t w = make_blocks(blks, bs, cctx->prefs.frameInfo.blockChecksumFlag != 0, (uint8_t*)dstBuffer,
                                 cctx->prefs.compressionLevel < 0 ? -cctx->prefs.compressionLevel + 1 : 1);
    if (LZ4F_isError(w)) return w;
    track_independent_blocks(cctx, blks);
    if (sp < se) { memcpy(cctx->tmp.data(), sp, (size_t)(se - sp)); cctx->tmpSize = (size_t)(se - sp); }   // < blockSize: buffered
    if (cctx->prefs.frameInfo.contentChecksumFlag) cctx->xxh.update(srcBuffer, srcSize);
    cctx->total += srcSize;
    return w;
}

static size_t LZ4F_flush_impl(LZ4F_cctx* cctx, void* dstBuffer, size_t dstCapacity, const LZ4F_compressOptions_t* opts)
{   // lz4frame.c:942-985
    (void)opts;
    if (!cctx) return err(E_GENERIC);
    if (cctx->tmpSize == 0) return 0;
    if (cctx->stage != 1 || !dstBuffer) return err(E_GENERIC);
    if (dstCapacity < cctx->tmpSize + kBH + kBF) return err(E_dstMaxSize_tooSmall);
    if (cctx->linked) {                                                   // lz4frame.c:956-973
        std::vector<StreamBlock> sb(1, StreamBlock{cctx->tmp.data(), (uint32_t)cctx->tmpSize, kTmpSpace + cctx->tmpInOff});
        const size_t w = make_stream_blocks(cctx, sb, (uint8_t*)dstBuffer);
        if (LZ4F_isError(w)) return w;
        cctx->tmpInOff += cctx->tmpSize; cctx->tmpSize = 0;
        if (cctx->tmpInOff + cctx->bs > cctx->maxBufferSize) cctx->tmpInOff = lz_save_dict(cctx->lz);
        return w;
    }
    std::vector<BlockRef> blks(1, BlockRef{cctx->tmp.data(), (uint32_t)cctx->tmpSize});
    const size_t w = make_blocks(blks, cctx->bs, cctx->prefs.frameInfo.blockChecksumFlag != 0, (uint8_t*)dstBuffer,
                                 cctx->prefs.compressionLevel < 0 ? -cctx->prefs.compressionLevel + 1 : 1);
    if (LZ4F_isError(w)) return w;
    track_independent_blocks(cctx, blks);
    cctx->tmpSize = 0;
    return w;
}

static size_t LZ4F_compressEnd_impl(LZ4F_cctx* cctx, void* dstBuffer, size_t dstCapacity, const LZ4F_compressOptions_t* opts)
{   // lz4frame.c:988-1030
    const size_t f = LZ4F_flush_impl(cctx, dstBuffer, dstCapacity, opts);
    if (LZ4F_isError(f)) return f;
    if (!cctx || cctx->stage != 1 || !dstBuffer) return err(E_GENERIC);
    uint8_t* op = (uint8_t*)dstBuffer + f;
    if (dstCapacity - f < 4) return err(E_dstMaxSize_tooSmall);
    wr32(op, 0); op += 4;
    if (cctx->prefs.frameInfo.contentChecksumFlag) {
        if (dstCapacity - f < 8) return err(E_dstMaxSize_tooSmall);
        wr32(op, cctx->xxh.digest()); op += 4;
    }
    cctx->stage = 0;
    if (cctx->prefs.frameInfo.contentSize && cctx->prefs.frameInfo.contentSize != cctx->total) return err(E_frameSize_wrong);
    return (size_t)(op - (uint8_t*)dstBuffer);
}

}  // extern "C"


// ---- the exported names: no C++ exception crosses the C boundary (std::bad_alloc / std::length_error from the host-side
//      staging vectors become ERROR_allocation_failed) ----
#define LZ4F_GUARD(call) try { return call; } catch (const std::exception&) { return err(E_allocation_failed); }
extern "C" {
size_t LZ4F_compressFrame(void* dstBuffer, size_t dstCapacity, const void* srcBuffer, size_t srcSize, const LZ4F_preferences_t* prefsPtr)
{ LZ4F_GUARD(LZ4F_compressFrame_impl(dstBuffer, dstCapacity, srcBuffer, srcSize, prefsPtr)) }
size_t LZ4F_getFrameInfo(LZ4F_dctx* dctx, LZ4F_frameInfo_t* info, const void* src, size_t* srcSizePtr)
{ LZ4F_GUARD(LZ4F_getFrameInfo_impl(dctx, info, src, srcSizePtr)) }
size_t LZ4F_decompress(LZ4F_dctx* dctx, void* dstBuffer, size_t* dstSizePtr, const void* srcBuffer, size_t* srcSizePtr,
                       const LZ4F_decompressOptions_t* opts)
{
    try { return LZ4F_decompress_impl(dctx, dstBuffer, dstSizePtr, srcBuffer, srcSizePtr, opts); }
    catch (const std::exception&) {
        if (dctx) { dctx->in.clear(); dctx->out.clear(); dctx->outPos = 0; }
        if (srcSizePtr) *srcSizePtr = 0;
        if (dstSizePtr) *dstSizePtr = 0;
        return err(E_allocation_failed);
    }
}
size_t LZ4F_compressBegin(LZ4F_cctx* cctx, void* dstBuffer, size_t dstCapacity, const LZ4F_preferences_t* prefsPtr)
{ LZ4F_GUARD(LZ4F_compressBegin_impl(cctx, dstBuffer, dstCapacity, prefsPtr)) }
size_t LZ4F_compressUpdate(LZ4F_cctx* cctx, void* dstBuffer, size_t dstCapacity, const void* srcBuffer, size_t srcSize,
                           const LZ4F_compressOptions_t* opts)
{ LZ4F_GUARD(LZ4F_compressUpdate_impl(cctx, dstBuffer, dstCapacity, srcBuffer, srcSize, opts)) }
size_t LZ4F_flush(LZ4F_cctx* cctx, void* dstBuffer, size_t dstCapacity, const LZ4F_compressOptions_t* opts)
{ LZ4F_GUARD(LZ4F_flush_impl(cctx, dstBuffer, dstCapacity, opts)) }
size_t LZ4F_compressEnd(LZ4F_cctx* cctx, void* dstBuffer, size_t dstCapacity, const LZ4F_compressOptions_t* opts)
{ LZ4F_GUARD(LZ4F_compressEnd_impl(cctx, dstBuffer, dstCapacity, opts)) }
}  // extern "C"
